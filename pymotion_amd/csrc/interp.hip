// interp.hip -- linear resampling of a clip along its time axis for gfx950.
//   time.interpolate_positions   pymotion/ops/time.py:4-66   (torch twin: ops/time_torch.py)
// The reference gathers positions[..., idx, ...] and positions[..., idx + 1, ...] along the time axis and
// blends them with (1 - w), w (time.py:49-64); idx / w come from a searchsorted over the two 1-D time
// arrays (S and T entries -- tiny next to the payload, computed by the caller and passed as device arrays).
// Viewed as [A, T, B] -> [A, S, B] (A = product of the axes before the time axis, B = after) the op is a
// streaming row gather: every output row of B floats reads two input rows.  One lane per dwordx4 of the
// output (B % 4 == 0 and 16-byte aligned pointers: a vector never straddles a row) or per float; consecutive
// lanes are consecutive in memory on both sides, so loads and stores are fully coalesced; neighbouring
// samples that fall between the same two frames re-read the same rows out of L2.
#include "common.hpp"

namespace pm {

struct InterpArgs {
    const float *pos;    // [A, T, B]
    const int32_t *idx;  // [S]  0 <= idx <= T - 2
    const float *w;      // [S]
    float *out;          // [A, S, B]
    int64_t A, T, S, B;
};

template <bool VEC>
__global__ __launch_bounds__(256) void interp_linear_kernel(const InterpArgs a) {
    constexpr int VW = VEC ? 4 : 1;
    const int64_t Bv = a.B / VW;                       // row length in lane-elements
    const int64_t nv = a.A * a.S * Bv;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const int64_t row = i / Bv, b = i - row * Bv;      // output row (a, s), position inside the row
    const int64_t aa = row / a.S, s = row - aa * a.S;
    const int32_t i0 = a.idx[s];
    const float w = a.w[s], u = 1.0f - w;              // time.py:61-64: (1 - w) * p[idx] + w * p[idx + 1]
    const float *p0 = a.pos + ((aa * a.T + i0) * a.B) + b * VW;
    const float *p1 = p0 + a.B;
    float *o = a.out + row * a.B + b * VW;
    if (VEC) {
        const v4f x = *reinterpret_cast<const v4f *>(p0), y = *reinterpret_cast<const v4f *>(p1);
        const v4f r = {u * x.x + w * y.x, u * x.y + w * y.y, u * x.z + w * y.z, u * x.w + w * y.w};
        __builtin_nontemporal_store(r, reinterpret_cast<v4f *>(o));
    } else {
        o[0] = u * p0[0] + w * p1[0];
    }
}

}  // namespace pm

extern "C" int pm_interpolate_linear_f32(const float *positions, const int32_t *idx, const float *weights, int64_t A,
                                         int64_t T, int64_t S, int64_t B, float *out, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(A >= 0 && S >= 0 && B >= 0 && T >= 2, "interpolate_linear: need A, S, B >= 0 and T >= 2");
    if (A == 0 || S == 0 || B == 0) return PM_OK;
    PM_CHECK_ARGS(positions && idx && weights && out, "interpolate_linear: null pointer");
    InterpArgs a;
    a.pos = positions; a.idx = idx; a.w = weights; a.out = out; a.A = A; a.T = T; a.S = S; a.B = B;
    const bool vec = (B % 4 == 0) && aligned16(positions) && aligned16(out);
    const int64_t nv = A * S * (vec ? B / 4 : B);
    const int64_t grid = (nv + 255) / 256;
    if (grid > 0x7fffffffLL) { set_error("interpolate_linear: grid too large"); return PM_EUNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(interp_linear_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(interp_linear_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return check_hip(hipGetLastError(), "interpolate_linear launch");
}
