// interp.hip -- linear resampling of a clip along its time axis for gfx950.
//   time.interpolate_positions   pymotion/ops/time.py:4-66   (torch twin: ops/time_torch.py)
// The reference gathers positions[..., idx, ...] and positions[..., idx + 1, ...] along the time axis and
// blends them with (1 - w), w (time.py:49-64); idx / w come from a searchsorted over the two 1-D time
// arrays (S and T entries -- tiny next to the payload, computed by the caller and passed as device arrays).
// Viewed as [A, T, B] -> [A, S, B] (A = product of the axes before the time axis, B = after) the op is a
// streaming row gather: every output row of B floats reads two input rows.  One lane per dwordx4 / dwordx2 of
// the output (row length a multiple of 4 / 2 floats: a vector never straddles a row) or per float; consecutive
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

constexpr int IP_PER_THREAD = 4;                       // lane-elements per thread
constexpr int IP_BLOCK = 256 * IP_PER_THREAD;          // per workgroup: a contiguous run of the flattened output

template <int VW>  // floats per lane-element: 4 / 2 when the row length allows (a vector never straddles a row), else 1
__global__ __launch_bounds__(256) void interp_linear_kernel(const InterpArgs a) {
    const int64_t Bv = a.B / VW;                       // row length in lane-elements
    const int64_t nv = a.A * a.S * Bv;
    const int64_t base = (int64_t)blockIdx.x * IP_BLOCK;
    // Where the block's run starts: one 64-bit divide per wave (uniform), then every lane-element is a small
    // offset from it and finds its row with a float-reciprocal divide (exact: offsets stay below 2^22).
    const int64_t row0 = base / Bv;
    const int b0 = (int)(base - row0 * Bv);
    const int64_t a0 = row0 / a.S;
    const int64_t s0 = row0 - a0 * a.S;
    const float invB = 1.0f / (float)Bv, invS = 1.0f / (float)a.S;
    const bool narrow = Bv < (1 << 21);                // otherwise a block never leaves its first two rows
    const bool short_s = a.S < (1 << 21);              // likewise for the sample axis
#pragma unroll
    for (int k = 0; k < IP_PER_THREAD; ++k) {
        const int o = k * 256 + threadIdx.x;
        if (base + o >= nv) break;
        int dr;                                        // rows past row0 (<= 1024)
        int64_t b;
        if (narrow) {
            const int e = b0 + o;                      // < Bv + 1024
            dr = (int)(((float)e + 0.5f) * invB);
            b = e - dr * (int)Bv;
        } else {
            const int64_t e = (int64_t)b0 + o;
            dr = e >= Bv ? 1 : 0;
            b = e >= Bv ? e - Bv : e;
        }
        const int64_t row = row0 + dr;
        int64_t aa, s;
        if (short_s) {
            const int t = (int)s0 + dr, q = (int)(((float)t + 0.5f) * invS);
            aa = a0 + q; s = t - q * (int)a.S;
        } else {
            const int64_t t = s0 + dr;
            aa = a0 + (t >= a.S ? 1 : 0); s = t >= a.S ? t - a.S : t;
        }
        const int32_t i0 = a.idx[s];
        const float w = a.w[s], u = 1.0f - w;          // time.py:61-64: (1 - w) * p[idx] + w * p[idx + 1]
        const float *p0 = a.pos + ((aa * a.T + i0) * a.B) + b * VW;
        const float *p1 = p0 + a.B;
        float *o_ = a.out + row * a.B + b * VW;
        if constexpr (VW == 4) {
            const v4f x = *reinterpret_cast<const v4f *>(p0), y = *reinterpret_cast<const v4f *>(p1);
            const v4f r = {u * x.x + w * y.x, u * x.y + w * y.y, u * x.z + w * y.z, u * x.w + w * y.w};
            __builtin_nontemporal_store(r, reinterpret_cast<v4f *>(o_));
        } else if constexpr (VW == 2) {
            const v2f x = *reinterpret_cast<const v2f *>(p0), y = *reinterpret_cast<const v2f *>(p1);
            const v2f r = {u * x.x + w * y.x, u * x.y + w * y.y};
            __builtin_nontemporal_store(r, reinterpret_cast<v2f *>(o_));
        } else {
            __builtin_nontemporal_store(u * p0[0] + w * p1[0], o_);
        }
    }
}

// Rows whose length is not a multiple of four floats (66 = 22 joints x 3, 63, ...): a lane still moves dwordx4 -- gfx9 global
// loads and stores only need 4-byte alignment -- counted per row, the last one of a row holding its B % 4 floats (per-float
// accesses, one lane in ceil(B / 4)).  Round 2 fell back to dwordx2 / single floats per lane there: 58 % / 40 % of the HBM spec
// against 78-82 % for rows of 72 / 156 floats.
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float v2f_a4 __attribute__((ext_vector_type(2), aligned(4)));

__global__ __launch_bounds__(256) void interp_linear_flat_kernel(const InterpArgs a) {
    // lane-elements are dwordx4 counted PER ROW: V = ceil(B / 4) to a row, the last one of a row holding B % 4 floats
    const int64_t B = a.B, V = (B + 3) >> 2, nv = a.A * a.S * V;
    const int64_t base = (int64_t)blockIdx.x * IP_BLOCK;
    const int64_t row0 = base / V;                              // one 64-bit divide per wave (uniform)
    const int v0 = (int)(base - row0 * V);
    const int64_t a0 = row0 / a.S;
    const int64_t s0 = row0 - a0 * a.S;
    const float invV = 1.0f / (float)V, invS = 1.0f / (float)a.S;
    const bool narrow = V < (1 << 21), short_s = a.S < (1 << 21);
    const int tail = (int)(B & 3);                              // floats of a row's last vector (0: full)
#pragma unroll
    for (int k = 0; k < IP_PER_THREAD; ++k) {
        const int o = k * 256 + threadIdx.x;
        if (base + o >= nv) break;
        int dr;
        int64_t v;
        if (narrow) {
            const int e = v0 + o;
            dr = (int)(((float)e + 0.5f) * invV);
            v = e - dr * (int)V;
        } else {
            const int64_t e = (int64_t)v0 + o;
            dr = e >= V ? 1 : 0;
            v = e >= V ? e - V : e;
        }
        int64_t aa, s;
        if (short_s) {
            const int t = (int)s0 + dr, q = (int)(((float)t + 0.5f) * invS);
            aa = a0 + q; s = t - q * (int)a.S;
        } else {
            const int64_t t = s0 + dr;
            aa = a0 + (t >= a.S ? 1 : 0); s = t >= a.S ? t - a.S : t;
        }
        const int32_t i0 = a.idx[s];
        const float w = a.w[s], u = 1.0f - w;                  // time.py:61-64
        const float *p0 = a.pos + ((aa * a.T + i0) * B) + 4 * v;
        float *o_ = a.out + (row0 + dr) * B + 4 * v;
        // A row's last vector holds B % 4 floats.  Its LOADS are whole dwordx4 like everybody's (the floats past the row's end are the next
        // row's first ones, read and dropped) -- per-float loads and stores there cost the whole wave six more memory instructions per vector
        // for one lane in 17 (rows of 66 floats: 39.7 us against 31-32 us for rows of 64 / 72) -- unless that would read past the end of the
        // array (the last vector of the last frame pair); its store is one dwordx2 (or one / three floats).
        const bool whole = tail == 0 || v != V - 1;
        const bool at_end = aa == a.A - 1 && (int64_t)i0 + 2 == a.T;
        if (whole || !at_end) {
            const v4f_a4 x = *reinterpret_cast<const v4f_a4 *>(p0), y = *reinterpret_cast<const v4f_a4 *>(p0 + B);
            const v4f_a4 r = v4f_a4{u * x.x + w * y.x, u * x.y + w * y.y, u * x.z + w * y.z, u * x.w + w * y.w};
            if (whole) *reinterpret_cast<v4f_a4 *>(o_) = r;
            else if (tail == 2) *reinterpret_cast<v2f_a4 *>(o_) = v2f_a4{r.x, r.y};
            else { o_[0] = r.x; if (tail == 3) { o_[1] = r.y; o_[2] = r.z; } }
        } else {
            for (int j = 0; j < tail; ++j) o_[j] = u * p0[j] + w * p0[B + j];
        }
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
    const int vw = ((B % 4 == 0) && aligned16(positions) && aligned16(out)) ? 4
                   : ((B % 2 == 0) && ((uintptr_t)positions % 8 == 0) && ((uintptr_t)out % 8 == 0)) ? 2 : 1;
    const int64_t nv = A * S * (B / vw);
    const int64_t grid = (nv + IP_BLOCK - 1) / IP_BLOCK;
    if (grid > 0x7fffffffLL) { set_error("interpolate_linear: grid too large"); return PM_EUNSUPPORTED; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vw != 4 && B >= 4) {  // any row length: dword-aligned dwordx4 per row (see interp_linear_flat_kernel)
        const int64_t n4 = A * S * ((B + 3) >> 2);
        const int64_t gridf = (n4 + IP_BLOCK - 1) / IP_BLOCK;
        if (gridf > 0x7fffffffLL) { set_error("interpolate_linear: grid too large"); return PM_EUNSUPPORTED; }
        hipLaunchKernelGGL(interp_linear_flat_kernel, dim3((unsigned)gridf), dim3(256), 0, s, a);
        return PM_AFTER_LAUNCH("interpolate_linear launch");
    }
    if (vw == 4) hipLaunchKernelGGL(interp_linear_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else if (vw == 2) hipLaunchKernelGGL(interp_linear_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(interp_linear_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return PM_AFTER_LAUNCH("interpolate_linear launch");
}
