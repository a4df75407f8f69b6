// elementwise.hip -- quaternion / dual-quaternion / ortho6d element-wise conversions for gfx950.
//
// One templated streaming kernel: a wave owns a tile of 128 elements (2 per lane; measured 64/128/256: small
// tiles = more resident waves win by 5-12 % on the narrow records, 128 is the best overall).  Each operand is
// an AoS record of W floats (W in {1,3,4,6,8,9}).  Records of 1, 3 or 4 floats move straight between HBM and
// registers (one record per lane = a contiguous stream); the wider / odd ones go HBM -> LDS (contiguous
// dwordx4, perfectly coalesced, tile bases are multiples of 1 KiB) -> registers (per-record read, widest
// conflict-free DS op) and back the same way.  All of these ops are HBM-bound (16-60 B per element, a few
// dozen flops): the kernel's job is to keep every byte it touches inside full 128-byte lines.
#include <stdlib.h>

#include "common.hpp"
#include "trig.hpp"

#ifdef PM_DEBUG
static int g_debug_shrink = 0;  // tests/test_gpu_debug_build.py: what the LDS checker is told the element-wise tiles own (0 = the truth)
#define PM_DEBUG_SHRINK g_debug_shrink
extern "C" void pm_debug_shrink_lds(int bytes) { g_debug_shrink = bytes; }
#else
#define PM_DEBUG_SHRINK 0
#endif

namespace pm {

constexpr int EW_TILE = 128;  // elements per wave (the default; EwTileOf below)
constexpr int EW_PER_LANE = EW_TILE / PM_WAVE;
// Elements per wave of ew_kernel, per op.  Round 4, same box, 64 / 128 / 256 per wave (us at 23 M elements): the Euler conversions want more
// in flight per wave (from_euler 246 / 112 / 105.5, to_euler 114 / 112.6 / 107.7; dq.to_rotation_translation 220 / 226.6 / 216 on one box and
// 213 -> 215 on the next: left at 128),
// the ops with two or three operand tiles fewer and more waves (from_to_axis 171 / 182.6 / 196.7, from_to 140 / 143 / 147, mul_vec 146.8 / 149.8 /
// 151.3, ortho6d.to_quat 141.7 / 145.4 / 147); the rest is best at 128 or indifferent (specialisations next to the entry points).
template <class Op> struct EwTileOf { static constexpr int v = EW_TILE; };

struct EwArgs {
    const float *in0, *in1, *in2;
    float *out0, *out1;
    const uint8_t *order;  // euler ops
    int64_t N;
    float eps;
    int flag;  // slerp: shortest; from_to: normalize_input; euler: 0 = one order, 1 = an order per element, P >= 2 = a table
               // of P orders, element e using row e % P (one order per joint of a [F, J, 3] clip)
    int64_t tile_e0;   // filled by the kernel: first element of the wave's tile ...
    int order_r0;      // ... and its row in the order table (tile_e0 % P), so that a row index is a 32-bit affair
    int order_pk;      // ... and, per element, its order as three packed bytes (fetched BEFORE the tile's operands: see ew_kernel)
};

// Records of 4 (dwordx4), 3 (dwordx3) or 1 float: one record per lane with consecutive lanes on consecutive
// records is already a contiguous, fully coalesced stream -- straight between HBM and registers.  The other
// widths (6, 8, 9) go through an LDS tile so that HBM only ever sees contiguous dwordx4 (measured on the
// 32-byte records of to_root_dual_quat: two dwordx4 per lane at a 32-byte stride are 5 % slower than the
// staged copy).
__host__ __device__ constexpr bool ew_direct_out(const int W) { return W == 1 || W == 3 || W == 4; }
__host__ __device__ constexpr bool ew_direct_in(const int W) { return false && ew_direct_out(W); }
__host__ __device__ constexpr int ew_lds_in(const int W) { return ew_direct_in(W) ? 0 : W; }
__host__ __device__ constexpr int ew_lds_out(const int W) { return ew_direct_out(W) ? 0 : W; }

template <int W, bool VEC>
__device__ __forceinline__ void ew_stage_in(const float *g, float *s, int64_t e0, int n, int lane) {
    if constexpr (W > 0 && !ew_direct_in(W)) tile_load<VEC>(g + e0 * W, s, n * W, lane);
}
template <int W, bool VEC>
__device__ __forceinline__ void ew_stage_out(float *g, const float *s, int64_t e0, int n, int lane) {
    if constexpr (W > 0 && !ew_direct_out(W)) tile_store<VEC>(g + e0 * W, s, n * W, lane);
}
// record `idx` of the tile -> registers (idx already clamped into the tile)
template <int W, bool VEC>
__device__ __forceinline__ void ew_get(const float *g, const float *s, const int64_t e0, const int idx, float (&x)[W ? W : 1]) {
    if constexpr (W == 0) {
    } else if constexpr (!ew_direct_in(W)) {
        lds_get<W>(s, idx, reinterpret_cast<float(&)[W]>(x));
    } else if constexpr (W == 4) {
        const float *p = g + (e0 + idx) * 4;
        if (VEC) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
        else { x[0] = p[0]; x[1] = p[1]; x[2] = p[2]; x[3] = p[3]; }
    } else if constexpr (W == 3) {
        const v3f_a4 t = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(g + (e0 + idx) * 3));
        x[0] = t.x; x[1] = t.y; x[2] = t.z;
    } else {
        x[0] = __builtin_nontemporal_load(g + e0 + idx);
    }
}
// registers -> record `idx` (valid slots only): straight to HBM, or into the LDS tile for the staged widths
template <int W, bool VEC>
__device__ __forceinline__ void ew_put(float *g, float *s, const int64_t e0, const int idx, const float (&y)[W ? W : 1]) {
    if constexpr (W == 0) {
    } else if constexpr (!ew_direct_out(W)) {
        lds_put<W>(s, idx, reinterpret_cast<const float(&)[W]>(y));
    } else if constexpr (W == 4) {
        float *p = g + (e0 + idx) * 4;
        if (VEC) __builtin_nontemporal_store(v4f{y[0], y[1], y[2], y[3]}, reinterpret_cast<v4f *>(p));
        else { p[0] = y[0]; p[1] = y[1]; p[2] = y[2]; p[3] = y[3]; }
    } else if constexpr (W == 3) {
        __builtin_nontemporal_store(v3f_a4{y[0], y[1], y[2]}, reinterpret_cast<v3f_a4 *>(g + (e0 + idx) * 3));
    } else {
        __builtin_nontemporal_store(y[0], g + e0 + idx);
    }
}

// Op concept: static constexpr int I0,I1,I2,O0,O1 (0 = absent);
//   static __device__ void apply(const float(&)[I0|1], const float(&)[I1|1], const float(&)[I2|1],
//                                float(&)[O0|1], float(&)[O1|1], const EwArgs&, int64_t elem)
struct OpToEuler;
template <class Op, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void ew_kernel(const EwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int I0 = Op::I0, I1 = Op::I1, I2 = Op::I2, O0 = Op::O0, O1 = Op::O1;
    constexpr int TILE = EwTileOf<Op>::v, PER_LANE = TILE / PM_WAVE;
    const int lane = threadIdx.x;
    const int64_t ntiles = (a.N + TILE - 1) / TILE;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t e0 = tile * TILE;
    const int n = (int)((a.N - e0) < TILE ? (a.N - e0) : TILE);

    float *s0 = smem;  // tiles of the staged operands only
    float *s1 = s0 + TILE * ew_lds_in(I0);
    float *s2 = s1 + TILE * ew_lds_in(I1);
    float *t0 = s2 + TILE * ew_lds_in(I2);
    float *t1 = t0 + TILE * ew_lds_out(O0);

    EwArgs b = a;
    b.tile_e0 = e0;
    b.order_r0 = (a.order != nullptr && a.flag >= 2) ? (int)(e0 % a.flag) : 0;  // wave-uniform, once per tile
    // Euler orders: three bytes per element from a side table.  Requested here, ahead of the operands -- inside the op they were a
    // second, dependent trip to memory per tile (to_euler: waves waiting 77 % of their time, 51 % of the HBM spec).
    // (to_euler needs its order first thing; from_euler only after three sincos, which hide the trip -- fetched early it was 3.5 % slower)
    int opk[PER_LANE];
#pragma unroll
    for (int m = 0; m < PER_LANE; ++m) opk[m] = 0;
    if constexpr (__is_same(Op, OpToEuler)) {
#pragma unroll
        for (int m = 0; m < PER_LANE; ++m) {
            const int idx = m * PM_WAVE + lane, ic = idx < n ? idx : n - 1;
            int64_t row = 0;
            if (a.flag == 1) row = e0 + ic;
            else if (a.flag >= 2) row = (unsigned)(b.order_r0 + ic) % (unsigned)a.flag;  // < P + TILE: 32-bit
            const uint8_t *p = a.order + row * 3;
            opk[m] = (int)p[0] | ((int)p[1] << 8) | ((int)p[2] << 16);
        }
    }
    ew_stage_in<I0, VEC>(a.in0, s0, e0, n, lane);
    ew_stage_in<I1, VEC>(a.in1, s1, e0, n, lane);
    ew_stage_in<I2, VEC>(a.in2, s2, e0, n, lane);
    wave_sync();
    // Reads and arithmetic are unconditional (a slot past a partial tile re-reads the tile's last record), so
    // that the elements of a lane are scheduled and packed together with no exec-mask branch between them;
    // only the write-back is guarded.
    float x0[PER_LANE][I0 ? I0 : 1], x1[PER_LANE][I1 ? I1 : 1], x2[PER_LANE][I2 ? I2 : 1];
#pragma unroll
    for (int m = 0; m < PER_LANE; ++m) {
        const int idx = m * PM_WAVE + lane, ic = idx < n ? idx : n - 1;
        ew_get<I0, VEC>(a.in0, s0, e0, ic, x0[m]);
        ew_get<I1, VEC>(a.in1, s1, e0, ic, x1[m]);
        ew_get<I2, VEC>(a.in2, s2, e0, ic, x2[m]);
    }
#pragma unroll
    for (int m = 0; m < PER_LANE; ++m) {
        const int idx = m * PM_WAVE + lane, ic = idx < n ? idx : n - 1;
        float y0[O0 ? O0 : 1], y1[O1 ? O1 : 1];
        b.order_pk = opk[m];
        Op::apply(x0[m], x1[m], x2[m], y0, y1, b, e0 + ic);
        if (idx < n) {
            ew_put<O0, VEC>(a.out0, t0, e0, idx, y0);
            ew_put<O1, VEC>(a.out1, t1, e0, idx, y1);
        }
    }
    wave_sync();
    ew_stage_out<O0, VEC>(a.out0, t0, e0, n, lane);
    ew_stage_out<O1, VEC>(a.out1, t1, e0, n, lane);
}

template <class Op>
static int launch_ew(const EwArgs &a, pm_stream_t stream, const char *name) {
    if (a.N < 0) { set_error("%s: negative N", name); return PM_EINVAL; }
    if (a.N == 0) return PM_OK;
    if ((Op::I0 && !a.in0) || (Op::I1 && !a.in1) || (Op::I2 && !a.in2) || (Op::O0 && !a.out0) || (Op::O1 && !a.out1)) {
        set_error("%s: null pointer", name);
        return PM_EINVAL;
    }
    constexpr int TILE = EwTileOf<Op>::v;
    constexpr size_t lds = (size_t)TILE * (ew_lds_in(Op::I0) + ew_lds_in(Op::I1) + ew_lds_in(Op::I2) + ew_lds_out(Op::O0) + ew_lds_out(Op::O1)) * sizeof(float);
    const int64_t ntiles = (a.N + TILE - 1) / TILE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("%s: grid too large", name); return PM_EUNSUPPORTED; }
    const bool vec = aligned16(a.in0) && aligned16(a.in1) && aligned16(a.in2) && aligned16(a.out0) && aligned16(a.out1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PM_SET_LDS(PM_DEBUG_SHRINK > 0 ? (size_t)PM_DEBUG_SHRINK : lds);
    if (vec) hipLaunchKernelGGL((ew_kernel<Op, true>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    else hipLaunchKernelGGL((ew_kernel<Op, false>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH(name);
}

#define PM_OP(NAME, i0, i1, i2, o0, o1)                                                          \
    struct NAME {                                                                                \
        static constexpr int I0 = i0, I1 = i1, I2 = i2, O0 = o0, O1 = o1;                        \
        static __device__ __forceinline__ void apply(const float(&x0)[i0 ? i0 : 1], const float(&x1)[i1 ? i1 : 1], \
                                                     const float(&x2)[i2 ? i2 : 1], float(&y0)[o0 ? o0 : 1],       \
                                                     float(&y1)[o1 ? o1 : 1], const EwArgs &a, int64_t elem)
#define PM_OP_END };

// rotations/quat.py:411-423
PM_OP(OpNormalize, 4, 0, 0, 4, 0) { qnormalize(x0, a.eps, y0); } PM_OP_END
// rotations/quat.py:364-376
PM_OP(OpLength, 4, 0, 0, 1, 0) { y0[0] = __fsqrt_rn(x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2] + x0[3] * x0[3]); } PM_OP_END
// rotations/quat.py:276-317
PM_OP(OpToMatrix, 4, 0, 0, 9, 0) { q2m(x0, y0); } PM_OP_END
// rotations/quat.py:85-156
PM_OP(OpFromMatrix, 9, 0, 0, 4, 0) { m2q(x0, y0); } PM_OP_END
// rotations/quat.py:337-361
PM_OP(OpMul, 4, 4, 0, 4, 0) { qmul(x0, x1, y0); } PM_OP_END
// rotations/quat.py:320-334
PM_OP(OpMulVec, 4, 3, 0, 3, 0) { qmulvec(x0, x1, y0); } PM_OP_END
// rotations/quat.py:396-408
PM_OP(OpConj, 4, 0, 0, 4, 0) { y0[0] = x0[0]; y0[1] = -x0[1]; y0[2] = -x0[2]; y0[3] = -x0[3]; } PM_OP_END
// rotations/dual_quat.py:12-36
PM_OP(OpDqFromRt, 4, 3, 0, 8, 0) { rt2dq(x0, x1, y0); } PM_OP_END
// rotations/dual_quat.py:62-83
PM_OP(OpDqToRt, 8, 0, 0, 4, 3) { dq2rt(x0, y0, y1); } PM_OP_END
// rotations/dual_quat.py:39-59
PM_OP(OpDqFromT, 3, 0, 0, 8, 0) {
    y0[0] = 1.0f; y0[1] = y0[2] = y0[3] = y0[4] = 0.0f;
    y0[5] = x0[0] * 0.5f; y0[6] = x0[1] * 0.5f; y0[7] = x0[2] * 0.5f;
} PM_OP_END
// rotations/ortho6d.py:67-90
// zero / non-finite / (anti-)parallel columns are re-done in float64 (o6d2m's `ill`): what the reference returns for them is
// decided by digits fp32 does not have
PM_OP(OpO6dToMatrix, 6, 0, 0, 9, 0) {
    bool ill;
    o6d2m(x0, y0, ill);
    if (__builtin_amdgcn_ballot_w64(ill) != 0) {
        double md[9];
        o6d2m_f64(x0, a.eps, md);
#pragma unroll
        for (int k = 0; k < 9; ++k) y0[k] = ill ? (float)md[k] : y0[k];
    }
} PM_OP_END
// rotations/ortho6d.py:50-64
PM_OP(OpO6dToQuat, 6, 0, 0, 4, 0) {
    float m[9];
    bool ill;
    o6d2m(x0, m, ill);
    m2q(m, y0);
    if (__builtin_amdgcn_ballot_w64(ill) != 0) {
        double md[9], qd[4];
        o6d2m_f64(x0, a.eps, md);
        m2q_f64(md, qd);
#pragma unroll
        for (int k = 0; k < 4; ++k) y0[k] = ill ? (float)qd[k] : y0[k];
    }
} PM_OP_END
// rotations/ortho6d.py:14-28
PM_OP(OpO6dFromQuat, 4, 0, 0, 6, 0) {
    float m[9]; q2m(x0, m);
    y0[0] = m[0]; y0[1] = m[1]; y0[2] = m[3]; y0[3] = m[4]; y0[4] = m[6]; y0[5] = m[7];
} PM_OP_END
// rotations/ortho6d.py:31-47
PM_OP(OpO6dFromMatrix, 9, 0, 0, 6, 0) {
    y0[0] = x0[0]; y0[1] = x0[1]; y0[2] = x0[3]; y0[3] = x0[4]; y0[4] = x0[6]; y0[5] = x0[7];
} PM_OP_END

// (sincos_rr and the Euler -> quaternion step live in trig.hpp: unroll.hip's fused BVH ingest kernel uses them too)

// atan2 of the same class (Cephes atanf: reduction at tan(pi/8) + degree-4 polynomial in z = u^2, |error| < 2^-23
// relative), with np.arctan2's quadrant / signed-zero conventions; 0/0, inf and NaN operands go to libm (a branch
// the wave skips when no lane needs it -- a select would drag libm's code along for every element).
__device__ __forceinline__ float atan2_rr(const float y, const float x) {
    const float ay = fabsf(y), ax = fabsf(x);
    const float hi = fmaxf(ay, ax), lo = fminf(ay, ax);
    // t = lo / hi in [0, 1], reduced at tan(pi / 8): u = (t - 1) / (t + 1) = (lo - hi) / (lo + hi) -- decided and formed
    // before the division, so there is ONE reciprocal (round 2: lo rcp(hi), then rcp(t + 1))
    const bool mid = lo > 0.4142135623730950f * hi;
    const float u = (mid ? lo - hi : lo) * frcp(mid ? lo + hi : hi);
    const float z = u * u;
    float a = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * u + u;
    a += mid ? 0.7853981633974483f : 0.0f;
    a = (ay > ax) ? 1.5707963267948966f - a : a;           // atan(ay / ax) in [0, pi/2]
    a = (x < 0.0f) ? 3.141592653589793f - a : a;           // left half plane
    a = copysignf(a, y);
    if (!((hi > 0.0f) && (hi < 3.0e38f) && (lo == lo))) a = atan2f(y, x);
    return a;
}

// atan2(sqrt(p), sqrt(q)) for p, q >= 0 (the middle Euler angle, quat.py:214: np.arctan2 of two hypotenuses): the square
// roots are monotonic, so the reduced argument is sqrt(min / max) -- one hardware sqrt and one rcp instead of two correctly
// rounded square roots and a general atan2 (quadrants, signs): same polynomial, first quadrant only.  atan2(0, 0) = 0.
__device__ __forceinline__ float atan2_sqrt_rr(const float p, const float q) {
    const float hi = fmaxf(p, q), lo = fminf(p, q);
    const float t = fsqrt(lo * frcp(hi));                  // in [0, 1]
    const bool mid = t > 0.4142135623730950f;
    const float u = mid ? (t - 1.0f) * frcp(t + 1.0f) : t;
    const float z = u * u;
    float a = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * u + u;
    a += mid ? 0.7853981633974483f : 0.0f;
    a = (p > q) ? 1.5707963267948966f - a : a;
    a = (hi > 0.0f) ? a : ((hi == 0.0f) ? 0.0f : hi);      // 0/0 -> 0 like np.arctan2(0, 0); NaN stays NaN
    return (hi < 3.0e38f) ? a : atan2f(sqrtf(p), sqrtf(q)); // inf operands: libm's conventions
}

// rotations/quat.py:24-40
__device__ __forceinline__ void aa2q(float angle, float ax, float ay, float az, float (&o)[4]) {
    const float h = angle / 2.0f;
    float s, c;
    sincos_rr(h, s, c);
    o[0] = c; o[1] = s * ax; o[2] = s * ay; o[3] = s * az;
}
PM_OP(OpFromAngleAxis, 1, 3, 0, 4, 0) { aa2q(x0[0], x1[0], x1[1], x1[2], y0); } PM_OP_END
// rotations/quat.py:6-21 (zero vector: 0/0 -> NaN, like the reference)
PM_OP(OpFromScaledAA, 3, 0, 0, 4, 0) {
    const float ang = __fsqrt_rn(x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2]);
    aa2q(ang, x0[0] / ang, x0[1] / ang, x0[2] / ang, y0);
} PM_OP_END
// rotations/quat.py:247-273
__device__ __forceinline__ void q2aa(const float (&q)[4], float &angle, float (&axis)[3]) {
    const float w = q[0];
    angle = 2.0f * acosf(fminf(fmaxf(w, -1.0f), 1.0f));
    // 1 - w^2 cancels near |w| = 1 (small rotations, where the axis matters most): in fp32 its 6e-8 absolute error is a
    // 3e-4 relative error of s at a 1-degree rotation.  The product and the difference are exact in float64 (three
    // full-rate instructions); only the result is rounded.
    const float s = __fsqrt_rn(fminf(fmaxf((float)__builtin_fma(-(double)w, (double)w, 1.0), 0.0f), 1.0f));
    const bool ok = s > 1e-8f;
    axis[0] = ok ? q[1] / s : 0.0f; axis[1] = ok ? q[2] / s : 0.0f; axis[2] = ok ? q[3] / s : 0.0f;
}
PM_OP(OpToAngleAxis, 4, 0, 0, 1, 3) { q2aa(x0, y0[0], y1); } PM_OP_END
// rotations/quat.py:230-244
PM_OP(OpToScaledAA, 4, 0, 0, 3, 0) {
    float ang, ax[3]; q2aa(x0, ang, ax);
    y0[0] = ang * ax[0]; y0[1] = ang * ax[1]; y0[2] = ang * ax[2];
} PM_OP_END

__device__ __forceinline__ void load_order(const EwArgs &a, int64_t elem, int (&o)[3]) {  // from the side table
    int64_t row = 0;
    if (a.flag == 1) row = elem;
    else if (a.flag >= 2) row = (unsigned)(a.order_r0 + (int)(elem - a.tile_e0)) % (unsigned)a.flag;  // < P + EW_TILE: 32-bit
    const uint8_t *p = a.order + row * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
__device__ __forceinline__ void unpack_order(const EwArgs &a, int (&o)[3]) {  // fetched by the kernel ahead of the operands
    o[0] = a.order_pk & 0xff; o[1] = (a.order_pk >> 8) & 0xff; o[2] = (a.order_pk >> 16) & 0xff;
}
// rotations/quat.py:43-82 : q = q0 (x) (q1 (x) q2), each an axis rotation about order[k]
PM_OP(OpFromEuler, 3, 0, 0, 4, 0) {
    int o[3]; load_order(a, elem, o);
    euler2q(x0, o, y0);
} PM_OP_END
// rotations/quat.py:159-227
PM_OP(OpToEuler, 4, 0, 0, 3, 0) {
    int o[3]; unpack_order(a, o);
    const int i = o[2], j = o[1], k = o[0];
    const int prod = (i - j) * (j - k) * (k - i);
    const float sg = (float)(prod >= 0 ? prod / 2 : -((-prod + 1) / 2));  // python floor division
    // (opaque copies: left as array elements, the three selects become x0[i + 1] -- a dynamically indexed array, i.e. 48 bytes of
    // scratch memory per lane and a trip to it per element: the kernel ran at 51 % with its waves waiting 77 % of the time)
    float vx = x0[1], vy = x0[2], vz = x0[3];
    asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz));
    const float qi = (i == 0) ? vx : (i == 1 ? vy : vz);
    const float qj = (j == 0) ? vx : (j == 1 ? vy : vz);
    const float qk = (k == 0) ? vx : (k == 1 ? vy : vz);
    // (these are correctly rounded differences and sums of two fp32 numbers: exactly what forming them in float64 and rounding
    // once gives, bit for bit -- round 2 went through float64 for them, twelve instructions at its rate)
    const float qks = qk * sg;  // sg = +-1: exact
    const float aa = x0[0] - qj, bb = qi + qks, cc = qj + x0[0], dd = qks - qi;
    const float two_pi = 6.283185307179586f;
    float e[3];
    // (np.hypot on quaternion-sized operands: no overflow to guard against, plain sqrt of the sum of squares)
    e[1] = 2.0f * atan2_sqrt_rr(cc * cc + dd * dd, aa * aa + bb * bb) - 1.5707963267948966f;
    const float hs = atan2_rr(bb, aa), hd = atan2_rr(dd, cc);
    e[2] = hs - hd;
    e[0] = (hs + hd) * sg;
    // np.mod(e, 2pi): the result carries the divisor's sign.  The half sums lie in [-pi, pi], so e[0], e[2] in [-2pi, 2pi] and the
    // middle angle in [-pi/2, pi/2]: one conditional addition, and for the outer angles one conditional subtraction for what was
    // >= 2pi BEFORE it.  A tiny negative angle (-1e-8: rounding noise of hs - hd on a single-axis rotation) comes out as
    // fp32(2pi), the fp32 neighbour of the 6.2831853 the reference's float64 np.mod returns there -- not as 0, which is 2pi
    // away from it element-wise.
    y0[1] = (e[1] < 0.0f) ? e[1] + two_pi : e[1];
#pragma unroll
    for (int c = 0; c < 3; c += 2) {
        const float r = e[c];
        y0[c] = (r < 0.0f) ? r + two_pi : ((r >= two_pi) ? r - two_pi : r);
    }
} PM_OP_END
// rotations/quat.py:465-501
PM_OP(OpSlerp, 4, 4, 1, 4, 0) {
    float b[4] = {x1[0], x1[1], x1[2], x1[3]};
    float dot = x0[0] * b[0] + x0[1] * b[1] + x0[2] * b[2] + x0[3] * b[3];
    if (a.flag && dot < 0.0f) { b[0] = -b[0]; b[1] = -b[1]; b[2] = -b[2]; b[3] = -b[3]; dot = -dot; }
    dot = fminf(fmaxf(dot, -1.0f), 1.0f);
    const float th = acosf(dot) * x2[0];
    float q2[4], nn = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { q2[c] = b[c] - x0[c] * dot; const float u = q2[c] + 0.000001f; nn += u * u; }
    nn = __fsqrt_rn(nn);
    float cs, sn;
    sincos_rr(th, sn, cs);
#pragma unroll
    for (int c = 0; c < 4; ++c) y0[c] = cs * x0[c] + sn * (q2[c] / nn);
} PM_OP_END

// rotations/quat.py:504-576 / :579-650 (flag = normalize_input)
PM_OP(OpFromTo, 3, 3, 0, 4, 0) { from_to(x0, x1, a.flag != 0, y0); } PM_OP_END
PM_OP(OpFromToAxis, 3, 3, 3, 4, 0) { from_to_axis(x0, x1, x2, a.flag != 0, y0); } PM_OP_END

static EwArgs mk(const float *i0, const float *i1, const float *i2, float *o0, float *o1, int64_t N, float eps = 0.0f,
                 int flag = 0, const uint8_t *order = nullptr) {
    EwArgs a;
    a.in0 = i0; a.in1 = i1; a.in2 = i2; a.out0 = o0; a.out1 = o1; a.order = order; a.N = N; a.eps = eps; a.flag = flag;
    a.tile_e0 = 0; a.order_r0 = 0; a.order_pk = 0;
    return a;
}

// ---------------------------------------------------------------------------------------------------
// dual_quat.normalize / is_unit (rotations/dual_quat.py:86-136).  The reference decides ONE branch for
// the whole batch from a global `.all()`; here the kernels accumulate violation counts in three device
// ints (wave ballot + one atomic per wave) and the host front-end reads them, exactly where the
// reference's Python `if` synchronises.
//   flags[0] += #(|qr|^2 not close to 0)   flags[1] += #(|qr|^2 not close to 1)   flags[2] += #(qr.qd not close to 0)
// np.isclose semantics: |a - b| <= atol + rtol |b| with rtol 1e-5, atol 1e-8 (NaN is never close).
// MODE 0: out = [qr, qd] / |qr|, flags evaluated on that result (normalize's is_unit test, :102-106)
// MODE 1: out = [qr/|qr|, qd/|qr| - qr/|qr| * (qr.qd)/|qr|^2]  (the orthogonalising branch, :107-113)
// MODE 2: no output, flags evaluated on the input (is_unit itself)
// ---------------------------------------------------------------------------------------------------
template <int MODE, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void dq_norm_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t N,
                                                          float atol, int *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int64_t ntiles = (N + EW_TILE - 1) / EW_TILE;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t e0 = tile * EW_TILE;
    const int n = (int)((N - e0) < EW_TILE ? (N - e0) : EW_TILE);
    float *sIn = smem, *sOut = smem + EW_TILE * 8;
    tile_load<VEC>(in + e0 * 8, sIn, n * 8, lane);
    wave_sync();
    int bad0 = 0, bad1 = 0, bad2 = 0;
#pragma unroll
    for (int m = 0; m < EW_PER_LANE; ++m) {
        const int idx = m * PM_WAVE + lane;
        if (idx < n) {
            float d[8], o[8];
            lds_get<8>(sIn, idx, d);
            const float n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
            const float dot = d[0] * d[4] + d[1] * d[5] + d[2] * d[6] + d[3] * d[7];
            float sq = n2, dt = dot;
            if (MODE != 2) {
                const float nrm = fsqrt(n2);
                const float inv = 1.0f / nrm;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = d[c] * inv;
                if (MODE == 1) {
                    const float k = dot / (nrm * nrm);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[4 + c] -= o[c] * k;
                }
                sq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
                dt = o[0] * o[4] + o[1] * o[5] + o[2] * o[6] + o[3] * o[7];
                lds_put<8>(sOut, idx, o);
            }
            bad0 += !(fabsf(sq) <= 1e-8f);
            bad1 += !(fabsf(sq - 1.0f) <= 1e-8f + 1e-5f);
            bad2 += !(fabsf(dt) <= atol);
        }
    }
    if (flags) {
        for (int off = 32; off > 0; off >>= 1) {
            bad0 += __shfl_down(bad0, off);
            bad1 += __shfl_down(bad1, off);
            bad2 += __shfl_down(bad2, off);
        }
        if (lane == 0) {
            // The caller only asks "any violation?" (the reference's `.all()`): raise a flag once, and let every later
            // wave see it with a plain (L2-served) load instead of queueing another atomic on the same word -- on a
            // batch where everything violates, one atomic per wave serialises 10^5 waves (4-6 ms at 23 M records).
            auto raise = [&](int *f, const int bad) {
                if (bad && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicAdd(f, bad);
            };
            raise(flags, bad0);
            raise(flags + 1, bad1);
            raise(flags + 2, bad2);
        }
    }
    if (MODE != 2) {
        wave_sync();
        tile_store<VEC>(out + e0 * 8, sOut, n * 8, lane);
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming ceiling probe: fk's traffic shape (contiguous tiles, rd floats in / wr floats out per
// frame) with no arithmetic.  Same one-wave-per-workgroup tiling, data passes through LDS once.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PM_WAVE) void ceiling_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          int64_t F, int rd, int wr, int fpw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int64_t ntiles = (F + fpw - 1) / fpw;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t f0 = tile * fpw;
    const int nf = (int)((F - f0) < fpw ? (F - f0) : fpw);
    tile_load<true>(src + f0 * rd, smem, nf * rd, lane);
    wave_sync();
    // replicate the staged input over the (larger) output tile
    const v4f *l4 = reinterpret_cast<const v4f *>(smem);
    v4f *g4 = reinterpret_cast<v4f *>(dst + f0 * wr);
    const int n4 = (nf * wr) >> 2, m4 = (nf * rd) >> 2;
    for (int i = lane; i < n4; i += PM_WAVE) __builtin_nontemporal_store(l4[i % m4], g4 + i);
}

// Plain streaming probe without LDS: every thread reads one dwordx4 and writes `ratio` dwordx4 (each store
// instruction of a wave covers 1 KiB contiguous).  256-thread blocks, grid-stride.  What the memory system
// sustains for a given read:write mix, independent of any tiling of ours.
template <int MODE>  // bit 0: nontemporal loads, bit 1: nontemporal stores
__global__ __launch_bounds__(256) void plain_stream_kernel(const v4f *__restrict__ src, v4f *__restrict__ dst, int64_t n4,
                                                           int ratio) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (ratio == 0) {  // pure read: the loads feed a sum that is stored only if it comes out as a value it cannot have
        v4f acc = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += (MODE & 1) ? __builtin_nontemporal_load(src + i) : src[i];
        if (acc.x + acc.y + acc.z + acc.w == -1.2345678e37f) dst[threadIdx.x] = acc;
        return;
    }
    if (ratio < 0) {  // pure write: n4 dwordx4 of a pattern, nothing read
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const v4f v = v4f{(float)(int)i, 1.0f, 2.0f, 3.0f};
            if (MODE & 2) __builtin_nontemporal_store(v, dst + i);
            else dst[i] = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const v4f v = (MODE & 1) ? __builtin_nontemporal_load(src + i) : src[i];
        const int64_t w = i >> 6, l = i & 63;
        for (int k = 0; k < ratio; ++k) {
            if (MODE & 2) __builtin_nontemporal_store(v, dst + (w * ratio + k) * 64 + l);
            else dst[(w * ratio + k) * 64 + l] = v;
        }
    }
}

// (see EwTileOf)
template <> struct EwTileOf<OpFromEuler> { static constexpr int v = 256; };
template <> struct EwTileOf<OpToEuler> { static constexpr int v = 256; };
template <> struct EwTileOf<OpFromToAxis> { static constexpr int v = 64; };
template <> struct EwTileOf<OpFromTo> { static constexpr int v = 64; };
template <> struct EwTileOf<OpMulVec> { static constexpr int v = 64; };
template <> struct EwTileOf<OpO6dToQuat> { static constexpr int v = 64; };

}  // namespace pm

using namespace pm;

extern "C" int pm_quat_normalize_f32(const float *q, int64_t N, float eps, float *out, pm_stream_t s) {
    return launch_ew<OpNormalize>(mk(q, nullptr, nullptr, out, nullptr, N, eps), s, "quat_normalize");
}
extern "C" int pm_quat_length_f32(const float *q, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpLength>(mk(q, nullptr, nullptr, out, nullptr, N), s, "quat_length");
}
extern "C" int pm_quat_to_matrix_f32(const float *q, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpToMatrix>(mk(q, nullptr, nullptr, out, nullptr, N), s, "quat_to_matrix");
}
extern "C" int pm_quat_from_matrix_f32(const float *m, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpFromMatrix>(mk(m, nullptr, nullptr, out, nullptr, N), s, "quat_from_matrix");
}
extern "C" int pm_quat_mul_f32(const float *q0, const float *q1, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpMul>(mk(q0, q1, nullptr, out, nullptr, N), s, "quat_mul");
}
extern "C" int pm_quat_mul_vec_f32(const float *q, const float *v, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpMulVec>(mk(q, v, nullptr, out, nullptr, N), s, "quat_mul_vec");
}
extern "C" int pm_quat_conjugate_f32(const float *q, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpConj>(mk(q, nullptr, nullptr, out, nullptr, N), s, "quat_conjugate");
}
extern "C" int pm_dq_from_rt_f32(const float *q, const float *t, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpDqFromRt>(mk(q, t, nullptr, out, nullptr, N), s, "dq_from_rt");
}
extern "C" int pm_dq_to_rt_f32(const float *dq, int64_t N, float *q, float *t, pm_stream_t s) {
    return launch_ew<OpDqToRt>(mk(dq, nullptr, nullptr, q, t, N), s, "dq_to_rt");
}
extern "C" int pm_dq_from_t_f32(const float *t, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpDqFromT>(mk(t, nullptr, nullptr, out, nullptr, N), s, "dq_from_t");
}
extern "C" int pm_o6d_to_matrix_f32(const float *x, int64_t N, float eps, float *out, pm_stream_t s) {
    return launch_ew<OpO6dToMatrix>(mk(x, nullptr, nullptr, out, nullptr, N, eps), s, "o6d_to_matrix");
}
extern "C" int pm_o6d_to_quat_f32(const float *x, int64_t N, float eps, float *out, pm_stream_t s) {
    return launch_ew<OpO6dToQuat>(mk(x, nullptr, nullptr, out, nullptr, N, eps), s, "o6d_to_quat");
}
extern "C" int pm_o6d_from_quat_f32(const float *q, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpO6dFromQuat>(mk(q, nullptr, nullptr, out, nullptr, N), s, "o6d_from_quat");
}
extern "C" int pm_o6d_from_matrix_f32(const float *m, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpO6dFromMatrix>(mk(m, nullptr, nullptr, out, nullptr, N), s, "o6d_from_matrix");
}
extern "C" int pm_quat_from_angle_axis_f32(const float *angle, const float *axis, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpFromAngleAxis>(mk(angle, axis, nullptr, out, nullptr, N), s, "quat_from_angle_axis");
}
extern "C" int pm_quat_from_scaled_angle_axis_f32(const float *v, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpFromScaledAA>(mk(v, nullptr, nullptr, out, nullptr, N), s, "quat_from_scaled_angle_axis");
}
extern "C" int pm_quat_to_angle_axis_f32(const float *q, int64_t N, float *angle, float *axis, pm_stream_t s) {
    return launch_ew<OpToAngleAxis>(mk(q, nullptr, nullptr, angle, axis, N), s, "quat_to_angle_axis");
}
extern "C" int pm_quat_to_scaled_angle_axis_f32(const float *q, int64_t N, float *out, pm_stream_t s) {
    return launch_ew<OpToScaledAA>(mk(q, nullptr, nullptr, out, nullptr, N), s, "quat_to_scaled_angle_axis");
}
extern "C" int pm_quat_from_euler_f32(const float *euler, const uint8_t *order, int order_per_element, int64_t N,
                                      float *out, pm_stream_t s) {
    PM_CHECK_ARGS(order != nullptr || N == 0, "quat_from_euler: null order");
    PM_CHECK_ARGS(order_per_element >= 0, "quat_from_euler: order_per_element must be 0, 1 or a table length");
    return launch_ew<OpFromEuler>(mk(euler, nullptr, nullptr, out, nullptr, N, 0.0f, order_per_element, order), s, "quat_from_euler");
}
extern "C" int pm_quat_to_euler_f32(const float *q, const uint8_t *order, int order_per_element, int64_t N, float *out,
                                    pm_stream_t s) {
    PM_CHECK_ARGS(order != nullptr || N == 0, "quat_to_euler: null order");
    PM_CHECK_ARGS(order_per_element >= 0, "quat_to_euler: order_per_element must be 0, 1 or a table length");
    return launch_ew<OpToEuler>(mk(q, nullptr, nullptr, out, nullptr, N, 0.0f, order_per_element, order), s, "quat_to_euler");
}
extern "C" int pm_quat_slerp_f32(const float *q0, const float *q1, const float *t, int64_t N, int shortest, float *out,
                                 pm_stream_t s) {
    return launch_ew<OpSlerp>(mk(q0, q1, t, out, nullptr, N, 0.0f, shortest), s, "quat_slerp");
}

extern "C" int pm_stream_ceiling_f32(const float *src, float *dst, int64_t F, int32_t rd, int32_t wr, pm_stream_t stream) {
    PM_CHECK_ARGS(src && dst && F >= 0 && rd >= 4 && wr >= 4 && rd % 4 == 0 && wr % 4 == 0 && aligned16(src) && aligned16(dst),
                  "stream_ceiling: need aligned pointers and rd, wr multiples of 4 floats");
    if (F == 0) return PM_OK;
    // fk's tile for that frame size: 16 frames while seven tiles fit a CU's LDS (the 22-joint skeleton), 4 beyond (fk.hip: dispatch_fk)
    const int fpw = tune_env("PM_CEIL_FPW", 7 * 16 * (size_t)wr * sizeof(float) <= kMaxLds ? 16 : 4);
    const size_t lds = (size_t)fpw * (rd > wr ? rd : wr) * sizeof(float);
    if (lds > 64 * 1024) { set_error("stream_ceiling: tile too large"); return PM_EUNSUPPORTED; }
    const int64_t ntiles = (F + fpw - 1) / fpw;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    PM_SET_LDS(lds);
    hipLaunchKernelGGL(ceiling_kernel, dim3((unsigned)grid), dim3(PM_WAVE), lds, static_cast<hipStream_t>(stream), src, dst, F,
                       (int)rd, (int)wr, fpw);
    return PM_AFTER_LAUNCH("stream_ceiling");
}

extern "C" int pm_dq_normalize_f32(const float *dq, int64_t N, int orthogonalize, float atol, float *out, int32_t *flags,
                                   pm_stream_t stream) {
    PM_CHECK_ARGS(N >= 0, "dq_normalize: negative N");
    if (N == 0) return PM_OK;
    PM_CHECK_ARGS(dq && out, "dq_normalize: null pointer");
    const int64_t ntiles = (N + EW_TILE - 1) / EW_TILE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("dq_normalize: grid too large"); return PM_EUNSUPPORTED; }
    const size_t lds = (size_t)EW_TILE * 16 * sizeof(float);
    const bool vec = aligned16(dq) && aligned16(out);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (orthogonalize) {
        PM_SET_LDS(lds);
        if (vec) hipLaunchKernelGGL((dq_norm_kernel<1, true>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, out, N, atol, flags);
        else hipLaunchKernelGGL((dq_norm_kernel<1, false>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, out, N, atol, flags);
    } else {
        PM_SET_LDS(lds);
        if (vec) hipLaunchKernelGGL((dq_norm_kernel<0, true>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, out, N, atol, flags);
        else hipLaunchKernelGGL((dq_norm_kernel<0, false>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, out, N, atol, flags);
    }
    return PM_AFTER_LAUNCH("dq_normalize");
}

extern "C" int pm_dq_unit_flags_f32(const float *dq, int64_t N, float atol, int32_t *flags, pm_stream_t stream) {
    PM_CHECK_ARGS(N >= 0, "dq_unit_flags: negative N");
    if (N == 0) return PM_OK;
    PM_CHECK_ARGS(dq && flags, "dq_unit_flags: null pointer");
    const int64_t ntiles = (N + EW_TILE - 1) / EW_TILE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("dq_unit_flags: grid too large"); return PM_EUNSUPPORTED; }
    const size_t lds = (size_t)EW_TILE * 16 * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    PM_SET_LDS(lds);
    if (aligned16(dq)) hipLaunchKernelGGL((dq_norm_kernel<2, true>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, nullptr, N, atol, flags);
    else hipLaunchKernelGGL((dq_norm_kernel<2, false>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, dq, nullptr, N, atol, flags);
    return PM_AFTER_LAUNCH("dq_unit_flags");
}

extern "C" int pm_stream_plain_f32(const float *src, float *dst, int64_t n4, int32_t ratio, int32_t blocks, pm_stream_t stream) {
    PM_CHECK_ARGS(src && dst && n4 >= 0 && ratio >= -1 && blocks >= 1 && aligned16(src) && aligned16(dst), "stream_plain: bad arguments");
    if (n4 == 0) return PM_OK;
    const int mode = tune_env("PM_PLAIN_MODE", 3);  // PM_TUNING build only: 0..3 = nontemporal {none, loads, stores, both}
    auto *s4 = reinterpret_cast<const v4f *>(src);
    auto *d4 = reinterpret_cast<v4f *>(dst);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (mode & 3) {
        case 0: hipLaunchKernelGGL(plain_stream_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, (int)ratio); break;
        case 1: hipLaunchKernelGGL(plain_stream_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, (int)ratio); break;
        case 2: hipLaunchKernelGGL(plain_stream_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, (int)ratio); break;
        default: hipLaunchKernelGGL(plain_stream_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, s4, d4, n4, (int)ratio); break;
    }
    return PM_AFTER_LAUNCH("stream_plain");
}

extern "C" int pm_quat_from_to_f32(const float *v1, const float *v2, int64_t N, int normalize_input, float *out, pm_stream_t s) {
    return launch_ew<OpFromTo>(mk(v1, v2, nullptr, out, nullptr, N, 0.0f, normalize_input), s, "quat_from_to");
}
extern "C" int pm_quat_from_to_axis_f32(const float *v1, const float *v2, const float *axis, int64_t N, int normalize_input,
                                        float *out, pm_stream_t s) {
    return launch_ew<OpFromToAxis>(mk(v1, v2, axis, out, nullptr, N, 0.0f, normalize_input), s, "quat_from_to_axis");
}
