// trig.hpp -- sin / cos of the accuracy class the reference's NumPy calls have, and the Euler -> quaternion step built on them
// (elementwise.hip: quat.from_euler and friends; unroll.hip: the fused BVH ingest kernel).
#pragma once

#include "common.hpp"

namespace pm {

// ---- second wave: trig-heavy conversions (libm-grade accuracy: parity first) ----------------------------
// These are VALU-bound, not HBM-bound (SQ_INSTS_VALU: ~480 per 128 elements for from_euler with libm's sincosf,
// every SIMD cycle busy), so sin/cos use a lean path of the same accuracy class: k = rint(x 2/pi) and the
// reduction r = x - k pi/2 in double precision (three full-rate f64 instructions, exact to 2^-53 |x|: valid for
// any |x| < 1e8 with no Cody-Waite constant juggling and no Payne-Hanek), then the Cephes minimax polynomials
// on [-pi/4, pi/4] (|error| < 2^-24) and the quadrant fix-up.  Beyond 1e8 rad (nobody's Euler angle) libm takes over.
// LIBM = false drops the libm detour beyond 1e8 rad (its inlined code is what keeps a caller's record loop from unrolling: unroll.hip);
// NaN and Inf still come out as NaN, finite angles beyond 1e8 rad lose accuracy there.
template <bool LIBM = true>
__device__ __forceinline__ void sincos_rr(const float x, float &s, float &c) {
    // Ordinary angles (|x| < 4096: every Euler angle, every half angle of a rotation): k = rint(x 2/pi) and two FMAs against
    // pi/2 = 1.5707963705062866 (its fp32 value) - 4.3711390e-8: the first is exact or rounds a value of magnitude < 1 once, the
    // split is good to 2e-15 k.  Six fp32 instructions; the float64 reduction (convert, multiply, rint, fma, two converts at
    // the float64 rate) was 30 of the 190 instructions from_euler spent per element.  Larger arguments keep it (a branch the
    // wave takes only if a lane needs it).
    float kf = __builtin_rintf(x * 0.6366197723675814f);
    float r = __builtin_fmaf(kf, 4.3711390001862412e-08f, __builtin_fmaf(kf, -1.5707963705062866f, x));
    int k = (int)kf;
    if (__builtin_amdgcn_ballot_w64(!(fabsf(x) < 4096.0f)) != 0) {  // valid for any |x| < 1e8 with no constant juggling and no Payne-Hanek
        const double xd = (double)x, kd = rint(xd * 0.6366197723675814);
        const bool far = !(fabsf(x) < 4096.0f);
        r = far ? (float)fma(kd, -1.5707963267948966, xd) : r;
        k = far ? (int)kd : k;
    }
    const float r2 = r * r;
    const float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    const float cp = 1.0f - 0.5f * r2 + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    const bool swap = (k & 1) != 0;  // x = r + k pi/2:  k mod 4 = 0: (s, c)   1: (c, -s)   2: (-s, -c)   3: (-c, s)
    float ss = swap ? cp : sp, cc = swap ? sp : cp;
    ss = (k & 2) ? -ss : ss;
    cc = ((k + 1) & 2) ? -cc : cc;
    if constexpr (LIBM) {
        if (!(fabsf(x) < 1e8f)) { ss = sinf(x); cc = cosf(x); }  // huge, inf, NaN
    }
    s = ss; c = cc;
}


// rotations/quat.py:43-82 : q = q0 (x) (q1 (x) q2), each an axis rotation about o[k] (0 / 1 / 2 = x / y / z) by e[k] radians
template <bool LIBM = true>
__device__ __forceinline__ void euler2q(const float (&e)[3], const int (&o)[3], float (&q)[4]) {
    float sn[3], cs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sincos_rr<LIBM>(e[k] / 2.0f, sn[k], cs[k]);  // quat.py:38: half angles
    // Three DISTINCT axes (a, b, c) -- every Tait-Bryan order, the only ones the reference documents -- have a closed form: with
    // sigma = +1 for a cyclic order (xyz, yzx, zxy) and -1 otherwise,
    //     w = c0 c1 c2 - sigma s0 s1 s2      v[a] = s0 c1 c2 + sigma c0 s1 s2
    //     v[b] = c0 s1 c2 - sigma s0 c1 s2   v[c] = c0 c1 s2 + sigma s0 s1 c2
    // 13 multiply-adds and six selects for what two general Hamilton products of axis quaternions (two non-zeros each) spend
    // 41 instructions on.  Anything else (a repeated axis) takes the general products, a branch the wave skips otherwise.
    const int ia = o[0], ib = o[1], ic = o[2];
    const bool distinct = ia != ib && ib != ic && ia != ic;
    const float sg = ((ib - ia + 3) % 3 == 1) ? 1.0f : -1.0f;
    const float cc = cs[1] * cs[2], ss = sn[1] * sn[2], csn = cs[1] * sn[2], scn = sn[1] * cs[2];
    const float s0 = sg * sn[0];
    const float w = __builtin_fmaf(cs[0], cc, -(s0 * ss));
    const float va = __builtin_fmaf(sn[0], cc, sg * cs[0] * ss);
    const float vb = __builtin_fmaf(cs[0], scn, -(s0 * csn));
    const float vc = __builtin_fmaf(cs[0], csn, s0 * scn);
    q[0] = w;
#pragma unroll
    for (int m = 0; m < 3; ++m) q[1 + m] = (ia == m) ? va : ((ib == m) ? vb : vc);
    if (__builtin_amdgcn_ballot_w64(!distinct) != 0) {
        float qa[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            qa[k][0] = cs[k]; qa[k][1] = o[k] == 0 ? sn[k] : 0.0f; qa[k][2] = o[k] == 1 ? sn[k] : 0.0f; qa[k][3] = o[k] == 2 ? sn[k] : 0.0f;
        }
        float t[4], g[4];
        qmul(qa[1], qa[2], t);
        qmul(qa[0], t, g);
#pragma unroll
        for (int m = 0; m < 4; ++m) q[m] = distinct ? q[m] : g[m];
    }
}

}  // namespace pm
