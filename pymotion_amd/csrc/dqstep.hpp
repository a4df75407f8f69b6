// dqstep.hpp -- the quad-distributed composition step of to_root_dual_quat (pymotion/ops/skeleton.py:230-241): shared by the tile
// kernels of dq.hip and the wave-per-frame / joints-in-step kernel of dqwide.hip.  Lane c of a quad holds component c of the running
// root-space quaternion and of the translation written as the pure quaternion (0, t); see dq.hip for the layout of a slot.
#pragma once
#include "common.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------------
// Big-magnitude tiles (centimetre mocap, far-away roots: the test of fk.hip, kBigOffset / kBigRoot) take a PRECISE step.
// The reference composes in float64 (skeleton.py:230-241 on float64 arrays, dual_quat.py:32) and its output multiplies the
// running translation (hundreds of units) with the running quaternion: 0.5 (0, T_j) (x) Q_j.  An fp32 quaternion chain is
// off by ~4e-7 after ten joints, which TIMES |T| = 400 is 4-5 ulp of the largest dual component (measured 4.3 at J = 52;
// the accumulation of T itself is the smaller term: an emulation with a float64 quaternion chain and an fp32 translation
// chain reads 1.6 ulp, the other way round 4.4).  So on those tiles
//   * the quaternion chain runs in float64: lane c keeps component c as a double, the three foreign components of the
//     parent arrive as two v_mov_b32_dpp each, and the four products are float64 FMAs (fp32 x fp32 is exact there);
//   * a parent that is not the previous joint is re-read from the image as hi + lo, lo = an 8-bit residual in units of
//     2^-31 packed four to a word into the slot's spare eighth float (the fp32 image alone would put back 3e-8 per branch
//     point: 2.2 ulp at J = 52 in the same emulation);
//   * translations accumulate in 32-bit fixed point like fk's (integer adds do not round; scale from fx_scale), which is
//     what keeps a 128-joint chain at the bar (fp32 adds: 6.8 ulp there).
// The step is ~45 instructions against ~32, ~17 of them at the float64 rate: the walk of such a tile takes about twice as
// long, the tile as a whole ~20 % more.  Metre-scale tiles keep the fp32 step (their error is 6e-7 absolute).
// ---------------------------------------------------------------------------------------------------
template <int CTRL>  // CTRL = quad_perm selector byte: a | b << 2 | c << 4 | d << 6
__device__ __forceinline__ double quad_perm_f64(const double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// component c of pq (x) b with pq distributed over the quad in float64 and sb_k = S[c][k] b_{c xor k} as in dq_step_math
__device__ __forceinline__ double quad_qmul_f64(const double pq, const float b0, const float sb1, const float sb2, const float sb3) {
    // (one term at a time: left to itself the compiler moves all eight exchanges and the four conversions to the top, and the
    // sixteen registers that takes set the budget of the whole kernel)
    double q = quad_perm_f64<0x00>(pq) * (double)b0;
    asm volatile("" : "+v"(q));
    q = __builtin_fma(quad_perm_f64<0x55>(pq), (double)sb1, q);
    asm volatile("" : "+v"(q));
    q = __builtin_fma(quad_perm_f64<0xaa>(pq), (double)sb2, q);
    asm volatile("" : "+v"(q));
    return __builtin_fma(quad_perm_f64<0xff>(pq), (double)sb3, q);
}

// the rotation part of dq_step_math alone: x = pv x tt + pw tt, tt = 2 (pv x v)  (quat.py:320-334; w1 = 2 v_nextnext, w2 = 2 v_next)
__device__ __forceinline__ float dq_step_rot(const float pq, const float w1, const float w2) {
    float tt, an, ann, x;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %5 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, -%4, %6 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %4 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %4 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %0, %1 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, -%0, %2 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"
        : "=&v"(tt), "=&v"(an), "=&v"(ann), "=&v"(x)
        : "v"(pq), "v"(w1), "v"(w2));
    return x;
}

// The scale of the precise step's fixed-point translations (round 6): fk's fx_scale rounds the bound up to a power of two (an exact scaling of its fp32 products)
// and bounds a bone by its 1-norm; here the increments are float64 anyway, so the words use the range they have -- S = 0.99 x 2^30 / B with B from the bones'
// 2-NORMS (a coordinate of a rotated bone is at most its length) -- which is 1.4 bits more resolution on random offsets: on a 55-deep chain of 30-unit bones the
// words' rounding was a random walk of 2 ulp of the largest dual component (resolution 7.6e-6 at coordinates of ~64), now 0.8.  S and 1 / S are doubles:
// S x (1 / S) must be 1 to far better than an fp32 ulp, or every position comes back scaled.
struct FxScaleD { double S, invS; };
__device__ __forceinline__ bool fx_scale_exact(const float tbound, const float rmax, FxScaleD &fx) {
    const float B = uniform_f32(wave_max(rmax) + tbound);
    const float Sf = 1.06e9f * frcp(B);  // ~0.99 x 2^30 / B: any S at or below 2^30 / B will do -- one per cent of headroom for the roundings of the chain
    fx.S = (double)Sf;
    const double y = (double)frcp(Sf);   // ... but 1 / S must be THIS S's reciprocal: one Newton step in float64 (1e-14) instead of a float64 division per tile
    fx.invS = __builtin_fma(y, __builtin_fma(-fx.S, y, 1.0), y);
    return B < 1e30f && B > 0.0f;        // false for NaN / Inf / absurd magnitudes: the fp32 step, which propagates them like the reference
}
// shallow skeletons (below kDqF64RotMinDepth) keep fk's power-of-two scale: their increments and the conversion back are exact fp32 scalings
template <bool DEEP>
__device__ __forceinline__ bool fx_scale_for(const float tbound, const float rmax, FxScaleD &fx) {
    if constexpr (DEEP) return fx_scale_exact(tbound, rmax, fx);
    FxScale f;
    const bool ok = fx_scale(tbound, rmax, f);
    fx.S = (double)uniform_f32(f.S); fx.invS = (double)uniform_f32(f.invS);
    return ok;
}

// From this depth on the precise step rotates its bones in float64 (below: fp32 -- the 22-joint body is 7 deep, SMPL-H 10: a random walk of 0.5 ulp x sqrt(depth)
// stays under 2 ulp there, and the float64 rotation costs centimetre-scale tiles 6.5 %: same-box A/B, 2^20 x 22, 267-269 -> 285-286 us)
constexpr int kDqF64RotMinDepth = 12;
// The same rotation in float64, for the precise step (round 6).  Rotating a 30-unit bone in fp32 costs ~5e-6 per joint whatever the quaternion's precision, and
// down a chain that is a random walk: randomised fuzz runs read 3.4 ulp of the largest dual component on a 32-deep chain of 30-unit bones, 4.1 on a 55-deep one
// (tests/test_gpu_large_magnitude.py pins them).  With the products in float64 (the quaternion component already is; an offset is an exact fp32 input) what is
// left per joint is the rounding of the fixed-point word.  next = quad_perm [0,2,3,1] (0x78), next-next = [0,3,1,2] (0x9c); one term at a time, see quad_qmul_f64.
__device__ __forceinline__ double dq_step_rot_f64(const double pqd, const float w1, const float w2) {
    const double pn = quad_perm_f64<0x78>(pqd), pnn = quad_perm_f64<0x9c>(pqd);
    double tt = pn * (double)w1;
    tt = __builtin_fma(-pnn, (double)w2, tt);          // 2 (pv x v)
    asm volatile("" : "+v"(tt));
    double x = quad_perm_f64<0x9c>(tt) * pn;
    asm volatile("" : "+v"(x));
    x = __builtin_fma(-quad_perm_f64<0x78>(tt), pnn, x);  // pv x tt
    asm volatile("" : "+v"(x));
    return __builtin_fma(quad_perm_f64<0x00>(pqd), tt, x);  // + pw tt
}

// What a precise step leaves in the translation word of its slot: the fixed-point translation -- or, on lane 0 (whose
// translation component is the zero scalar part), the four 8-bit residuals qd - qh of the quad in units of 2^-31.
__device__ __forceinline__ int dq_pack_residual(const double qd, const float qh, const int ti, const int c) {
    // (rounded, not truncated: a truncated residual is a bias of a quarter unit per re-read, and the step-list kernel of dqwide.hip re-reads EVERY parent --
    // 2.07 ulp of the largest dual component on two 32-deep chains of 30-unit bones against 1.8 with the scheduled walk's register chains, round 6)
    int k = (int)__builtin_rint((qd - (double)qh) * 0x1p31);  // |residual| <= 2^-25 for |q| < 1: |k| <= 64
    k = k < -128 ? -128 : (k > 127 ? 127 : k);
    int pk = (k & 0xff) << (8 * c);
    pk |= __builtin_amdgcn_mov_dpp(pk, 0xb1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
    pk |= __builtin_amdgcn_mov_dpp(pk, 0x4e, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
    return (c == 0) ? pk : ti;
}

// One PRECISE step for the lane holding component c (see above).  pqd: the parent's component in float64; pti: the parent's
// fixed-point translation word.  Returns the float64 component; `qh` / `tword` are what goes into the slot.
template <bool DEEP>
__device__ __forceinline__ double dq_step_precise(const double pqd, const int pti, const float b, const float sb1, const float sb2,
                                                  const float sb3, const float vc, const float w1, const float w2, const float live,
                                                  const double S, const int c, float &qh, int &ti, int &tword) {
    const double qd = quad_qmul_f64(pqd, b, sb1, sb2, sb3);
    if constexpr (DEEP) {  // kDqF64RotMinDepth: the bone rotated in float64, the increment scaled in float64
        const double x = dq_step_rot_f64(pqd, w1, w2);
        ti = pti + (int)__builtin_rint(__builtin_fma((double)live, x, (double)vc) * S);
    } else {               // shallow skeletons: fp32 as in rounds 3-5 (S is a power of two there: an exact scaling)
        const float x = dq_step_rot((float)pqd, w1, w2);
        ti = pti + (int)__builtin_rintf(__builtin_fmaf(live, x, vc) * (float)S);
    }
    qh = (float)qd;
    tword = dq_pack_residual(qd, qh, ti, c);
    return qd;
}

// a parent re-read from the image: fp32 head + its 8-bit residual (units of 2^-31) out of the packed word
__device__ __forceinline__ double dq_parent_f64(const float head, const int packed, const int c) {
    const int k = (packed << (24 - 8 * c)) >> 24;  // sign-extended byte c
    return __builtin_fma((double)k, 0x1p-31, (double)head);
}

// One step of the quad walk for the lane holding component c, as ONE block of 12 VALU instructions
// with the quad exchanges folded into the DPP operand of the multiplies (these kernels sit near the
// VALU issue limit):
//   q = pq (x) b                                        quat.py:337-361, component-parallel:
//       q_c = sum_k S[c][k] pq_k b_{c xor k},  sb_k = S[c][k] b_{c xor k} prepared off the chain
//   t = live * (pv x tt + pw tt) + s,  tt = 2 (pv x v), s = live v_c + pt      quat.py:320-334
//       w1 = 2 v_nextnext, w2 = 2 v_next (from the joint table), `next` = quad_perm [0,2,3,1]
// pq must have been written at least two instructions earlier by VALU (the leading s_nop covers it;
// inside the block the instruction order keeps every VALU write two slots away from its DPP read).
__device__ __forceinline__ void dq_step_math(const float pq, const float s, const float b0, const float sb1,
                                             const float sb2, const float sb3, const float w1, const float w2,
                                             const float live, float &q, float &t) {
    float tt, an, ann, x;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %6, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %6, %12 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %6, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, -%6, %13 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %6, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %6 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %6, %11 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %6 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %5, %2, %3 quad_perm:[0,3,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, -%2, %4 quad_perm:[0,2,3,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %6, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %1, %14, %5, %7"
        : "=&v"(q), "=&v"(t), "=&v"(tt), "=&v"(an), "=&v"(ann), "=&v"(x)
        : "v"(pq), "v"(s), "v"(b0), "v"(sb1), "v"(sb2), "v"(sb3), "v"(w1), "v"(w2), "v"(live));
}

}  // namespace pm
