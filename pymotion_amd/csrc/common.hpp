// common.hpp -- shared device helpers for the gfx950 kernels of libpmhip.so.
//
// Execution model used by every kernel in this library: ONE WAVE (64 lanes) PER WORKGROUP
// owns a contiguous tile of frames / elements and a private LDS slice.  The tile moves
//     HBM --(coalesced 16 B/lane loads)--> LDS --(per-lane AoS reads)--> VALU
//     VALU --(per-lane AoS writes)--> LDS --(coalesced 16 B/lane stores)--> HBM
// so no s_barrier is ever needed: LDS operations of one wave execute in program order, and
// the only fence required is a compiler-level one (wave_sync below).  Waves on a CU are fully
// independent pipelines, which is what keeps HBM requests in flight while others compute.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pmhip.h"

#define PM_WAVE 64
#define PM_NXCD 8

namespace pm {

typedef float v4f __attribute__((ext_vector_type(4)));  // native vectors: nontemporal builtins need them
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f_a4 __attribute__((ext_vector_type(3), aligned(4)));  // 12-byte record at 4-byte alignment (global dwordx3)

// ---- PM_DEBUG build (libpmhip_debug.so, `make debug`) -------------------------------------------------------
// Every LDS access that goes through the helpers of this file (tile copies, lds_get / lds_put, the image copies of
// fk.hip) is checked against the workgroup's dynamic LDS allocation; a violation is recorded (first offender wins:
// source line | 1 << 31) instead of performed, and every launch is followed by a device synchronisation, the HIP error
// check and a read of that record, so the failing call returns PM_EHIP with the line.  The production build compiles
// all of it away.  (The tree walks' raw pointer reads -- e.g. the "slack past the last joint" look-ahead -- stay inside
// the allocation by construction of the launchers' LDS sizes; tests/test_gpu_debug_build.py runs them under this build
// with guard words around every global buffer.)
#ifdef PM_DEBUG
static __device__ unsigned int g_pm_lds_bytes = 0;   // set by the launcher (post_launch serialises the stream anyway)
static __device__ unsigned int g_pm_violation = 0;
__device__ __forceinline__ bool pm_lds_ok(const void *p, const unsigned bytes, const int line) {
    const unsigned off = (unsigned)(uintptr_t)p;  // generic pointer into the LDS aperture: the low 32 bits are the LDS offset
    if (off + bytes <= g_pm_lds_bytes) return true;
    atomicCAS(&g_pm_violation, 0u, 0x80000000u | (unsigned)line);
    return false;
}
#define PM_LDS_OK(p, bytes) pm::pm_lds_ok((p), (bytes), __LINE__)
#else
#define PM_LDS_OK(p, bytes) true
#endif

// Compiler-level ordering of LDS traffic inside one wave (hardware already executes a wave's
// DS instructions in order).  Emits no instruction beyond the waitcnt the compiler needs.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// blockIdx -> tile index.  Workgroup b runs on XCD b % 8 (observed dispatch order), and tiles that are neighbours in memory should run on ONE XCD: the partial
// cache lines at their boundaries are then written by workgroups that share an L2 and merge there before going to HBM (measured: HBM traffic of every skeleton
// kernel within 1.1 % of its algorithmic bytes).  Rounds 1-5 gave every XCD one contiguous EIGHTH of the tiles.  Round 6 (profiles/r06_levels.txt): the eight
// XCDs then sweep eight windows an eighth of the array apart, in step, for the whole launch -- and on most placements of the arrays in physical memory that costs
// DRAM efficiency (the same kernel reads 150 or 167 us by the allocation; the L2's read queue towards the fabric holds a request 8-15 % longer on the slow ones).
// With the XCDs' ranges cut into CHUNKS that take turns -- XCD x owns tiles [(8 c + x) chunk, (8 c + x + 1) chunk), c = 0, 1, ... -- neighbours inside a chunk
// still share an L2 (one boundary in `chunk` is between XCDs) and the eight XCDs sweep ONE window, 8 chunks wide.  Measured with production libraries of the commits
// before and after on three or four allocation sets of each of three or four boxes (profiles/r06_levels.txt blocks 12-14): fk J = 52 slow placements 166 -> 160.8 us
// (-3.2 %), fast ones 149-153 -> 150-157 (+0 ... +2 %); fk J = 22 slow 251-255 -> 245.5 (-3 %), fast 240-243 -> 242-246.5; from_root_dual_quat J = 22 / 52 240 -> 234 /
// 144.8 -> 140.3; quat.to_matrix 195.5 -> 186-191 (-2.4 ... -4.8 %); to_root_dual_quat J = 22 -1 ... -2.5 % -- the levels move together, and most placements are slow
// ones (thirteen boxes: ~60 % of the sets).  The TILE kernels of fk.hip, dq.hip, elementwise.hip and mirror.hip take chunks of kXcdChunk.  The lane-per-frame kernels
// (deep.hip, the ring / order kernels) keep the contiguous eighths of xcd_tile: to_root_dq_deep_kernel on a 128-joint chain read 329 -> 416 us with chunks (their
// partial lines are completed by the NEXT tile of the same XCD, a walk later), mirror's deep kernel +2.7 %; the from_root_positions tile kernels read +-0.5 % either way.
// Launch with grid = 8 * ceil(ntiles / 8); returns -1 for the padding blocks.  chunk <= 0: the contiguous eighths.
#ifndef PM_XCD_CHUNK
#define PM_XCD_CHUNK 0
#endif
__device__ __forceinline__ int64_t xcd_tile_chunked(const int64_t ntiles, const int chunk) {
    const int64_t per_xcd = (ntiles + PM_NXCD - 1) / PM_NXCD;
    const int64_t b = blockIdx.x, x = b % PM_NXCD, i = b / PM_NXCD;
    int64_t t;
    if (chunk <= 0) t = x * per_xcd + i;
    else {
        const int64_t full = per_xcd / chunk * chunk;  // the i's that lie in whole chunks
        // (the last, partial round of chunks: r = per_xcd - full tiles per XCD, packed the same way -- every tile index below 8 per_xcd exactly once)
        t = i < full ? ((i / chunk) * PM_NXCD + x) * chunk + i % chunk : full * PM_NXCD + x * (per_xcd - full) + (i - full);
    }
    return t < ntiles ? t : -1;
}
__device__ __forceinline__ int64_t xcd_tile(const int64_t ntiles) { return xcd_tile_chunked(ntiles, PM_XCD_CHUNK); }
constexpr int kXcdChunk = 32;  // what the tile kernels pass to xcd_tile_chunked

// ---- DPP arithmetic inside a quad (4 consecutive lanes): register-to-register, no LDS ---------------
// quad_perm:[a,b,c,d] = lane i of every quad reads the operand of lane {a,b,c,d}[i] of the same quad.  The
// exchange rides on the DPP operand of the multiply(-add) itself (inline asm: the compiler would emit a
// v_mov_dpp plus the arithmetic instruction), which is what keeps the tree-walk steps short.

// S[c][k] * b_{c xor k} in one instruction (b was loaded from LDS: no VALU -> DPP hazard)
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ float quad_perm_mul(const float b, const float sign) {
    float r;
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[%3,%4,%5,%6] row_mask:0xf bank_mask:0xf"
        : "=&v"(r)
        : "v"(b), "v"(sign), "n"(P0), "n"(P1), "n"(P2), "n"(P3));
    return r;
}

// Component c of the Hamilton product pq (x) b held across a quad (lane c = component c of every operand):
//     q_c = sum_k S[c][k] pq_k b_{c xor k}      (quat.py:337-361; Klein-group structure of the product)
// with pq_k a quad broadcast folded into the DPP operand and sb_k = S[c][k] b_{c xor k} prepared by the
// caller (quad_perm_mul).  pq must not have been written by the VALU instruction right before (s_nop covers it).
__device__ __forceinline__ float quad_qmul(const float pq, const float b0, const float sb1, const float sb2, const float sb3) {
    float q;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(q)
        : "v"(pq), "v"(b0), "v"(sb1), "v"(sb2), "v"(sb3));
    return q;
}

// ---- wave-private linear tile copies ------------------------------------------------------------

// global -> LDS, n floats, contiguous.  VEC: g is 16-byte aligned -> dwordx4 per lane, loads
// batched 4 deep so that up to 4 KiB per wave is in flight before the first LDS write.
template <bool VEC>
__device__ __forceinline__ void tile_load(const float *__restrict__ g, float *lds, int n, int lane) {
    if (VEC) {
        const v4f *g4 = reinterpret_cast<const v4f *>(g);
        v4f *l4 = reinterpret_cast<v4f *>(lds);
        const int n4 = n >> 2;
        int i = lane;
        for (; i + 3 * PM_WAVE < n4; i += 4 * PM_WAVE) {
            v4f a = __builtin_nontemporal_load(g4 + i);
            v4f b = __builtin_nontemporal_load(g4 + i + PM_WAVE);
            v4f c = __builtin_nontemporal_load(g4 + i + 2 * PM_WAVE);
            v4f d = __builtin_nontemporal_load(g4 + i + 3 * PM_WAVE);
            l4[i] = a;
            l4[i + PM_WAVE] = b;
            l4[i + 2 * PM_WAVE] = c;
            l4[i + 3 * PM_WAVE] = d;
        }
        for (; i < n4; i += PM_WAVE) l4[i] = __builtin_nontemporal_load(g4 + i);
        for (int k = (n4 << 2) + lane; k < n; k += PM_WAVE) lds[k] = g[k];
    } else {
        for (int k = lane; k < n; k += PM_WAVE) lds[k] = g[k];
    }
    (void)PM_LDS_OK(lds, (unsigned)n * 4u);  // the whole destination range of the tile
}

// LDS -> global, n floats, contiguous, streaming (write-once) stores.
template <bool VEC>
__device__ __forceinline__ void tile_store(float *__restrict__ g, const float *lds, int n, int lane) {
    if (!PM_LDS_OK(lds, (unsigned)n * 4u)) return;
    if (VEC) {
        v4f *g4 = reinterpret_cast<v4f *>(g);
        const v4f *l4 = reinterpret_cast<const v4f *>(lds);
        const int n4 = n >> 2;
        for (int i = lane; i < n4; i += PM_WAVE) __builtin_nontemporal_store(l4[i], g4 + i);
        for (int k = (n4 << 2) + lane; k < n; k += PM_WAVE) g[k] = lds[k];
    } else {
        for (int k = lane; k < n; k += PM_WAVE) g[k] = lds[k];
    }
}

// Stream n 16-byte records (lane-consecutive, dwordx4 each) from HBM through registers, calling
// fn(e, record, valid) for every element slot e: two batches of 4 loads per lane (8 KiB per wave) are
// always in flight -- batch k+2 is requested before batch k is consumed -- so a tile costs one memory
// latency.  Loads are unconditional inside a batch (a slot past the end re-reads the last record, and fn
// is told so through `valid`): fn should compute unconditionally and guard only its stores, so that the
// four records of a batch are scheduled together with no exec-mask branch between them.
template <bool VEC, class Fn>
__device__ __forceinline__ void for_each_record4(const float *__restrict__ gsrc, const int n, const int lane, Fn &&fn) {
    constexpr int B = 4 * PM_WAVE;
    auto load = [&](const int e0, v4f (&q)[4]) {
        if (e0 >= n) return;  // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
            if (VEC) q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gsrc) + ec);
            else q[u] = v4f{gsrc[4 * ec], gsrc[4 * ec + 1], gsrc[4 * ec + 2], gsrc[4 * ec + 3]};
        }
    };
    auto use = [&](const int e0, const v4f (&q)[4]) {
        if (e0 >= n) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            fn(e, q[u], e < n);
        }
    };
    v4f qa[4], qb[4];
    load(0, qa);
    load(B, qb);
    for (int e0 = 0; e0 < n; e0 += 2 * B) {
        use(e0, qa);
        load(e0 + 2 * B, qa);
        use(e0 + B, qb);
        load(e0 + 3 * B, qb);
    }
}

// fn(slot, valid) over the n element slots of an LDS tile, U slots per lane and trip.  A slot index past
// the end is clamped to the last element and flagged invalid: fn computes unconditionally and guards only
// its stores, so the U elements of a trip are scheduled together with no exec-mask branch between them.
template <int U, class Fn>
__device__ __forceinline__ void for_each_slot(const int n, const int lane, Fn &&fn) {
    for (int e0 = 0; e0 < n; e0 += U * PM_WAVE) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            fn(e < n ? e : n - 1, e < n);
        }
    }
}

// ---- per-lane AoS access to LDS with the widest conflict-free instruction per width -------------

template <int W>
__device__ __forceinline__ void lds_get(const float *base, int idx, float (&v)[W]) {
    const float *p = base + idx * W;
    if (!PM_LDS_OK(p, W * 4u)) {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = 0.0f;
        return;
    }
    if constexpr (W % 4 == 0) {
#pragma unroll
        for (int k = 0; k < W / 4; ++k) {
            v4f t = reinterpret_cast<const v4f *>(p)[k];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else if constexpr (W % 2 == 0) {
#pragma unroll
        for (int k = 0; k < W / 2; ++k) {
            v2f t = reinterpret_cast<const v2f *>(p)[k];
            v[2 * k] = t.x; v[2 * k + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = p[k];
    }
}

template <int W>
__device__ __forceinline__ void lds_put(float *base, int idx, const float (&v)[W]) {
    float *p = base + idx * W;
    if (!PM_LDS_OK(p, W * 4u)) return;
    if constexpr (W % 4 == 0) {
#pragma unroll
        for (int k = 0; k < W / 4; ++k)
            reinterpret_cast<v4f *>(p)[k] = v4f{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
    } else if constexpr (W % 2 == 0) {
#pragma unroll
        for (int k = 0; k < W / 2; ++k) reinterpret_cast<v2f *>(p)[k] = v2f{v[2 * k], v[2 * k + 1]};
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = v[k];
    }
}

// 1-ulp hardware sqrt / reciprocal (v_sqrt_f32, v_rcp_f32): ~8 VALU instructions cheaper per use than
// the correctly-rounded expansions and two orders of magnitude inside the 1e-5 parity budget.
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- rotation arithmetic (fp32, term order of the reference kept; file:line = pymotion/...) -------

// rotations/quat.py:337-361
__device__ __forceinline__ void qmul(const float (&a)[4], const float (&b)[4], float (&o)[4]) {
    // the FMAs are spelled out: left to -ffp-contract the pairing differs from one kernel instance to the next, and the
    // same record would round differently depending on which kernel (tile size, tail, pipeline chunk) it lands in
    o[0] = __builtin_fmaf(-a[3], b[3], __builtin_fmaf(-a[2], b[2], __builtin_fmaf(-a[1], b[1], a[0] * b[0])));
    o[1] = __builtin_fmaf(-a[3], b[2], __builtin_fmaf(a[2], b[3], __builtin_fmaf(b[0], a[1], a[0] * b[1])));
    o[2] = __builtin_fmaf(-a[1], b[3], __builtin_fmaf(a[3], b[1], __builtin_fmaf(b[0], a[2], a[0] * b[2])));
    o[3] = __builtin_fmaf(-a[2], b[1], __builtin_fmaf(a[1], b[2], __builtin_fmaf(b[0], a[3], a[0] * b[3])));
}

// rotations/quat.py:320-334 : t = 2 (qv x v); v' = v + w t + qv x t
__device__ __forceinline__ void qmulvec(const float (&q)[4], const float (&v)[3], float (&o)[3]) {
    const float t0 = 2.0f * __builtin_fmaf(q[2], v[2], -(q[3] * v[1]));
    const float t1 = 2.0f * __builtin_fmaf(q[3], v[0], -(q[1] * v[2]));
    const float t2 = 2.0f * __builtin_fmaf(q[1], v[1], -(q[2] * v[0]));
    o[0] = __builtin_fmaf(q[0], t0, v[0]) + __builtin_fmaf(q[2], t2, -(q[3] * t1));
    o[1] = __builtin_fmaf(q[0], t1, v[1]) + __builtin_fmaf(q[3], t0, -(q[1] * t2));
    o[2] = __builtin_fmaf(q[0], t2, v[2]) + __builtin_fmaf(q[1], t1, -(q[2] * t0));
}

// rotations/quat.py:364-376, 411-423 : q / (|q| + eps)  (eps ADDED TO THE NORM)
__device__ __forceinline__ void qnormalize(const float (&q)[4], float eps, float (&o)[4]) {
    const float n = fsqrt(__builtin_fmaf(q[3], q[3], __builtin_fmaf(q[2], q[2], __builtin_fmaf(q[1], q[1], q[0] * q[0]))));
    const float inv = frcp(n + eps);
    o[0] = q[0] * inv; o[1] = q[1] * inv; o[2] = q[2] * inv; o[3] = q[3] * inv;
}

// rotations/quat.py:276-317
__device__ __forceinline__ void q2m(const float (&q)[4], float (&m)[9]) {
    const float x2 = q[1] + q[1], y2 = q[2] + q[2], z2 = q[3] + q[3];
    const float yy = q[2] * y2, zz = q[3] * z2;
    const float wx = q[0] * x2, wy = q[0] * y2, wz = q[0] * z2;
    m[0] = 1.0f - __builtin_fmaf(q[2], y2, zz); m[1] = __builtin_fmaf(q[1], y2, -wz);       m[2] = __builtin_fmaf(q[1], z2, wy);
    m[3] = __builtin_fmaf(q[1], y2, wz);        m[4] = 1.0f - __builtin_fmaf(q[1], x2, zz); m[5] = __builtin_fmaf(q[2], z2, -wx);
    m[6] = __builtin_fmaf(q[1], z2, -wy);       m[7] = __builtin_fmaf(q[2], z2, wx);        m[8] = 1.0f - __builtin_fmaf(q[1], x2, yy);
}

// rotations/quat.py:85-156 : same predicates and candidates, then normalize(eps = 1e-8).
// Branch-free selects (v_cndmask) keep the wave converged.
__device__ __forceinline__ void m2q(const float (&m)[9], float (&o)[4]) {
    const float r00 = m[0], r01 = m[1], r02 = m[2], r10 = m[3], r11 = m[4], r12 = m[5], r20 = m[6],
                r21 = m[7], r22 = m[8];
    const bool neg = r22 < 0.0f, a = r00 > r11, b = r00 < -r11;
    float c[4];
    c[0] = neg ? (a ? r21 - r12 : r02 - r20) : (b ? r10 - r01 : 1.0f + r00 + r11 + r22);
    c[1] = neg ? (a ? 1.0f + r00 - r11 - r22 : r10 + r01) : (b ? r02 + r20 : r21 - r12);
    c[2] = neg ? (a ? r10 + r01 : 1.0f - r00 + r11 - r22) : (b ? r21 + r12 : r02 - r20);
    c[3] = neg ? (a ? r02 + r20 : r21 + r12) : (b ? 1.0f - r00 - r11 + r22 : r10 - r01);
    qnormalize(c, 1e-8f, o);
}

// rotations/ortho6d.py:67-90 : Gram-Schmidt on the two COLUMNS of x[3][2]  (denominators max(norm, eps): eps = 0 -> NumPy
// path, NaN on a zero column; 1e-12 -> torch twin, zeros -- both live in the float64 twin below).
//
// The reference's projection c2 ~ b - (c1.b) c1 cancels when the columns are close to (anti-)parallel: evaluated in fp32
// what is left of b is rounding noise amplified by 1 / sin(angle) (1.5e-6 already for ordinary records, 2.5e-4 at half a
// degree; round 2 re-did everything below 0.6 degrees in float64 and still read 5e-5 down a 52-joint chain).  The SAME
// frame written without a cancelling step:  c3 = (a x b) / |a x b|,  c2 = c3 x c1  -- identical in exact arithmetic
// ((a x b) x a = b (a.a) - a (a.b)), and each component of a x b is a difference of two products of fp32 INPUTS, which
// Kahan's FMA form (w = rn(a2 b1); (fma(a1, b2, -w)) + fma(-a2, b1, w)) returns to 1.5 ulp of the RESULT however
// much cancels.  Measured against the float64 reference on 2e6 records incl. 2e5 with sin(angle) down to 1e-7:
// max error 2.8e-7 at every angle (+ 8 instructions per record).
// `ill`: records whose answer the reference's eps floors / NaN rules / own rounding noise decide -- a column
// that is zero, non-finite or outside [1e-6, 1e9] in length, or sin(angle) <= 1e-6 (exactly parallel columns included).
// Callers re-do those in float64 with the reference's own sequence of operations (o6d2m_f64 / o6d_chain_f64).
__device__ __forceinline__ float diff_of_products(const float a, const float b, const float c, const float d) {  // a b - c d
    const float w = c * d;
    const float e = __builtin_fmaf(-c, d, w);  // exact: w - c d
    const float f = __builtin_fmaf(a, b, -w);
    return f + e;
}
__device__ __forceinline__ void o6d2m(const float (&x)[6], float (&m)[9], bool &ill) {
    // (every multiply-add is spelled out: with -ffp-contract=fast the compiler fuses a * b + c or not depending on how it packs
    // the unrolled records of a batch (v_pk_mul + v_pk_add vs v_fma), and a record's result would depend on its place in the batch)
    const float a0 = x[0], a1 = x[2], a2 = x[4], b0 = x[1], b1 = x[3], b2 = x[5];
    const float na2 = __builtin_fmaf(a0, a0, __builtin_fmaf(a1, a1, a2 * a2));
    const float ia = __builtin_amdgcn_rsqf(na2);
    const float c10 = a0 * ia, c11 = a1 * ia, c12 = a2 * ia;
    const float n0 = diff_of_products(a1, b2, a2, b1), n1 = diff_of_products(a2, b0, a0, b2), n2 = diff_of_products(a0, b1, a1, b0);
    const float nn2 = __builtin_fmaf(n0, n0, __builtin_fmaf(n1, n1, n2 * n2));
    const float in = __builtin_amdgcn_rsqf(nn2);
    const float c30 = n0 * in, c31 = n1 * in, c32 = n2 * in;
    m[0] = c10; m[1] = __builtin_fmaf(c31, c12, -(c32 * c11)); m[2] = c30;
    m[3] = c11; m[4] = __builtin_fmaf(c32, c10, -(c30 * c12)); m[5] = c31;
    m[6] = c12; m[7] = __builtin_fmaf(c30, c11, -(c31 * c10)); m[8] = c32;
    // (the reference's floors max(norm, eps) only ever decide records that are `ill`)
    const float nb2 = __builtin_fmaf(b0, b0, __builtin_fmaf(b1, b1, b2 * b2));
    ill = !(na2 > 1e-12f && na2 < 1e18f) || !(nb2 > 1e-12f && nb2 < 1e18f) || !(nn2 > 1e-12f * (na2 * nb2));
}

// float64 twins for the rare records o6d2m flags.  o6d_chain_f64 = the whole chain ortho6d.to_quat -> fk's local
// rotation: rotations/ortho6d.py:67-90 (Gram-Schmidt, denominators max(norm, eps)), quat.py:85-156 (from_matrix incl.
// its normalize), then fk's own normalize (skeleton.py:45) and quat.py:276-317; Q = the quaternion to_quat returns.
__device__ __forceinline__ void o6d2m_f64(const float (&x)[6], const float eps_f, double (&r)[9]) {
    const double eps = (double)eps_f;
    const double a0 = x[0], a1 = x[2], a2 = x[4], b0 = x[1], b1 = x[3], b2 = x[5];
    const double ia = 1.0 / fmax(__builtin_sqrt(a0 * a0 + a1 * a1 + a2 * a2), eps);
    const double c10 = a0 * ia, c11 = a1 * ia, c12 = a2 * ia;
    const double d = c10 * b0 + c11 * b1 + c12 * b2;
    double c20 = b0 - d * c10, c21 = b1 - d * c11, c22 = b2 - d * c12;
    const double ib = 1.0 / fmax(__builtin_sqrt(c20 * c20 + c21 * c21 + c22 * c22), eps);
    c20 *= ib; c21 *= ib; c22 *= ib;
    r[0] = c10; r[1] = c20; r[2] = c11 * c22 - c12 * c21;
    r[3] = c11; r[4] = c21; r[5] = c12 * c20 - c10 * c22;
    r[6] = c12; r[7] = c22; r[8] = c10 * c21 - c11 * c20;
}

// Gram-Schmidt of a big-magnitude tile (fk.hip: PREC_F64): the reference's own sequence (ortho6d.py:67-90) in float64, where its
// cancelling projection b - (c1.b) c1 keeps 1e-16 / sin(angle) -- nothing at the angles that are not `ill` -- and one rounding to
// fp32 per matrix entry (3e-8 against the 2.8e-7 of o6d2m: down a chain of 30-unit bones that is the difference between 2 and 16
// ulp of the positions).  Reciprocal square roots: hardware rsq (1 ulp of fp32) + one Newton step in float64 (relative error 5e-15).
// `ill` as in o6d2m (|b - (c1.b) c1|^2 = |a x b|^2 / |a|^2); those records are re-done by the caller with the eps floors.
__device__ __forceinline__ double rsqrt_newton_f64(const double n2) {
    const double y = (double)__builtin_amdgcn_rsqf((float)n2);
    const double e = __builtin_fma(-n2 * y, y, 1.0);
    return __builtin_fma(y * e, 0.5, y);
}
__device__ __forceinline__ void o6d2m_precise(const float (&x)[6], float (&m)[9], bool &ill) {
    const double a0 = x[0], a1 = x[2], a2 = x[4], b0 = x[1], b1 = x[3], b2 = x[5];
    const double na2 = __builtin_fma(a0, a0, __builtin_fma(a1, a1, a2 * a2));
    const double ia = rsqrt_newton_f64(na2);
    const double c10 = a0 * ia, c11 = a1 * ia, c12 = a2 * ia;
    const double d = __builtin_fma(c10, b0, __builtin_fma(c11, b1, c12 * b2));
    double c20 = __builtin_fma(-d, c10, b0), c21 = __builtin_fma(-d, c11, b1), c22 = __builtin_fma(-d, c12, b2);
    const double np2 = __builtin_fma(c20, c20, __builtin_fma(c21, c21, c22 * c22));
    const double ib = rsqrt_newton_f64(np2);
    c20 *= ib; c21 *= ib; c22 *= ib;
    m[0] = (float)c10; m[1] = (float)c20; m[2] = (float)__builtin_fma(c11, c22, -(c12 * c21));
    m[3] = (float)c11; m[4] = (float)c21; m[5] = (float)__builtin_fma(c12, c20, -(c10 * c22));
    m[6] = (float)c12; m[7] = (float)c22; m[8] = (float)__builtin_fma(c10, c21, -(c11 * c20));
    const float na2f = (float)na2, nb2f = (float)__builtin_fma(b0, b0, __builtin_fma(b1, b1, b2 * b2)), np2f = (float)np2;
    ill = !(na2f > 1e-12f && na2f < 1e18f) || !(nb2f > 1e-12f && nb2f < 1e18f) || !(np2f > 1e-12f * nb2f);
}

// quat.py:85-156 in float64, incl. its normalize(eps = 1e-8)
__device__ __forceinline__ void m2q_f64(const double (&m)[9], double (&q)[4]) {
    const double r00 = m[0], r01 = m[1], r02 = m[2], r10 = m[3], r11 = m[4], r12 = m[5], r20 = m[6], r21 = m[7], r22 = m[8];
    const bool neg = r22 < 0.0, a = r00 > r11, b = r00 < -r11;
    double c[4];
    c[0] = neg ? (a ? r21 - r12 : r02 - r20) : (b ? r10 - r01 : 1.0 + r00 + r11 + r22);
    c[1] = neg ? (a ? 1.0 + r00 - r11 - r22 : r10 + r01) : (b ? r02 + r20 : r21 - r12);
    c[2] = neg ? (a ? r10 + r01 : 1.0 - r00 + r11 - r22) : (b ? r21 + r12 : r02 - r20);
    c[3] = neg ? (a ? r02 + r20 : r21 + r12) : (b ? 1.0 - r00 - r11 + r22 : r10 - r01);
    const double iq = 1.0 / (__builtin_sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) + 1e-8);
    q[0] = c[0] * iq; q[1] = c[1] * iq; q[2] = c[2] * iq; q[3] = c[3] * iq;
}

__device__ __forceinline__ void o6d_chain_f64(const float (&x)[6], const float eps_f, float (&L)[9], float (&Q)[4]) {
    double m[9], q[4];
    o6d2m_f64(x, eps_f, m);
    m2q_f64(m, q);
    const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    Q[0] = (float)q0; Q[1] = (float)q1; Q[2] = (float)q2; Q[3] = (float)q3;
    const double in = 1.0 / (__builtin_sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) + 1e-8);
    const double w = q0 * in, xq = q1 * in, y = q2 * in, z = q3 * in;
    const double x2 = xq + xq, y2 = y + y, z2 = z + z;
    const double xx = xq * x2, yy = y * y2, wx = w * x2, xy = xq * y2, yz = y * z2, wy = w * y2, xz = xq * z2, zz = z * z2, wz = w * z2;
    L[0] = (float)(1.0 - (yy + zz)); L[1] = (float)(xy - wz);         L[2] = (float)(xz + wy);
    L[3] = (float)(xy + wz);         L[4] = (float)(1.0 - (xx + zz)); L[5] = (float)(yz - wx);
    L[6] = (float)(xz - wy);         L[7] = (float)(yz + wx);         L[8] = (float)(1.0 - (xx + yy));
}

// rotations/dual_quat.py:12-36 : dq = [qr, 0.5 * (0,t) (x) qr]
__device__ __forceinline__ void rt2dq(const float (&q)[4], const float (&t)[3], float (&dq)[8]) {
    const float tq[4] = {0.0f, t[0], t[1], t[2]};
    float d[4];
    qmul(tq, q, d);
    dq[0] = q[0]; dq[1] = q[1]; dq[2] = q[2]; dq[3] = q[3];
    dq[4] = 0.5f * d[0]; dq[5] = 0.5f * d[1]; dq[6] = 0.5f * d[2]; dq[7] = 0.5f * d[3];
}

// rotations/dual_quat.py:62-83 : t = (2 * qd (x) conj(qr))[1:]
__device__ __forceinline__ void dq2rt(const float (&dq)[8], float (&q)[4], float (&t)[3]) {
    const float cj[4] = {dq[0], -dq[1], -dq[2], -dq[3]};
    const float qd[4] = {dq[4], dq[5], dq[6], dq[7]};
    float d[4];
    qmul(qd, cj, d);
    q[0] = dq[0]; q[1] = dq[1]; q[2] = dq[2]; q[3] = dq[3];
    t[0] = 2.0f * d[1]; t[1] = 2.0f * d[2]; t[2] = 2.0f * d[3];
}

// np.isclose(x, target) with the default rtol 1e-5, atol 1e-8 (NaN is never close)
__device__ __forceinline__ bool isclose_to(float x, float target) { return fabsf(x - target) <= 1e-8f + 1e-5f * fabsf(target); }

// ops/vector.py:4-19 : v / (|v| + eps)
__device__ __forceinline__ void vnormalize(const float (&v)[3], float eps, float (&o)[3]) {
    const float inv = frcp(fsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + eps);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}

// rotations/quat.py:504-576 from_to(v1, v2): rotation taking direction v1 to v2.  `a`, `b` are the directions AFTER
// the optional normalisation (:541-543), so that a caller with a constant v1 can normalise it once.
__device__ __forceinline__ void from_to_unit(const float (&a)[3], const float (&b)[3], float (&o)[4]) {
    const float cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    float ax[3];
    vnormalize(cr, 1e-8f, ax);  // quat.normalize on a 3-vector (:545): same formula
    const float w = fsqrt((1.0f + dot) * 0.5f), s = fsqrt((1.0f - dot) * 0.5f);
    o[0] = w; o[1] = ax[0] * s; o[2] = ax[1] * s; o[3] = ax[2] * s;
    if (isclose_to(dot, 1.0f)) { o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; }  // parallel (:551-552)
    const bool anti = isclose_to(dot, -1.0f);
    if (__builtin_amdgcn_ballot_w64(anti) != 0 && anti) {  // anti-parallel (:554-571), rare: skipped by the whole wave otherwise
        const bool xlike = isclose_to(fabsf(a[0]), 1.0f);
        const float og[3] = {xlike ? 0.0f : 1.0f, xlike ? 1.0f : 0.0f, 0.0f};
        const float c2[3] = {a[1] * og[2] - a[2] * og[1], a[2] * og[0] - a[0] * og[2], a[0] * og[1] - a[1] * og[0]};
        float ax2[3];
        vnormalize(c2, 1e-8f, ax2);
        o[0] = 0.0f; o[1] = ax2[0]; o[2] = ax2[1]; o[3] = ax2[2];
    }
}
// normalize_input = True (the default, :541-543), from the RAW vectors (round 5).  The literal form above computes 1 - dot from a rounded
// dot: between nearly parallel directions (1 - dot ~ 1e-4) one ulp of dot is 2e-6 of sqrt((1 - dot) / 2) -- the reference's own suite
// (test_quat.py:507-660: random vectors of one octant) read 1.3e-6 / 2.2e-6 there.  Here, as in from_root_positions' alignment (ik.hip):
//     cr = u x v (Kahan's difference of products), dt = u . v, N = |u| |v|;   N + dt and N - dt: the one that does not cancel as it
//     stands, the other as |cr|^2 / (that one);
//     the reference's dot is (dt / N) k with k = |u| / (|u| + 1e-8) x |v| / (|v| + 1e-8) (vec.normalize), so N (1 -+ dot) = (N -+ dt) +- dt (1 - k);
//     w = sqrt((1 + dot) / 2), s = sqrt((1 - dot) / 2), axis = normalize(un x vn) with ITS 1e-8; np.isclose(dot, +-1) = a test of
//     N (1 -+ dot) against 1.001e-5 N.
// `ok` false (a zero-length, non-finite or subnormal-length vector): the caller takes the literal form, whose NaN / zero rules are the reference's.
struct FromToTerms { float cr[3], w, s, npr, nmr, tol, kn; bool ok; };
__device__ __forceinline__ FromToTerms from_to_terms(const float (&u)[3], const float (&v)[3]) {
    FromToTerms r;
    r.cr[0] = diff_of_products(u[1], v[2], u[2], v[1]);
    r.cr[1] = diff_of_products(u[2], v[0], u[0], v[2]);
    r.cr[2] = diff_of_products(u[0], v[1], u[1], v[0]);
    const float cr2 = __builtin_fmaf(r.cr[0], r.cr[0], __builtin_fmaf(r.cr[1], r.cr[1], r.cr[2] * r.cr[2]));
    const float dt = __builtin_fmaf(u[0], v[0], __builtin_fmaf(u[1], v[1], u[2] * v[2]));
    const float u2 = __builtin_fmaf(u[0], u[0], __builtin_fmaf(u[1], u[1], u[2] * u[2]));
    const float v2 = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], v[2] * v[2]));
    const float iu = __builtin_amdgcn_rsqf(u2), iv = __builtin_amdgcn_rsqf(v2);
    const float N = (u2 * iu) * (v2 * iv);
    r.ok = N > 1e-30f && N < 1e30f && u2 > 1e-30f && v2 > 1e-30f;  // (NaN fails every test)
    const float big = N + fabsf(dt), small = cr2 * frcp(big);
    const float npd = (dt >= 0.0f) ? big : small, nmd = (dt >= 0.0f) ? small : big;
    const float eu = 1e-8f * iu, ev = 1e-8f * iv, kden = frcp((1.0f + eu) * (1.0f + ev));
    const float dte = dt * (eu + ev + eu * ev) * kden;  // dt (1 - k)
    r.npr = npd - dte; r.nmr = nmd + dte;
    r.tol = 1.001e-5f * N;
    const float i2n = 0.5f * frcp(N);
    r.w = fsqrt(fmaxf(r.npr, 0.0f) * i2n); r.s = fsqrt(fmaxf(r.nmr, 0.0f) * i2n);
    r.kn = kden * frcp(N);  // un x vn = cr kn
    return r;
}
__device__ __forceinline__ void from_to(const float (&v1)[3], const float (&v2)[3], bool normalize_input, float (&o)[4]) {
    float a[3] = {v1[0], v1[1], v1[2]}, b[3] = {v2[0], v2[1], v2[2]};
    if (!normalize_input) { from_to_unit(a, b, o); return; }  // (wave-uniform)
    const FromToTerms t = from_to_terms(v1, v2);
    if (__builtin_amdgcn_ballot_w64(!t.ok) != 0) {  // rare: the literal form for the whole wave, kept by the lanes that need it
        vnormalize(v1, 1e-8f, a); vnormalize(v2, 1e-8f, b);
        from_to_unit(a, b, o);
    }
    if (t.ok) {
        const float cn[3] = {t.cr[0] * t.kn, t.cr[1] * t.kn, t.cr[2] * t.kn};  // un x vn
        float ax[3];
        vnormalize(cn, 1e-8f, ax);  // quat.normalize on a 3-vector (:545)
        o[0] = t.w; o[1] = ax[0] * t.s; o[2] = ax[1] * t.s; o[3] = ax[2] * t.s;
        if (t.nmr <= t.tol) { o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; }  // parallel (:551-552)
    }
    const bool anti = t.ok && t.npr <= t.tol;
    if (__builtin_amdgcn_ballot_w64(anti) != 0 && anti) {  // anti-parallel (:554-571), rare
        vnormalize(v1, 1e-8f, a);
        const bool xlike = isclose_to(fabsf(a[0]), 1.0f);
        const float og[3] = {xlike ? 0.0f : 1.0f, xlike ? 1.0f : 0.0f, 0.0f};
        const float c2[3] = {a[1] * og[2] - a[2] * og[1], a[2] * og[0] - a[0] * og[2], a[0] * og[1] - a[1] * og[0]};
        float ax2[3];
        vnormalize(c2, 1e-8f, ax2);
        o[0] = 0.0f; o[1] = ax2[0]; o[2] = ax2[1]; o[3] = ax2[2];
    }
}

// rotations/quat.py:579-650 from_to_axis(v1, v2, rot_axis): same angle, rotation axis fixed (a, b as above).
__device__ __forceinline__ void from_to_axis_unit(const float (&a)[3], const float (&b)[3], const float (&axis)[3], float (&o)[4]) {
    const float cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    const float w = fsqrt((1.0f + dot) * 0.5f);
    float s = fsqrt((1.0f - dot) * 0.5f);
    const float cda = cr[0] * axis[0] + cr[1] * axis[1] + cr[2] * axis[2];
    s *= (cda > 0.0f) ? 1.0f : ((cda < 0.0f) ? -1.0f : cda);  // np.sign (0 -> 0, NaN -> NaN)
    o[0] = w; o[1] = axis[0] * s; o[2] = axis[1] * s; o[3] = axis[2] * s;
    if (isclose_to(dot, 1.0f)) { o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; }
    if (isclose_to(dot, -1.0f)) { o[0] = 0.0f; o[1] = axis[0]; o[2] = axis[1]; o[3] = axis[2]; }
}
__device__ __forceinline__ void from_to_axis(const float (&v1)[3], const float (&v2)[3], const float (&axis)[3],
                                             bool normalize_input, float (&o)[4]) {
    float a[3] = {v1[0], v1[1], v1[2]}, b[3] = {v2[0], v2[1], v2[2]};
    if (!normalize_input) { from_to_axis_unit(a, b, axis, o); return; }  // (wave-uniform)
    const FromToTerms t = from_to_terms(v1, v2);  // the same (w, s) without the cancellation in 1 - dot, see from_to
    if (__builtin_amdgcn_ballot_w64(!t.ok) != 0) {
        vnormalize(v1, 1e-8f, a); vnormalize(v2, 1e-8f, b);
        from_to_axis_unit(a, b, axis, o);
    }
    if (t.ok) {
        const float cda = t.cr[0] * axis[0] + t.cr[1] * axis[1] + t.cr[2] * axis[2];  // (sign of (un x vn) . axis = sign of cr . axis)
        const float s = t.s * ((cda > 0.0f) ? 1.0f : ((cda < 0.0f) ? -1.0f : cda));   // np.sign (0 -> 0, NaN -> NaN)
        o[0] = t.w; o[1] = axis[0] * s; o[2] = axis[1] * s; o[3] = axis[2] * s;
        if (t.nmr <= t.tol) { o[0] = 1.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; }
        if (t.npr <= t.tol) { o[0] = 0.0f; o[1] = axis[0]; o[2] = axis[1]; o[3] = axis[2]; }
    }
}

template <int V>
struct IntC { static constexpr int value = V; };  // a compile-time int as a function argument (generic lambdas)

// ---- big-magnitude tiles: when fp32 roundings of |p| matter, and the fixed-point translation chain (fk.hip, dq.hip) -------
// The fp32 walks round every position to an ulp of ITS magnitude once per joint and multiply the rotation error by the bone
// lengths.  With bones under a metre and roots within 16 m of the origin that stays inside the 1e-5 parity bar with a factor to
// spare; beyond (centimetre mocap, creatures with metre-long bones, far-away roots) a tile takes more precise rotations and a
// translation chain in 32-bit FIXED POINT: integer adds do not round.  Both tests are ballots on values a tile loads anyway.
constexpr float kBigOffset = 1.0f, kBigRoot = 16.0f;
// |t_j|_1 of a joint-table entry {-, t0, t1, t2} (what it adds to the position bound) and whether it trips the "big" test
__device__ __forceinline__ float const_l1(const v4f c) { return fabsf(c.y) + fabsf(c.z) + fabsf(c.w); }
__device__ __forceinline__ bool const_is_big(const v4f c) {
    return !(fabsf(c.y) < kBigOffset) || !(fabsf(c.z) < kBigOffset) || !(fabsf(c.w) < kBigOffset);  // NaN counts as big
}

// Scale of the fixed-point positions of one tile: every coordinate is bounded by B = max |root| + (bound of |p_j - root|)
// (rotations have unit rows), so with B < 2^e the words p * 2^(30-e) stay below 2^30.
struct FxScale { float S, invS; };

// Wave-wide sum / maximum, the result in a scalar register.  Register-to-register: four DPP steps reduce every row of 16
// lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), four v_readlane pick up the rows.  (The __shfl_xor
// ladder these replace is six dependent ds_bpermute round trips: ~1000 cycles per reduction on a wave that waits for it.)
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
    v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xb1, 0xf, 0xf, true)));
    v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4e, 0xf, 0xf, true)));
    v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true)));  // row_half_mirror
    v = op(v, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true)));  // row_mirror
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum(const float v) {
    return wave_reduce(v, [](const float a, const float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(const float v) {  // NaN sticks (fmaxf would drop it)
    return wave_reduce(v, [](const float a, const float b) { return (b > a || b != b) ? b : a; });
}

// a wave-uniform float that the compiler cannot prove uniform (it came out of a cross-lane reduction): into a scalar register
__device__ __forceinline__ float uniform_f32(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// `tbound` (wave-uniform) bounds |p_j - root| for every joint of the tile, `rmax` = this lane's |root coordinate| (0 for idle
// lanes).  NaN / Inf anywhere make the bound non-finite and the tile stays on the plain fp32 path, which propagates them like the
// reference.  The bound along a path is min(sum over ALL joints of |t_j|_1, depth x max_j |t_j|_1): the first is tight for chains,
// the second for wide trees (a star of 128 thirty-unit bones: 5760 against 45 -- seven bits of the fixed-point word, which at
// coordinates of ~46 made its resolution coarser than fp32's; found by a randomised run of the fuzz tests).
__device__ __forceinline__ bool fx_scale(const float tbound, const float rmax, FxScale &fx) {
    const float B = wave_max(rmax) + tbound;
    const int e = __builtin_amdgcn_frexp_expf(B);  // B = m 2^e, m in [0.5, 1)
    fx.S = __builtin_ldexpf(1.0f, 30 - e);
    fx.invS = __builtin_ldexpf(1.0f, e - 30);
    return B < 1e30f;  // false for NaN / Inf / absurd magnitudes
}

// the homogeneous product's fourth term of a rotation row: + p_parent[r] * 0 (see `poison` at fk.hip's tree_walk).  The empty asm keeps the branch a
// branch: folded into selects it would cost every joint of every tile three more vector instructions.
__device__ __forceinline__ void poison_row(const float pt, float &g0, float &g1, float &g2) {
    asm volatile("");
    g0 = __builtin_fmaf(pt, 0.0f, g0); g1 = __builtin_fmaf(pt, 0.0f, g1); g2 = __builtin_fmaf(pt, 0.0f, g2);
}

// ---- fk's local rotation matrix from a quaternion; PREC_* levels: see fk.hip -----------------------------------
enum { PREC_FAST = 0, PREC_RESID = 1, PREC_F64 = 2, PREC_FX = 4, PREC_DYN = 16, PREC_BIG_RESID = 32 };  // BIG_RESID: see fk.hip

template <int PREC>
__device__ __forceinline__ void local_from_quat(const float (&qi)[4], float (&L)[9]) {
    if constexpr ((PREC & PREC_F64) != 0) {
        const double w = qi[0], x = qi[1], y = qi[2], z = qi[3];
        const double xx = x * x, yy = y * y, zz = z * z;
        const double n2 = __builtin_fma(w, w, xx + (yy + zz));
        const float n2f = (float)n2;
        // 1 / |q|: hardware rsq (1 ulp) + one Newton step in float64 -> relative error ~1e-14
        double yd = (double)__builtin_amdgcn_rsqf(n2f);
        const double e = __builtin_fma(-n2 * yd, yd, 1.0);
        yd = __builtin_fma(yd * e, 0.5, yd);
        // 1 / (|q| + eps) = yd / (1 + eps yd) = yd (1 - eps yd) up to (eps yd)^2 < 1e-12 for |q| > 1e-2
        const double inv = __builtin_fma(-1e-8 * yd, yd, yd);
        const double s = 2.0 * inv * inv;
        const double wz = w * z, wy = w * y, wx = w * x;
        L[0] = (float)__builtin_fma(-s, yy + zz, 1.0); L[1] = (float)(s * __builtin_fma(x, y, -wz)); L[2] = (float)(s * __builtin_fma(x, z, wy));
        L[3] = (float)(s * __builtin_fma(x, y, wz));   L[4] = (float)__builtin_fma(-s, xx + zz, 1.0); L[5] = (float)(s * __builtin_fma(y, z, -wx));
        L[6] = (float)(s * __builtin_fma(x, z, -wy));  L[7] = (float)(s * __builtin_fma(y, z, wx));   L[8] = (float)__builtin_fma(-s, xx + yy, 1.0);
        // tiny or zero quaternions (|q| < 1e-2: eps is no longer a perturbation; zero -> identity, skeleton.py:45): fp32 path
        // non-finite ones too: the reference's NaN PATTERN (e.g. (inf,0,0,0) -> NaN off the diagonal, 1 on it) comes out of its formula
        const bool tiny = !(n2f >= 1e-4f && n2f < 3e38f);
        if (__builtin_amdgcn_ballot_w64(tiny) != 0) {  // wave-uniform, practically never taken
            float q[4], Lf[9];
            qnormalize(qi, 1e-8f, q);
            q2m(q, Lf);
#pragma unroll
            for (int k = 0; k < 9; ++k) L[k] = tiny ? Lf[k] : L[k];
        }
    } else if constexpr ((PREC & PREC_RESID) != 0) {
        float q[4];
        const float n = fsqrt(__builtin_fmaf(qi[3], qi[3], __builtin_fmaf(qi[2], qi[2], __builtin_fmaf(qi[1], qi[1], qi[0] * qi[0]))));
        const float inv = frcp(n + 1e-8f);
        q[0] = qi[0] * inv; q[1] = qi[1] * inv; q[2] = qi[2] * inv; q[3] = qi[3] * inv;
        const float w = q[0], x = q[1], y = q[2], z = q[3];
        // |q^|^2 - 1 as the fp32 arithmetic left it; the reference's q^ has |q^| = |q| / (|q| + eps) = 1 - eps inv, so the
        // scale that reproduces ITS matrix is 2 (1 - eps inv)^2 / |q^|^2 = 2 (1 - r - 2 eps inv) to first order.
        const float r = __builtin_fmaf(w, w, __builtin_fmaf(x, x, __builtin_fmaf(y, y, __builtin_fmaf(z, z, -1.0f))));
        float s = __builtin_fmaf(-2.0f, __builtin_fmaf(2e-8f, inv, r), 2.0f);
        s = (n >= 1e-2f && n < 3e38f) ? s : 2.0f;  // tiny / zero / non-finite quaternions: the reference's formula as it is (L = I + 2 M(q^))
        const float zz = z * z, yy = y * y;
        const float wz = w * z, wy = w * y, wx = w * x;
        L[0] = __builtin_fmaf(-s, __builtin_fmaf(y, y, zz), 1.0f); L[1] = s * __builtin_fmaf(x, y, -wz); L[2] = s * __builtin_fmaf(x, z, wy);
        L[3] = s * __builtin_fmaf(x, y, wz); L[4] = __builtin_fmaf(-s, __builtin_fmaf(x, x, zz), 1.0f); L[5] = s * __builtin_fmaf(y, z, -wx);
        L[6] = s * __builtin_fmaf(x, z, -wy); L[7] = s * __builtin_fmaf(y, z, wx); L[8] = __builtin_fmaf(-s, __builtin_fmaf(x, x, yy), 1.0f);
    } else {
        float q[4];
        qnormalize(qi, 1e-8f, q);
        q2m(q, L);
    }
}



// ---- fk's local rotation from an ortho6d record (fk.hip, fkwide.hip) -----------------------------------------------------------------
// ortho6d record -> local rotation (and, with QOUT, the quaternion the reference would have produced):
// rotations/ortho6d.py:50-64 (6D -> matrix -> quaternion, itself normalised), then fk's own normalise and to_matrix.
template <bool TRANSPOSED>
__device__ __forceinline__ void put_local(float *slot, const float (&L)[9]);

template <bool QOUT, int M>
__device__ __forceinline__ bool local_from_o6d(const float (&xx)[6], const float eps, float (&L)[9], float (&Q)[4]) {
    // The trip matrix -> quaternion -> normalise -> matrix is the identity on an orthonormal matrix up to fp32 rounding (~2e-7, two
    // orders inside the parity budget): the Gram-Schmidt result IS the local rotation, with or without the quaternion output
    // (round 2 went through the quaternion when it was asked for: ~80 more VALU operations per joint and 40 more live registers,
    // 57.9 % against 61.8 %).  The quaternion, when wanted, is from_matrix of it (ortho6d.py:50-64).  None of this holds for what
    // Gram-Schmidt returns on degenerate columns (zeros, NaN, rounding noise): those records are re-done (o6d_redo_ill).
    // M & PREC_F64 (big-magnitude tiles: centimetre mocap, far-away roots): Gram-Schmidt in float64 like the reference's chain, as
    // local_from_quat does for the quaternion source -- the rotation error is multiplied by the bone lengths down the chain.
    bool ill;
    if constexpr ((M & PREC_F64) != 0) o6d2m_precise(xx, L, ill);
    else o6d2m(xx, L, ill);
    if constexpr (QOUT) m2q(L, Q);
    return ill;
}

// Zero / non-finite / (anti-)parallel columns: the reference's answer is decided by its eps floors and NaN rules and, for
// near-parallel columns, by digits fp32 does not have -- such a record's whole chain is re-done in float64, so that both
// variants of the fused kernel equal ortho6d.to_quat -> fk on EVERY input.  Called AFTER the tile's local rotations have
// been parked, on the record's LDS slot: the float64 chain is register-hungry, and inside the conversion loop it would set
// the register budget of the whole kernel (166 VGPRs = three waves per SIMD) for a branch ~1e-4 of the records take.
template <bool QOUT, bool TRANSPOSED>
__device__ __forceinline__ void o6d_redo_ill(const bool ill, const float (&xx)[6], const float eps, float *slot, float *qslot) {
    if (__builtin_amdgcn_ballot_w64(ill) == 0) return;  // wave-uniform
    float Ld[9], Qd[4];
    o6d_chain_f64(xx, eps, Ld, Qd);
    if (ill) {
        put_local<TRANSPOSED>(slot, Ld);
        // qslot: the record's LDS slot (fk_tile) or its place in HBM (fk_pipe_kernel: the only store of that record, the plain
        // conversion skips the records it flags)
        if (QOUT) { qslot[0] = Qd[0]; qslot[1] = Qd[1]; qslot[2] = Qd[2]; qslot[3] = Qd[3]; }
    }
}

// local rotation -> its slot of the image: as is for the three-lane walk, transposed for tree_walk_quad
template <bool TRANSPOSED>
__device__ __forceinline__ void put_local(float *slot, const float (&L)[9]) {
    if (TRANSPOSED) {
        const float T[9] = {L[0], L[3], L[6], L[1], L[4], L[7], L[2], L[5], L[8]};
        lds_put<9>(slot, 0, T);
    } else {
        lds_put<9>(slot, 0, L);
    }
}


// ---- host-side helpers -------------------------------------------------------------------------------

// Tuning aids (frames per wave, tiles per workgroup, ablations) exist ONLY in the -DPM_TUNING build
// (libpmhip_tuning.so, `make tuning`): the production library never reads the environment and carries no
// ablation branch, so a stray PM_* variable cannot change a result or turn a valid call into an error.
#ifdef PM_TUNING
int tune_env(const char *name, int dflt);  // atoi(getenv(name)) or dflt (host.hip)
#define PM_ABLATED(a, bit) (((a).ablate & (bit)) != 0)
#define PM_ABLATED_FLAG(f) ((f) != 0)
#else
constexpr int tune_env(const char *, int dflt) { return dflt; }
#define PM_ABLATED(a, bit) false
#define PM_ABLATED_FLAG(f) false
#endif

struct Parents {  // passed to kernels BY VALUE (kernarg segment -> s_load_dword, uniform index)
    int32_t p[PM_MAX_JOINTS];
};

// After every kernel launch.  Production: the HIP launch error.  PM_DEBUG: also waits for the kernel, surfaces
// asynchronous faults and reads this translation unit's LDS violation record.
#ifdef PM_DEBUG
int debug_after_launch(const char *what, unsigned int *violation_symbol_value);
#define PM_SET_LDS(bytes)                                                                                         \
    do {                                                                                                          \
        const unsigned int pm_lds_b_ = (unsigned int)(bytes);                                                     \
        (void)hipMemcpyToSymbol(HIP_SYMBOL(pm::g_pm_lds_bytes), &pm_lds_b_, sizeof(pm_lds_b_));                   \
    } while (0)
#define PM_AFTER_LAUNCH(what)                                                                                     \
    [&]() -> int {                                                                                                \
        if (int e_ = pm::check_hip(hipGetLastError(), what)) return e_;                                           \
        if (int e_ = pm::check_hip(hipDeviceSynchronize(), what)) return e_;                                      \
        unsigned int v_ = 0, z_ = 0;                                                                              \
        (void)hipMemcpyFromSymbol(&v_, HIP_SYMBOL(pm::g_pm_violation), sizeof(v_));                               \
        if (v_) {                                                                                                 \
            (void)hipMemcpyToSymbol(HIP_SYMBOL(pm::g_pm_violation), &z_, sizeof(z_));                             \
            pm::set_error("%s: LDS access outside the workgroup's allocation (common.hpp / kernel line %u)", what, v_ & 0x7fffffffu); \
            return PM_EHIP;                                                                                       \
        }                                                                                                         \
        return PM_OK;                                                                                             \
    }()
#else
#define PM_SET_LDS(bytes) ((void)0)
#define PM_AFTER_LAUNCH(what) pm::check_hip(hipGetLastError(), what)
#endif

void set_error(const char *fmt, ...);
void set_kernel_name(const char *fmt, ...);  // what the call dispatched to, as rocprofv3 prints it (pm_last_kernel_name)
inline const char *tf(bool b) { return b ? "true" : "false"; }
int check_hip(hipError_t e, const char *what);
int pack_parents(const int32_t *parents, int32_t J, Parents &out);  // validates topology
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Dynamic LDS above 64 KiB needs an explicit opt-in per kernel function.
template <class K>
int allow_lds(K kernel, size_t bytes) {
    PM_SET_LDS(bytes);  // PM_DEBUG: what the helpers check LDS accesses against
    if (bytes <= 64 * 1024) return PM_OK;
    return check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                     "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

constexpr size_t kMaxLds = 160 * 1024;

// ---- dq.hip / mirror.hip: joints list-scheduled onto C chains per frame ---------------------------------------------------
constexpr int kSchedMax = 768;  // bytes of schedule in the kernarg segment (steps x chains)
constexpr int kSchedMaxJoints = 250;
int schedule_chains(const Parents &par, int J, int C, uint8_t *sched, bool root_local);

// ---- deep.hip: lane-per-frame walks for long skeletons -------------------------------------------------------------
constexpr int kDeepM = 8;      // joints per chunk
// The lane-per-frame kernels (deep.hip, mirror_deep_kernel, from_root_positions_order_kernel) put 64 frames in a wave and walk a frame's J
// joints one after the other: a tile takes the same time however few tiles there are (SMPL-H's 52 joints: to_root_dual_quat 33 us,
// mirror 27 us, from_root_positions 38 us from 2^10 to 2^15 frames), where the tile kernels put 8-32 frames in a wave and two or four
// chains on a frame (9-13 / 8-10 / 21-24 us).  Clips of real length -- 2^10...2^15 frames -- are on that floor, so the long-skeleton
// kernels only take a call that has enough joint-frames to fill the chip; the crossovers measured on SMPL-H and chain-like skeletons of
// 24-128 joints (round 4, tools/scratch/*smallF_sweep.py) sit at F J = 2.1-2.6 M (to_root_dual_quat), 5.2-8.4 M (mirror), 2.2-3.7 M
// (from_root_positions).  The bounds-checked build keeps every kernel reachable at test sizes.
#ifdef PM_DEBUG
inline bool lane_per_frame_pays(const int64_t, const int, const int64_t) { return true; }
#else
inline bool lane_per_frame_pays(const int64_t F, const int J, const int64_t min_joint_frames) {
    const int over = tune_env("PM_LPF_MIN_JOINT_FRAMES", -1);  // PM_TUNING build only: the parity tests run these kernels at test sizes with 0
    return F * (int64_t)J >= (over >= 0 ? (int64_t)over : min_joint_frames);
}
#endif
constexpr int64_t kDeepDqMinJointFrames = 2400000, kMirrorDeepMinJointFrames = 6000000, kIkOrderMinJointFrames = 3000000, kFkStreamMinJointFrames = 1000000;
constexpr int kDeepSlots = 6;  // parent states kept in registers for children that do not follow their parent directly (the kernels are LDS-bound at two waves per SIMD: the registers are there)
enum : uint8_t { DEEP_CHAIN = 0xff, DEEP_LOCAL = 0xfe, DEEP_ROOT = 0xfd, DEEP_NONE = 0xff };
struct DeepTopo {                    // by value in the kernarg segment: one s_load_dword per joint (a byte table would be read with
    int32_t code[PM_MAX_JOINTS];     // global_load_ubyte, whose wait is a wait for every prefetch and store in flight): load | save << 8
};                                   // load: where joint j's parent state comes from -- a slot, DEEP_CHAIN (the previous joint), DEEP_LOCAL /
                                     // DEEP_ROOT (none); save: the slot joint j's state is kept in for later children, or DEEP_NONE
int deep_plan(const Parents &par, int J, bool root_is_identity, DeepTopo &t);
// ---- fkwide.hip: fk with a wave per frame and its lanes over the joints (long, wide trees) -------------------------------------
bool try_fk_wide(int src_kind, const float *rot, const float *root_pos, const float *offsets, bool offsets_per_frame, float *pos, float *rotmats,
                 float *quat_out, float eps, int64_t F, int32_t J, int32_t depth, const Parents &par, int ablate, int max_quad_steps_per_joint_x10,
                 hipStream_t s, int &rc);
int fk_wide_plan(const Parents &par, int J, int width, int max_steps, bool dup_idle, uint32_t *jobs);
// ---- dqwide.hip: to_root_dual_quat from a step list in registers, 16 / fpw joints of a frame a step ---------------------------------
bool try_to_root_dq_wide(int fpw, const float *rot, const float *root_pos, const float *offsets, float *dq, int64_t F, int32_t J, int32_t depth,
                         const Parents &par, int ablate, int max_quad_steps_per_joint_x10, hipStream_t s, int &rc);
int dq_wide_words(const Parents &par, int J, int fpw, uint32_t *jobs);      // [16 x 56] step words, dqwide.hip
int mirror_wide_words(const Parents &par, int J, int fpw, uint32_t *jobs);  // [16 x 56] step words, mirror.hip
int launch_to_root_deep(const float *rot, const float *root_pos, const float *offsets, float *dq, int64_t F, int32_t J,
                        const DeepTopo &topo, hipStream_t s);

}  // namespace pm

#define PM_CHECK_ARGS(cond, msg)       \
    do {                               \
        if (!(cond)) {                 \
            pm::set_error("%s", msg);  \
            return PM_EINVAL;          \
        }                              \
    } while (0)
