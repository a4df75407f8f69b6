// dq.hip -- root-centred dual-quaternion encode / decode and from_global_rotations for gfx950.
//   to_root_dual_quat    pymotion/ops/skeleton.py:207-244   (sequential along each parent chain)
//   from_root_dual_quat  pymotion/ops/skeleton.py:173-204   (every joint needs only its parent's INPUT)
//   from_global_rotations pymotion/ops/skeleton.py:64-93    (same gather shape)
// Same wave-private tiling as fk.hip: HBM <-> LDS with contiguous dwordx4, AoS access from LDS.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "dqstep.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------------
// to_root_dual_quat.  The payload is a quaternion + a translation; a QUAD (4 consecutive lanes) owns a
// frame, lane c holding component c of the running root-space quaternion and of the translation written
// as the pure quaternion (0, t).  Products with a distributed left operand use
//     (a (x) b)_c = sum_k S[c][k] a_k b_{c xor k}          (Klein-group structure of the Hamilton product)
// with a_k a quad broadcast and b_{c xor k} a quad permutation, both folded into the DPP operand of the
// multiply-adds; the cross products of the rotate-a-vector formula (quat.py:320-334) are DPP rotations of
// lanes 1..3.  A step is 12 chained VALU instructions (dq_step_math) plus ~20 of bookkeeping for 16
// frames, straight-line: no branch, and every LDS wait is a counted one.
// The LDS image is the tile's OUTPUT (32 J B per frame) plus, per frame, one IDENTITY slot and a 16-byte
// pad (bank-conflict free column access; the copy-out stays a dwordx4 stream):
//   phase A  lane per (frame, joint): quaternion straight from HBM (coalesced dwordx4, loads pipelined)
//            into the first half of the joint's 32-byte output slot;
//   walk     per joint: compose with the parent -- the register chain when parents[j] == j-1, else the
//            parent's finished slot, read one step ahead; joints whose parent is the root stay local
//            (skeleton.py:236-241), which here means "parent = the identity slot", and the root itself
//            (:232) is the same step with the frame's root position as its offset; slot <- (q, t, 0);
//   phase C  lane per (frame, joint): (q, t) -> [q, 0.5 (0,t) (x) q] in place (dual_quat.py:28-36);
//   out      contiguous dwordx4 streaming stores (measured: storing the 32-byte records straight from the
//            phase-C registers, two dwordx4 per lane at a 32-byte stride, is 5 % slower at J = 22).
// Per-joint constants sit in LDS in the form the lanes consume them: {v_c, 2 v_next-next, 2 v_next} per
// lane column, and the effective parent index (one v_readlane per step, 64 joints per VGPR).
// ---------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int to_root_frame_stride(const int J) { return 8 * J + 12; }

struct ToRootArgs {
    const float *rot;       // [F,J,4]
    const float *root_pos;  // [F,3]
    const float *offsets;   // [J,3]
    float *dq;              // [F,J,8]
    int64_t F;
    int32_t J;
    int32_t ablate;  // PM_TUNING build only (env PM_DQ_ABLATE): 1 = skip the walk, 2 = skip phase C; always 0 in production
    int32_t depth;   // edges on the longest root-to-leaf path (bound of the fixed-point translations, see fx_scale)
    Parents parents;
};


// DEEP: the skeleton is at least kDqF64RotMinDepth deep (chosen by the host: a kernel instance of its own, so that the shallow skeletons' code -- the 22-joint
// body, SMPL-H -- is rounds 3-5's to the instruction: the float64 rotation as a branch inside the step cost centimetre-scale data 5 %, as a third loop 14 more
// VGPRs = a wave per SIMD, on metre data too)
template <int FPW, bool VEC, bool DEEP>
__global__ __launch_bounds__(PM_WAVE) void to_root_dq_kernel(const ToRootArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int n = nf * J;
    const int FS = to_root_frame_stride(J);  // J slots + the identity slot + one 16-byte pad (odd FS/4: conflict-free columns)
    float *sDq = smem;                                        // [FPW * FS]
    float *sTab = sDq + FPW * FS;                             // [(J+3) * 12]  joint j, lane column c: {live v_c, 2 v_nn, 2 v_n}
    int *sPar = reinterpret_cast<int *>(sTab + 12 * (J + 3));  // [J+1]  effective parent (J = identity slot); entry J repeats J-1
    const float invJ = 1.0f / (float)J;

    // all global loads first: root position, constants, then the rotations in batches of 4
    // lanes >= 4*FPW shadow lanes 0.. ; frames past a partial tile walk their own (unused) slots: no masking
    const int wl = lane % (4 * FPW);
    const int fq = wl >> 2, c = wl & 3;
    const float rp = (c > 0 && fq < nf) ? a.root_pos[(f0 + fq) * 3 + c - 1] : 0.0f;  // (0, root_pos) component c
    bool tbig = false;                 // a bone of a metre or more (or NaN) somewhere in the table: see kBigOffset
    float tsum = 0.0f, tmx = 0.0f;     // sum / max over the joints of |t_j|_2 (this lane's share; NaN sticks)
    for (int i = lane; i < 4 * (J + 3); i += PM_WAVE) {  // the joint table: offsets in the form each lane column consumes
        const int j = i >> 2, cc = i & 3, jc = j < J ? j : J - 1;
        const float o[3] = {a.offsets[3 * jc], a.offsets[3 * jc + 1], a.offsets[3 * jc + 2]};
        float vc = 0.0f, w1 = 0.0f, w2 = 0.0f;  // column 0 carries the zero scalar part of (0, t)
        if (cc > 0) {
            const int cur = cc - 1, nx = cur == 2 ? 0 : cur + 1, nn = nx == 2 ? 0 : nx + 1;
            vc = o[cur]; w1 = 2.0f * o[nn]; w2 = 2.0f * o[nx];
        }
        sTab[3 * i] = vc; sTab[3 * i + 1] = w1; sTab[3 * i + 2] = w2;
        if (cc == 0 && j < J) {
            const float l1 = fsqrt(__builtin_fmaf(o[0], o[0], __builtin_fmaf(o[1], o[1], o[2] * o[2]))) * 1.000001f;  // the bone's LENGTH (see fx_scale_exact), rounded up
            tbig = tbig || !(fabsf(o[0]) < kBigOffset) || !(fabsf(o[1]) < kBigOffset) || !(fabsf(o[2]) < kBigOffset);
            tsum += l1;
            tmx = (l1 > tmx || l1 != l1) ? l1 : tmx;
        }
    }
    for (int j = lane; j <= J; j += PM_WAVE) {
        // joints hanging off the root stay local (skeleton.py:236-237) = composed with the identity slot;
        // so does the root itself (:232), whose "offset" is the frame's root position
        const int p = a.parents.p[j < J ? j : J - 1];
        sPar[j] = (p == 0) ? J : p;
    }
    const float *gsrc = a.rot + f0 * J * 4;
    auto load_batch = [&](const int e0, v4f (&q)[4]) {
        if (e0 >= n) return;  // wave-uniform; inside a batch the loads are unconditional (clamped): no exec branches
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
            if (VEC) q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gsrc) + ec);
            else q[u] = v4f{gsrc[4 * ec], gsrc[4 * ec + 1], gsrc[4 * ec + 2], gsrc[4 * ec + 3]};
        }
    };
    // The precise step bounds its fixed-point translations by max |root| + the bone lengths down the chain -- true for UNIT rotations only,
    // and this op does not normalise its inputs (skeleton.py:207-244: "inputs are not normalised"): with |q| = 1.05 a rotated offset has
    // grown 2.2x after eight ancestors, past the 2x headroom of the word.  A tile with a quaternion off unit length
    // therefore keeps the fp32 step, whose floats scale with the data like the reference's.
    bool offunit = false;
    auto park_batch = [&](const int e0, const v4f (&q)[4]) {
        if (e0 >= n) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
            const int j = e - f * J;
            const float n2 = __builtin_fmaf(q[u].w, q[u].w, __builtin_fmaf(q[u].z, q[u].z, __builtin_fmaf(q[u].y, q[u].y, q[u].x * q[u].x)));
            offunit = offunit || (fabsf(n2 - 1.0f) >= 1e-3f && n2 < 3e38f);  // (NaN / Inf: the float64 chain propagates them; records past the tile's end repeat its last one)
            if (e < n) *reinterpret_cast<v4f *>(sDq + f * FS + j * 8) = q[u];
        }
    };
    {   // two batches (8 KiB per wave) in flight, batch k+2 requested before batch k is consumed
        constexpr int B = 4 * PM_WAVE;
        v4f qa[4], qb[4];
        load_batch(0, qa);
        load_batch(B, qb);
        for (int e0 = 0; e0 < n; e0 += 2 * B) {
            park_batch(e0, qa);
            load_batch(e0 + 2 * B, qa);
            park_batch(e0 + B, qb);
            load_batch(e0 + 3 * B, qb);
        }
    }
    float *fD = sDq + fq * FS;
    fD[J * 8 + c] = (c == 0) ? 1.0f : 0.0f;  // the identity slot: (1,0,0,0 | 0,0,0,0)
    fD[J * 8 + 4 + c] = 0.0f;
    wave_sync();

    // ---- the walk: four lanes per frame, lane c owns component c of (q_j, (0, t_j)) -----------------------
    // Straight-line steps (no branch, every LDS wait a counted one).  Per step: the parent's (q, t) was
    // read from the image one step ago (`pe*`), unless the parent is the previous joint (register chain);
    // the joint's own quaternion component and its table row were requested two steps ago into the
    // register set the step consumes (A / B ping-pong) and are re-requested for joint j+2 once used.
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f;   // S[c][1]:  - + - +
    const float s2 = (c == 0 || c == 3) ? -1.0f : 1.0f;   // S[c][2]:  - + + -
    const float s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;   // S[c][3]:  - - + +
    const float live = (c == 0) ? 0.0f : 1.0f;            // lane 0 carries the zero scalar part of (0, t)
    const int toff = (c == 0) ? 7 : 3 + c;                // where component c of (0,t) sits in a slot: t0 t1 t2 0
    const float *fDq = fD + c, *fDt = fD + toff;          // parent reads
    float *oq = fD + c, *ot = fD + toff;                  // slot of the current pair's first joint
    const float *tb = sTab + 3 * c;                       // table row of the current pair's first joint
    struct Regs { float b, vc, w1, w2; };
    Regs A = {oq[0], rp, 0.0f, 0.0f};                     // the root: "offset" = root position (skeleton.py:232)
    Regs B = {oq[8], tb[12], tb[13], tb[14]};
    // which arithmetic this tile gets (wave-uniform; see "Big-magnitude tiles" above)
    bool precise = false;
    FxScaleD fx = {1.0, 1.0};
    // (a NaN / Inf QUATERNION needs no special case: it makes the float64 chain of its joint and of every descendant NaN in all
    // four components, and the dual part 0.5 (0,t) (x) q with them, whatever the fixed-point words hold -- the reference's pattern)
    if (__builtin_amdgcn_ballot_w64(tbig || !(fabsf(rp) < kBigRoot)) != 0 && __builtin_amdgcn_ballot_w64(offunit) == 0) {
        const float bsum = wave_sum(tsum), bmax = (float)a.depth * wave_max(tmx);  // (NaN sticks in both)
        precise = fx_scale_for<DEEP>((bmax < bsum) ? bmax : bsum, fabsf(rp), fx);  // false for a non-finite bound: fp32 step
    }
    float gq = 0.0f, gt = 0.0f;                           // previous joint, root space
    double gqd = 0.0;                                     // precise: the same quaternion component in float64
    float peqA = (c == 0) ? 1.0f : 0.0f, petA = 0.0f, peqB = 0.0f, petB = 0.0f;  // the root composes with the identity
    int pelA = 0, pelB = 0;                               // precise: the parent slot's packed residuals
    const float *fDl = fD + 7;  // (read as the floats they are stored as: an int-typed load does not alias the float-typed stores for the compiler, see dw_walk in dqwide.hip)
    int par = J;
    auto step = [&](auto tag, const int j, const int o, const int parn, Regs &S, const float peq, const float pet, const int pel,
                    float &peqn, float &petn, int &peln, const bool may_be_dummy) {
        constexpr bool PRECISE = decltype(tag)::value != 0;
        peqn = fDq[parn * 8];  // parent of joint j+1, if it is not joint j itself (then: a stale value, unused)
        petn = fDt[parn * 8];
        if (PRECISE) peln = __float_as_int(fDl[parn * 8]);
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(S.b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(S.b, s2),
                    sb3 = quad_perm_mul<3, 2, 1, 0>(S.b, s3);
        const bool chain = (par == j - 1);  // wave-uniform
        float q, t;
        if constexpr (PRECISE) {
            const double pqd = chain ? gqd : dq_parent_f64(peq, pel, c);
            const int pti = __float_as_int(chain ? gt : pet);
            int ti, tw;
            gqd = dq_step_precise<DEEP>(pqd, pti, S.b, sb1, sb2, sb3, S.vc, S.w1, S.w2, live, fx.S, c, q, ti, tw);
            gt = __int_as_float(ti);
            t = __int_as_float(tw);
        } else {
            const float pq = chain ? gq : peq, pt = chain ? gt : pet;
            const float s = S.vc + pt;
            dq_step_math(pq, s, S.b, sb1, sb2, sb3, S.w1, S.w2, live, q, t);
            gq = q; gt = t;
        }
        if (!may_be_dummy || j < J) { oq[o] = q; ot[o] = t; }  // slot <- root-space (q, t, 0)
        S.b = oq[o + 16];                                      // joint j+2: its slot still holds the input quaternion
        S.vc = tb[(o >> 3) * 12 + 24]; S.w1 = tb[(o >> 3) * 12 + 25]; S.w2 = tb[(o >> 3) * 12 + 26];
        par = parn;
    };
    auto walk = [&](auto tag) {
        for (int jb = PM_ABLATED(a, 1) ? J : 0; jb < J; jb += PM_WAVE) {
            // effective parents of joints jb+1 .. jb+64 across the lanes: one v_readlane per step
            const int i0 = jb + 1 + lane;
            const int pv = sPar[i0 < J ? i0 : J];
            const int jend = (J - jb) < PM_WAVE ? (J - jb) : PM_WAVE;
            asm volatile("" ::"v"(pv));  // settle the window load here, not as an lgkmcnt(0) inside the loop
            for (int jj = 0; jj < jend; jj += 2) {  // pairs; for odd J the very last step is a dummy that stores nothing
                step(tag, jb + jj, 0, __builtin_amdgcn_readlane(pv, jj), A, peqA, petA, pelA, peqB, petB, pelB, false);
                step(tag, jb + jj + 1, 8, __builtin_amdgcn_readlane(pv, jj + 1), B, peqB, petB, pelB, peqA, petA, pelA, true);
                oq += 16; ot += 16; tb += 24;
            }
        }
    };
    if (precise) walk(IntC<1>{});
    else walk(IntC<0>{});
    wave_sync();
    // phase C, lane per (frame, joint): (q, t) -> [q, 0.5 (0,t) (x) q]  (dual_quat.py:28-36), off the chain
    for_each_slot<2>(PM_ABLATED(a, 2) ? 0 : n, lane, [&](const int e, const bool valid) {
        const int f = (int)(((float)e + 0.5f) * invJ);
        const int j = e - f * J;
        float *slot = sDq + f * FS + j * 8;
        float qt[8], d[8];
        lds_get<8>(slot, 0, qt);
        if (precise) {  // wave-uniform: the translation words are fixed point
#pragma unroll
            for (int k = 4; k < 7; ++k)
                qt[k] = DEEP ? (float)((double)__float_as_int(qt[k]) * fx.invS) : (float)__float_as_int(qt[k]) * (float)fx.invS;
        }
        const float q[4] = {qt[0], qt[1], qt[2], qt[3]}, t[3] = {qt[4], qt[5], qt[6]};
        rt2dq(q, t, d);
        if (valid) *reinterpret_cast<v4f *>(slot + 4) = v4f{d[4], d[5], d[6], d[7]};
    });
    wave_sync();
    // copy-out: dwordx4 i of the tile lives at frame i / 2J, chunk i % 2J of the padded image
    float *gout = a.dq + f0 * J * 8;
    const int n4 = n * 2, J2 = 2 * J;
    const float invJ2 = 1.0f / (float)J2;
    for (int i = lane; i < n4; i += PM_WAVE) {
        const int f = (int)(((float)i + 0.5f) * invJ2);
        const int r = i - f * J2;
        const v4f v = *reinterpret_cast<const v4f *>(sDq + f * FS + r * 4);
        if (VEC) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(gout) + i);
        else { gout[4 * i] = v.x; gout[4 * i + 1] = v.y; gout[4 * i + 2] = v.z; gout[4 * i + 3] = v.w; }
    }
}

template <int FPW>
static int launch_to_root(const ToRootArgs &a, bool vec, hipStream_t s) {
    const size_t lds = ((size_t)FPW * to_root_frame_stride(a.J) + 12 * (a.J + 3) + (a.J + 1)) * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("to_root_dq: grid too large"); return PM_EUNSUPPORTED; }
    const bool deep = a.depth >= kDqF64RotMinDepth;
    set_kernel_name("void pm::to_root_dq_kernel<%d, %s, %s>(pm::ToRootArgs)", FPW, tf(vec), tf(deep));
    auto go = [&](auto k) {
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
        return (int)PM_OK;
    };
    if (int e = vec ? (deep ? go(to_root_dq_kernel<FPW, true, true>) : go(to_root_dq_kernel<FPW, true, false>))
                    : (deep ? go(to_root_dq_kernel<FPW, false, true>) : go(to_root_dq_kernel<FPW, false, false>))) return e;
    return PM_AFTER_LAUNCH("to_root_dq launch");
}

// ---------------------------------------------------------------------------------------------------
// to_root_dual_quat for bigger skeletons: C independent chains per frame.
// The walk above is J serial steps per tile, and from ~28 joints on the LDS image only leaves room for 8 frames per
// wave: half the lanes walk, and the serial part is a third of the kernel (2^18 x 52: 182 us with the walk, 120 without).
// A skeleton is a TREE, though: subtrees (the limbs, the fingers) are independent once their common ancestor is done.
// Here 4 C lanes own a frame -- C quads, each walking its own sequence of joints in lockstep -- and a wave covers
// 16 / C frames.  The host list-schedules the joints onto the C chains (longest remaining path first; a joint may
// follow its parent immediately only on the parent's own chain, where the parent's value is still in registers,
// otherwise two steps later, when the parent's slot has been written AND the look-ahead read was issued after it),
// so the walk is K = max(ceil(J / C), depth, ...) steps instead of J: 27 for the 52-joint SMPL-H tree with two
// chains, 15 with four.  The schedule travels in the kernarg segment (one byte per step and chain) and is expanded
// once per tile into an LDS "program": byte offsets of the joint's slot, its parent's slot and its table row.
// Everything else (image = output tile + identity slot, phase A / C, copy-out, the 12-instruction DPP step) is the
// kernel above.
// ---------------------------------------------------------------------------------------------------
constexpr int kDqWideMinJ = 16;     // to_root_dq_wide_kernel (dqwide.hip) from here on; below, sixteen frames a wave on the one-chain kernel
constexpr int kDeepDqHintMinJ = 20;  // ... and from here on when the caller says the bones are big (pm_to_root_dq_hint_f32)
constexpr int kDeepDqMinJ = 40;  // from here on the lane-per-frame kernels of deep.hip where the topology allows (2^19 frames, chain-like skeleton, deep / scheduled walk: J = 32 159 / 152 us, 40 202 / 206, 48 246 / 256, 56 284 / 309, 64 315 / 383; the 52-joint SMPL-H tree at 2^18: 149 / 146 us on metre data, 148 / 196 us on centimetre data -- the float64 state does not know the difference)
// (kSchedMax, kSchedMaxJoints: common.hpp -- mirror.hip schedules its walk the same way)

struct SchedArgs {
    const float *rot;
    const float *root_pos;
    const float *offsets;
    float *dq;
    int64_t F;
    int32_t J;
    int32_t K;                       // steps
    int32_t depth;                   // edges on the longest root-to-leaf path (fixed-point bound of the precise step)
    int16_t parent[kSchedMaxJoints + 2];
    uint8_t sched[kSchedMax];        // [K][C]: joint index, 255 = idle
};

__host__ __device__ constexpr int sched_frame_stride(const int J) { return 8 * J + 20; }  // J slots + identity + idle slot + pad; (FS / 4) odd

// (second launch bound: five waves per SIMD = at most 96 VGPRs.  Its tiles are 4-13 KB of LDS, so registers bound residency: 110 VGPRs read 210 -> 228 us at
// 2^20 x 22 on metre data, round 3; every instance sits at 96 today, the bound keeps it there)
// (Round 6 tried the schedule fk's wide walks use -- a joint in the very next step after its parent on ANY chain, the parent's slot read after the writes of the
// step before instead of two steps ahead: K shrinks by the depth, but the read's latency lands on every step: J = 22 210 -> 215 us, random trees of 96 / 128
// joints 282 -> 293 / 380 -> 401 us on one box.  Not kept.)
template <int C, bool VEC, bool DEEP>
__global__ __launch_bounds__(PM_WAVE, 5) void to_root_dq_sched_kernel(const SchedArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int FPW = 16 / C;
    const int lane = threadIdx.x;
    const int J = a.J, K = a.K;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t group = xcd_tile_chunked((ntiles + nt - 1) / nt, kXcdChunk);  // a workgroup owns `nt` consecutive tiles: table and program are built once
    if (group < 0) return;
    const int FS = sched_frame_stride(J);
    float *sDq = smem;                                             // [FPW * FS]
    float *sTab = sDq + FPW * FS;                                  // [(J + 2) * 12]  rows J, J+1 (identity, idle) are zero
    v4i *sProg = reinterpret_cast<v4i *>(sTab + 12 * (J + 2));     // [(K + 2) * C]  {own, parent, table row (bytes), on-chain}
    const float invJ = 1.0f / (float)J;

    const int fq = lane / (4 * C), k = (lane >> 2) % C, c = lane & 3;
    // (Staging the raw constants through LDS first, so that the tile pays one memory latency instead of one per table
    // batch, was measured and is SLOWER -- 150 -> 178 us at J = 52: with ~9 resident waves the latencies are hidden anyway and
    // the kernel is bound by instruction issue, which the extra LDS round trip adds to.)
    bool tbig_l = false;               // a bone of a metre or more (or NaN) in the table: see kBigOffset
    for (int i = lane; i < 4 * (J + 2); i += PM_WAVE) {  // the joint table, as in to_root_dq_kernel; row 0 is zero (skeleton.py:227)
        const int j = i >> 2, cc = i & 3;
        float vc = 0.0f, w1 = 0.0f, w2 = 0.0f;
        if (cc > 0 && j > 0 && j < J) {
            const float o[3] = {a.offsets[3 * j], a.offsets[3 * j + 1], a.offsets[3 * j + 2]};
            const int cur = cc - 1, nx = cur == 2 ? 0 : cur + 1, nn = nx == 2 ? 0 : nx + 1;
            vc = o[cur]; w1 = 2.0f * o[nn]; w2 = 2.0f * o[nx];
            tbig_l = tbig_l || !(fabsf(vc) < kBigOffset);
        }
        sTab[3 * i] = vc; sTab[3 * i + 1] = w1; sTab[3 * i + 2] = w2;
    }
    const bool tbig = __builtin_amdgcn_ballot_w64(tbig_l) != 0;  // per workgroup: the table is shared by its tiles
    // bound of |t_j - t_(depth-1 ancestor)| for the fixed-point scale, from the table in LDS: only tiles that take the precise
    // step pay for it (round-3 note: gathering these sums in the loop above for every workgroup cost metre-scale data 4 %)
    auto table_bound = [&]() {
        float tsum_l = 0.0f, tmx_l = 0.0f;  // sum / max over the joints of |t_j|_2 (this lane's share; NaN sticks)
        for (int j = 1 + lane; j < J; j += PM_WAVE) {
            const float o0 = sTab[12 * j + 3], o1 = sTab[12 * j + 6], o2 = sTab[12 * j + 9];
            const float l1 = fsqrt(__builtin_fmaf(o0, o0, __builtin_fmaf(o1, o1, o2 * o2))) * 1.000001f;  // the bone's LENGTH (see fx_scale_exact), rounded up
            tsum_l += l1;
            tmx_l = (l1 > tmx_l || l1 != l1) ? l1 : tmx_l;
        }
        const float bsum = wave_sum(tsum_l), bmax = (float)a.depth * wave_max(tmx_l);  // (NaN sticks in both)
        return (bmax < bsum) ? bmax : bsum;
    };
    for (int i = lane; i < (K + 2) * C; i += PM_WAVE) {  // the program; two idle steps of slack for the look-ahead
        const int st = i / C, kk = i - st * C;
        int j = (st < K) ? a.sched[i] : 255;
        v4i e;
        if (j == 255) {
            e = v4i{(J + 1) * 32, J * 32, (J + 1) * 48, 0};  // idle: the scratch slot, composed with the identity
        } else {
            const int p = a.parent[j];
            const int pe = (j == 0 || p == 0) ? J : p;  // the root and its children compose with the identity (skeleton.py:232-237)
            const int prev = (st > 0) ? a.sched[(st - 1) * C + kk] : 255;
            e = v4i{j * 32, pe * 32, j * 48, (pe != J && prev == pe) ? 1 : 0};
        }
        sProg[i] = e;
    }
  for (int64_t tile = group * nt; tile < ntiles && tile < (group + 1) * nt; ++tile) {
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int n = nf * J;
    const float rp = (c > 0 && fq < nf) ? a.root_pos[(f0 + fq) * 3 + c - 1] : 0.0f;  // (0, root_pos) component c
    const float *gsrc = a.rot + f0 * J * 4;
    auto load_batch = [&](const int e0, v4f (&q)[4]) {
        if (e0 >= n) return;  // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
            if (VEC) q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gsrc) + ec);
            else q[u] = v4f{gsrc[4 * ec], gsrc[4 * ec + 1], gsrc[4 * ec + 2], gsrc[4 * ec + 3]};
        }
    };
    bool offunit = false;  // a quaternion of the tile off unit length: the precise step's fixed-point bound does not hold, see to_root_dq_kernel
    auto park_batch = [&](const int e0, const v4f (&q)[4]) {
        if (e0 >= n) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
            const int j = e - f * J;
            const float n2 = __builtin_fmaf(q[u].w, q[u].w, __builtin_fmaf(q[u].z, q[u].z, __builtin_fmaf(q[u].y, q[u].y, q[u].x * q[u].x)));
            offunit = offunit || (fabsf(n2 - 1.0f) >= 1e-3f && n2 < 3e38f);
            if (e < n) *reinterpret_cast<v4f *>(sDq + f * FS + j * 8) = q[u];
        }
    };
    {
        constexpr int BATCH = 4 * PM_WAVE;
        v4f qa[4], qb[4];
        load_batch(0, qa);
        load_batch(BATCH, qb);
        for (int e0 = 0; e0 < n; e0 += 2 * BATCH) {
            park_batch(e0, qa);
            load_batch(e0 + 2 * BATCH, qa);
            park_batch(e0 + BATCH, qb);
            load_batch(e0 + 3 * BATCH, qb);
        }
    }
    float *fD = sDq + fq * FS;
    if (k == 0) {  // identity slot (1,0,0,0 | 0,0,0,0); the idle slot just has to hold finite words
        fD[J * 8 + c] = (c == 0) ? 1.0f : 0.0f;
        fD[J * 8 + 4 + c] = 0.0f;
        fD[(J + 1) * 8 + c] = 0.0f;
        fD[(J + 1) * 8 + 4 + c] = 0.0f;
    }
    wave_sync();

    // ---- the walk: K steps, every quad on its own joint ---------------------------------------------------
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f;   // S[c][1]:  - + - +
    const float s2 = (c == 0 || c == 3) ? -1.0f : 1.0f;   // S[c][2]:  - + + -
    const float s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;   // S[c][3]:  - - + +
    const float live = (c == 0) ? 0.0f : 1.0f;
    const int toff = (c == 0) ? 7 : 3 + c;                // where component c of (0,t) sits in a slot: t0 t1 t2 0
    const char *bq = reinterpret_cast<const char *>(fD + c), *bt = reinterpret_cast<const char *>(fD + toff);
    const char *btab = reinterpret_cast<const char *>(sTab + 3 * c);
    const v4i *prog = sProg + k;
    // which arithmetic this tile gets (wave-uniform; "Big-magnitude tiles" above)
    bool precise = false;
    FxScaleD fx = {1.0, 1.0};
    if ((tbig || __builtin_amdgcn_ballot_w64(!(fabsf(rp) < kBigRoot)) != 0) && __builtin_amdgcn_ballot_w64(offunit) == 0)
        precise = fx_scale_for<DEEP>(table_bound(), (k == 0) ? fabsf(rp) : 0.0f, fx);  // false for a non-finite bound: fp32 step
    const char *bl = reinterpret_cast<const char *>(fD + 7);  // precise: a slot's packed residuals
    struct In { float b, vc, w1, w2, peq, pet; int pel; v4i e; };
    auto fetch = [&](auto tag, const v4i e, In &x, const bool root_lane) {
        x.e = e;
        x.b = *reinterpret_cast<const float *>(bq + e.x);
        const float *row = reinterpret_cast<const float *>(btab + e.z);
        x.vc = row[0]; x.w1 = row[1]; x.w2 = row[2];
        if (root_lane) x.vc = rp;  // the root's "offset" is the frame's root position (skeleton.py:232)
        x.peq = *reinterpret_cast<const float *>(bq + e.y);
        x.pet = *reinterpret_cast<const float *>(bt + e.y);
        if (decltype(tag)::value != 0) x.pel = *reinterpret_cast<const int *>(bl + e.y);
    };
    float gq = 0.0f, gt = 0.0f;  // what this quad produced in the previous step
    double gqd = 0.0;            // precise: the same quaternion component in float64
    auto step = [&](auto tag, const In &x) {
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(x.b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(x.b, s2),
                    sb3 = quad_perm_mul<3, 2, 1, 0>(x.b, s3);
        const bool chain = x.e.w != 0;
        float q, t;
        if constexpr (decltype(tag)::value != 0) {
            const double pqd = chain ? gqd : dq_parent_f64(x.peq, x.pel, c);
            const int pti = __float_as_int(chain ? gt : x.pet);
            int ti, tw;
            gqd = dq_step_precise<DEEP>(pqd, pti, x.b, sb1, sb2, sb3, x.vc, x.w1, x.w2, live, fx.S, c, q, ti, tw);
            gt = __int_as_float(ti);
            t = __int_as_float(tw);
        } else {
            const float pq = chain ? gq : x.peq, pt = chain ? gt : x.pet;
            const float s = x.vc + pt;
            dq_step_math(pq, s, x.b, sb1, sb2, sb3, x.w1, x.w2, live, q, t);
            gq = q; gt = t;
        }
        *reinterpret_cast<float *>(const_cast<char *>(bq) + x.e.x) = q;
        *reinterpret_cast<float *>(const_cast<char *>(bt) + x.e.x) = t;
    };
    // The root (joint 0) is always the first entry of chain 0 (the scheduler puts it there).
    auto walk = [&](auto tag) {
        if constexpr (decltype(tag)::value != 0) {
            // precise: one step at a time, every operand fetched where it is used (quaternion chain first, then the table row
            // and the translation).  The two-steps-ahead ping-pong of the fp32 walk, in float64, sets the register budget of the
            // WHOLE kernel: 110 VGPRs = four waves per SIMD instead of five, 210 -> 228 us on metre-scale data that never
            // takes this path; so does anything over 96 here.
            int gti = 0;
            v4i e = prog[0];
            float b = *reinterpret_cast<const float *>(bq + e.x);
#pragma clang loop unroll(disable)
            for (int st = 0; st < K; ++st) {
                const v4i en = prog[(st + 1) * C];  // (the program has two idle steps of slack)
                const float sb1 = quad_perm_mul<1, 0, 3, 2>(b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(b, s2), sb3 = quad_perm_mul<3, 2, 1, 0>(b, s3);
                const bool chain = e.w != 0;
                double pqd = gqd;
                if (!chain) pqd = dq_parent_f64(*reinterpret_cast<const float *>(bq + e.y), *reinterpret_cast<const int *>(bl + e.y), c);
                double qd = quad_qmul_f64(pqd, b, sb1, sb2, sb3);
                asm volatile("" : "+v"(qd), "+v"(pqd));  // the quaternion part is done before the translation's operands are fetched
                const float bn = *reinterpret_cast<const float *>(bq + en.x);  // next joint's input quaternion: its slot is untouched until its own step
                const float *row = reinterpret_cast<const float *>(btab + e.z);
                const float vc = (st == 0 && k == 0) ? rp : row[0];  // the root's "offset" is the frame's root position (skeleton.py:232)
                const int pti = chain ? gti : *reinterpret_cast<const int *>(bt + e.y);
                int ti;
                if constexpr (DEEP) ti = pti + (int)__builtin_rint(__builtin_fma((double)live, dq_step_rot_f64(pqd, row[1], row[2]), (double)vc) * fx.S);
                else ti = pti + (int)__builtin_rintf(__builtin_fmaf(live, dq_step_rot((float)pqd, row[1], row[2]), vc) * (float)fx.S);
                const float qh = (float)qd;
                *reinterpret_cast<float *>(const_cast<char *>(bq) + e.x) = qh;
                *reinterpret_cast<int *>(const_cast<char *>(bt) + e.x) = dq_pack_residual(qd, qh, ti, c);
                gqd = qd; gti = ti;
                e = en; b = bn;
            }
            return;
        }
        In A, B;
        fetch(tag, prog[0], A, k == 0);
        v4i en = prog[C];
        for (int st = 0; st < K; st += 2) {
            // operands of step st+1 are requested before step st computes: a parent finished at step st-1 or earlier is in its
            // slot by now (in-order DS), one finished at step st is this quad's own register chain (the scheduler guarantees it)
            fetch(tag, en, B, false);
            en = prog[(st + 2) * C];
            step(tag, A);
            if (st + 1 >= K) break;
            fetch(tag, en, A, false);
            en = prog[(st + 3) * C];
            step(tag, B);
        }
    };
    if (precise) walk(IntC<1>{});
    else walk(IntC<0>{});
    wave_sync();
    // phase C, lane per (frame, joint): (q, t) -> [q, 0.5 (0,t) (x) q]  (dual_quat.py:28-36), off the chain
    for_each_slot<2>(n, lane, [&](const int e, const bool valid) {
        const int f = (int)(((float)e + 0.5f) * invJ);
        const int j = e - f * J;
        float *slot = sDq + f * FS + j * 8;
        float qt[8], d[8];
        lds_get<8>(slot, 0, qt);
        if (precise) {  // wave-uniform: the translation words are fixed point
#pragma unroll
            for (int kk = 4; kk < 7; ++kk)
                qt[kk] = DEEP ? (float)((double)__float_as_int(qt[kk]) * fx.invS) : (float)__float_as_int(qt[kk]) * (float)fx.invS;
        }
        const float q[4] = {qt[0], qt[1], qt[2], qt[3]}, t[3] = {qt[4], qt[5], qt[6]};
        rt2dq(q, t, d);
        if (valid) *reinterpret_cast<v4f *>(slot + 4) = v4f{d[4], d[5], d[6], d[7]};
    });
    wave_sync();
    float *gout = a.dq + f0 * J * 8;
    const int n4 = n * 2, J2 = 2 * J;
    const float invJ2 = 1.0f / (float)J2;
    for (int i = lane; i < n4; i += PM_WAVE) {
        const int f = (int)(((float)i + 0.5f) * invJ2);
        const int r = i - f * J2;
        const v4f v = *reinterpret_cast<const v4f *>(sDq + f * FS + r * 4);
        if (VEC) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(gout) + i);
        else { gout[4 * i] = v.x; gout[4 * i + 1] = v.y; gout[4 * i + 2] = v.z; gout[4 * i + 3] = v.w; }
    }
    wave_sync();  // the image is reused by the next tile
  }
}

// image + joint table + program
static size_t sched_lds_bytes(const int J, const int K, const int C) {
    return ((size_t)(16 / C) * sched_frame_stride(J) + 12 * (J + 2)) * sizeof(float) + (size_t)(K + 2) * C * 16;
}

// List scheduling of the joints onto C chains (see above).  Returns the number of steps K, or 0 if the schedule does
// not fit the kernarg table.  sched[st * C + k] = joint or 255.  root_local: to_root_dual_quat's convention -- joints hanging
// off the root do not wait for it (they stay local); mirror's world rotations do (root_local = false).
int schedule_chains(const Parents &par, const int J, const int C, uint8_t *sched, const bool root_local) {
    int height[PM_MAX_JOINTS], done_step[PM_MAX_JOINTS], done_chain[PM_MAX_JOINTS];
    for (int j = 0; j < J; ++j) { height[j] = 1; done_step[j] = -1; done_chain[j] = -1; }
    for (int j = J - 1; j >= 1; --j) {
        const int p = par.p[j];
        if ((p != 0 || !root_local) && height[p] < height[j] + 1) height[p] = height[j] + 1;
    }
    int left = J, K = 0;
    for (int st = 0; left > 0; ++st) {
        if ((st + 1) * C > kSchedMax) return 0;
        for (int k = 0; k < C; ++k) {
            int best = -1, best_on_chain = 0;
            if (st == 0 && k == 0) {
                best = 0;  // the root opens chain 0: the kernel hands it the frame's root position there
            } else {
                for (int j = 1; j < J; ++j) {
                    if (done_step[j] >= 0) continue;
                    const int p = par.p[j];
                    int on_chain = 0;
                    if (p != 0 || !root_local) {
                        if (done_step[p] < 0 || done_step[p] == st) continue;             // parent not done (or done in this very step)
                        if (done_step[p] == st - 1) { if (done_chain[p] != k) continue; on_chain = 1; }  // only on the parent's chain
                    }
                    // a continuation of this chain first (nobody else can take it now), then the longest remaining path
                    if (best < 0 || on_chain > best_on_chain || (on_chain == best_on_chain && height[j] > height[best])) { best = j; best_on_chain = on_chain; }
                }
            }
            sched[st * C + k] = (uint8_t)(best < 0 ? 255 : best);
            if (best >= 0) { done_step[best] = st; done_chain[best] = k; --left; }
        }
        K = st + 1;
    }
    return K;
}

constexpr int kSchedEightMinJ = 104, kSchedSixteenMinJ = 192;  // to_root_dual_quat: eight / sixteen chains per frame from these joint counts on (see to_root_dq_impl)
template <int C>
static int launch_to_root_sched(const SchedArgs &a, bool vec, hipStream_t s) {
    constexpr int FPW = 16 / C;
    const size_t lds = sched_lds_bytes(a.J, a.K, C);
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    // tiles per workgroup: the joint table and the program cost ~7 % of a tile's instructions; big batches share them
    // (measured at 2^18 frames, 1 / 2 / 4 / 8 tiles: J = 52 149 / 139 / 143 / 150 us, J = 128 507 / 441 / 409 / 401 us)
    // (eight / sixteen chains -- two / one frame a tile -- want many more: the table and the program are rebuilt per workgroup; J = 128 with eight
    // chains 481 us at two tiles, 382 at eight; J = 200 with sixteen 1058 / 718 / 681 us at 2 / 16 / 64)
    int nt = ntiles >= 16384 ? (C == 4 ? 4 : (C == 8 ? 8 : (C == 16 ? 32 : 2))) : 1;
    nt = tune_env("PM_DQ_NT", nt);  // PM_TUNING build only
    if (nt < 1) nt = 1;
    const int64_t ngroups = (ntiles + nt - 1) / nt;
    const int64_t grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("to_root_dq: grid too large"); return PM_EUNSUPPORTED; }
    const bool deep = a.depth >= kDqF64RotMinDepth;
    set_kernel_name("void pm::to_root_dq_sched_kernel<%d, %s, %s>(pm::SchedArgs, int)", C, tf(vec), tf(deep));
    auto go = [&](auto kf) {
        if (int e = allow_lds(kf, lds)) return e;
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
        return (int)PM_OK;
    };
    if (int e = vec ? (deep ? go(to_root_dq_sched_kernel<C, true, true>) : go(to_root_dq_sched_kernel<C, true, false>))
                    : (deep ? go(to_root_dq_sched_kernel<C, false, true>) : go(to_root_dq_sched_kernel<C, false, false>))) return e;
    return PM_AFTER_LAUNCH("to_root_dq launch");
}

// ---------------------------------------------------------------------------------------------------
// Gather-parent kernels: one lane per (frame, joint); every lane reads its own and its parent's
// INPUT record from the LDS tile -- no dependency chain.
//   MODE 0  from_root_dual_quat: in dq[8] -> out trans[3], rot[4]
//   MODE 1  from_global_rotations: in q[4] -> out q[4]
// ---------------------------------------------------------------------------------------------------
struct GatherArgs {
    const float *in;
    float *out0;  // trans (MODE 0) / local quats (MODE 1)
    float *out1;  // rot (MODE 0)
    int64_t F;
    int32_t J;
    Parents parents;
};

template <int MODE>
constexpr int gather_in_w() { return MODE == 0 ? 8 : 4; }
template <int MODE>
constexpr int gather_lds_w() { return gather_in_w<MODE>(); }  // only the input tile is staged; outputs leave from registers

template <int MODE, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void gather_parent_kernel(const GatherArgs a, const int fpw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IW = gather_in_w<MODE>();
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + fpw - 1) / fpw;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t f0 = tile * fpw;
    const int nf = (int)((a.F - f0) < fpw ? (a.F - f0) : fpw);
    const int FJ = fpw * J;
    float *sIn = smem;             // [fpw*J*IW]

    tile_load<VEC>(a.in + f0 * J * IW, sIn, nf * J * IW, lane);
    int *sPar = reinterpret_cast<int *>(sIn + FJ * IW);  // [J] parents, staged once
    // Outputs are one record per lane with consecutive lanes on consecutive records: a dwordx4 (rotation) or
    // dwordx3 (translation) store per lane is already a contiguous, fully coalesced stream -- no LDS staging.
    float *g0 = a.out0 + f0 * J * (MODE == 0 ? 3 : 4);
    float *g1 = (MODE == 0) ? a.out1 + f0 * J * 4 : nullptr;
    auto store4 = [&](float *g, const int e, const float (&v)[4]) {
        if (VEC) __builtin_nontemporal_store(v4f{v[0], v[1], v[2], v[3]}, reinterpret_cast<v4f *>(g) + e);
        else { g[4 * e] = v[0]; g[4 * e + 1] = v[1]; g[4 * e + 2] = v[2]; g[4 * e + 3] = v[3]; }
    };
    for (int j = lane; j < J; j += PM_WAVE) sPar[j] = (j == 0) ? 0 : a.parents.p[j];
    wave_sync();
    const int n = nf * J;
    const float invJ = 1.0f / (float)J;
    // Branch-free per element: every lane composes with its parent's record and a joint that stays as it is
    // (the root; in MODE 0 also the root's children, skeleton.py:194-203) selects its own value afterwards
    // -- no lane divergence, and two elements per trip in flight.
    for_each_slot<2>(n, lane, [&](const int e, const bool valid) {
        const int f = (int)(((float)e + 0.5f) * invJ);  // e / J without an integer divide (exact for e < 2^22)
        const int j = e - f * J;
        const int par = sPar[j];
        if constexpr (MODE == 0) {
            float d[8], q[4], t[3], pd[8], pq[4], pt[3], qq[4], tt[3];
            lds_get<8>(sIn, e, d);
            lds_get<8>(sIn, f * J + par, pd);
            dq2rt(d, q, t);  // dual_quat.py:75-83
            dq2rt(pd, pq, pt);
            const float inv[4] = {pq[0], -pq[1], -pq[2], -pq[3]};
            const float dv[3] = {t[0] - pt[0], t[1] - pt[1], t[2] - pt[2]};
            qmulvec(inv, dv, tt);
            qmul(inv, q, qq);
            const bool keep = (j == 0) || (par == 0);  // parent still in root space otherwise
#pragma unroll
            for (int k = 0; k < 3; ++k) tt[k] = keep ? t[k] : tt[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) qq[k] = keep ? q[k] : qq[k];
            if (valid) {
                __builtin_nontemporal_store(v3f_a4{tt[0], tt[1], tt[2]}, reinterpret_cast<v3f_a4 *>(g0 + 3 * e));
                store4(g1, e, qq);
            }
        } else {
            float g[4], pg[4], o[4];
            lds_get<4>(sIn, e, g);
            lds_get<4>(sIn, f * J + par, pg);
            const float inv[4] = {pg[0], -pg[1], -pg[2], -pg[3]};
            qmul(inv, g, o);  // skeleton.py:85-91 : conj(global_parent) (x) global_child
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (j == 0) ? g[k] : o[k];
            if (valid) store4(g0, e, o);
        }
    });
}

template <int MODE>
static int launch_gather(const GatherArgs &a, bool vec, hipStream_t s) {
    // Frames per wave: a multiple of 4 (16-byte tile bases) giving ~160-210 (frame,joint) items per wave.
    // Small tiles win here -- there is no chain to amortise and many resident waves hide the load latency
    // (measured at 2^20 x 22: 8/12/16/20 frames per wave -> 250/266/312/314 us; at 2^18 x 52: 4/8 -> 174/192).
    const size_t per_frame = (size_t)a.J * gather_lds_w<MODE>() * sizeof(float);
    const size_t extra = (size_t)a.J * sizeof(int);  // parents table
    int fpw = ((160 + a.J - 1) / a.J + 3) & ~3;
    if (fpw < 4) fpw = 4;
    if (const int v = tune_env("PM_GATHER_FPW", 0); v >= 4) fpw = v & ~3;  // PM_TUNING build only
    while (fpw > 4 && fpw * per_frame + extra > kMaxLds / 4) fpw -= 4;
    if (fpw * per_frame + extra > kMaxLds) { set_error("gather: J too large for LDS"); return PM_EUNSUPPORTED; }
    const size_t lds = fpw * per_frame + extra;
    const int64_t ntiles = (a.F + fpw - 1) / fpw;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("gather: grid too large"); return PM_EUNSUPPORTED; }
    set_kernel_name("void pm::gather_parent_kernel<%d, %s>(pm::GatherArgs, int)", MODE, tf(vec));
    if (vec) {
        auto k = gather_parent_kernel<MODE, true>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, fpw);
    } else {
        auto k = gather_parent_kernel<MODE, false>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, fpw);
    }
    return PM_AFTER_LAUNCH("gather launch");
}

}  // namespace pm

// offsets_abs_max: max |offsets[j][k]| if the caller knows it on the host (< 0 or NaN: unknown), see pm_to_root_dq_hint_f32
static int to_root_dq_impl(const float *rot, const float *root_pos, const int32_t *parents, const float *offsets, int64_t F, int32_t J,
                           float *dq, const float offsets_abs_max, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "to_root_dq: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(rot && root_pos && parents && offsets && dq, "to_root_dq: null pointer");
    ToRootArgs a;
    a.rot = rot; a.root_pos = root_pos; a.offsets = offsets; a.dq = dq; a.F = F; a.J = J;
    a.ablate = tune_env("PM_DQ_ABLATE", 0);
    if (int e = pack_parents(parents, J, a.parents)) return e;
    {
        int dep[PM_MAX_JOINTS];
        a.depth = 0;
        for (int32_t j = 0; j < J; ++j) {
            dep[j] = (j == 0) ? 0 : dep[a.parents.p[j]] + 1;
            if (dep[j] > a.depth) a.depth = dep[j];
        }
    }
    const bool vec = aligned16(rot) && aligned16(dq);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Long skeletons: one lane per frame, the joints streamed through LDS in chunks (deep.hip) -- if the topology's open
    // branch points fit its register slots and the call has the joint-frames to fill the chip (lane_per_frame_pays, common.hpp).
    // PM_DQ_DEEP (PM_TUNING build only): 0 never, 1 whenever the topology fits.
    // A caller that knows the bones are big (a centimetre-scale BVH skeleton: any |offset| >= kBigOffset makes EVERY tile of the tile
    // kernels take the precise step) gets the lane-per-frame kernel from kDeepDqHintMinJ joints on: its float64 state costs the same
    // at every magnitude (2^20 x 22: 235 us against 259 us for the precise step -- and 205 us for the fp32 step on metre data, which
    // is why the raw ABI, which cannot see the scale without reading device memory, keeps the per-tile test below 40 joints).
    const bool big_bones = offsets_abs_max >= kBigOffset && offsets_abs_max < 3e38f;
    // (round 6) The step-list kernel of dqwide.hip -- 16 / fpw joints of a frame a step, the list in registers, nothing in LDS but the image -- is the
    // faster shape on metre-scale data at every joint count from kDqWideMinJ on and whatever the tree (profiles/r06_dq_wide_sweep.txt: 22-joint body 212 -> 200 us,
    // SMPL-H 273 -> 233, random trees of 128 / 250 / 512 joints 54 / 39 / 7.5 % of the HBM spec -> 70 / 64 / 68), and on centimetre-scale data wherever the
    // tile kernels of this file ran (22 joints 266 -> 227 us).  What it does not beat is the lane-per-frame kernels' float64 state on centimetre-scale data
    // of skeletons deep enough for the float64 bone rotation (kDqF64RotMinDepth: 64-joint humanoid 321 us at every scale there, 274 / 394 here), and the raw
    // ABI cannot see the scale.  So: a caller who says the bones are BIG (pm_to_root_dq_hint_f32: the front doors do) gets the lane-per-frame kernels first where
    // they take the call (22 joints 228 us there against 242-256 here, SMPL-H 145 against 152-158) and this kernel when they decline; everybody else -- small bones
    // by the hint, or no hint at all -- gets this kernel first: without a hint a deep skeleton on centimetre-scale data is 10-25 % slower than the lane-per-frame
    // kernels would have been, one on metre-scale data 15-35 % faster (chain-like skeletons of 40-128 joints 60-63 % of the HBM spec there, 72-76 here), and
    // include/pmhip.h tells a C caller with big bones to say so.  PM_DQ_WIDE (PM_TUNING build only): 0 never, 1 / 2 / 4 / 8 force that many frames a wave.
    const int wide_env = tune_env("PM_DQ_WIDE", -1);
    auto try_wide = [&](int &rc) {
        if (!vec || wide_env == 0 || (wide_env < 0 && J < kDqWideMinJ)) return false;
        // (PM_TUNING build only: a test that forces one of the other kernels gets it)
        if (wide_env < 0 && (tune_env("PM_DQ_CHAINS", -1) >= 0 || tune_env("PM_DQ_FPW", 0) > 0 || tune_env("PM_DQ_DEEP", -1) == 1)) return false;
        if (wide_env > 0) return try_to_root_dq_wide(wide_env, rot, root_pos, offsets, dq, F, J, a.depth, a.parents, a.ablate, 0, s, rc);
        // frames a wave (same-box sweeps, profiles/r06_dq_wide_sweep.txt): eight up to 32 joints (22 joints 193 us against 237 with four, on centimetre-scale
        // data 256 against 330), four up to 128 (SMPL-H 229 against 240 with eight, 64 joints 263 against 293, 128 joints 532 against 547 with two), one beyond
        // (250 joints 546 against 558 with two); a narrow tree -- under a third of its quad-steps busy, or more steps than the list holds -- takes more frames
        // and fewer joints a step
        int fpw = J <= 32 ? 8 : (J <= 128 ? 4 : 1);
        // (first a width at which at least two thirds of the quad-steps are busy -- a 64-joint chain-like skeleton 274 us at eight frames a wave against 295 at four, a
        // 128-joint humanoid 574 at two against 604 at one --, then any at which a third are)
        for (const int bound : {15, 30})
            for (int w = fpw; w <= 8; w *= 2)
                if (try_to_root_dq_wide(w, rot, root_pos, offsets, dq, F, J, a.depth, a.parents, a.ablate, bound, s, rc)) return true;
        return false;
    };
    if (!big_bones || wide_env > 0) {
        int rc = PM_OK;
        if (try_wide(rc)) return rc;
    }
    if (const int deep = tune_env("PM_DQ_DEEP", -1); vec && deep != 0 && (deep == 1 || ((J >= kDeepDqMinJ || (big_bones && J >= kDeepDqHintMinJ)) && lane_per_frame_pays(F, J, kDeepDqMinJointFrames)))) {
        DeepTopo topo;
        if (deep_plan(a.parents, J, true, topo) >= 0) return launch_to_root_deep(rot, root_pos, offsets, dq, F, J, topo, s);
    }
    {
        int rc = PM_OK;
        if (try_wide(rc)) return rc;
    }
    const size_t per_frame = (size_t)to_root_frame_stride(J) * sizeof(float), fixed = (13 * (size_t)J + 37) * sizeof(float) + 256;
    int pick = (7 * (16 * per_frame + fixed) <= kMaxLds) ? 16 : 8;  // 4 lanes per frame; keep >= 7 waves per CU if possible
    // From 20 joints on: several chains per frame if the tree is wide enough for the shorter walk to pay
    // (cost of the walk per frame ~ steps x chains / 16; a pure chain stays on the one-chain kernel).  Measured at 2^20 / 2^18
    // frames, one chain (16 frames per wave) against the scheduled walk: J = 16: 142 / 158 us, 20: 195 / 191, 22: 217 / 211,
    // 24: 238 / 225, 26: 73 / 69, 32: 94 / 87, 36: 115 / 96; four chains only pay from ~36 joints (J = 28: 71 us with two, 76-79 with four).
    int chains = tune_env("PM_DQ_CHAINS", -1);  // PM_TUNING build only: 0 = the one-chain kernel, 2 / 4 / 8 / 16
    if ((chains < 0 ? (pick != 16 || J >= tune_env("PM_DQ_SCHED_MINJ", 20)) : (chains == 2 || chains == 4 || chains == 8 || chains == 16)) && J <= kSchedMaxJoints) {
        SchedArgs sa;
        int K2 = 0, K4 = 0, K8 = 0, K16 = 0;
        uint8_t s2[kSchedMax], s4[kSchedMax], s8[kSchedMax], s16[kSchedMax];
        if (chains < 0 || chains == 2) K2 = schedule_chains(a.parents, J, 2, s2, true);
        if (chains < 0 || chains == 4) K4 = schedule_chains(a.parents, J, 4, s4, true);
        // (round 5) EIGHT / SIXTEEN chains -- two / one frame a wave -- for long WIDE trees, with many tiles per workgroup (launch_to_root_sched):
        // random trees, 2^18 frames, % of the HBM spec, four / eight / sixteen chains: J = 112 50.4 / 52.6 / 40.0, 128 46.8 / 51.0 / 44.9, 160 38.9 /
        // 44.6 / 42.8, 200 31.8 / 42.4 / 44.4, 250 26.0 / 33.5 / 39.4 (centimetre data 31.3 / 35.8 / 28.7 at 128, 17.1 / 21.6 / 25.9 at 250;
        // profiles/r05_to_root_dq_chains16.txt); only when the wider schedule is not much emptier (a deep tree's is: K does not shrink)
        if ((chains < 0 && J >= kSchedEightMinJ) || chains == 8) K8 = schedule_chains(a.parents, J, 8, s8, true);
        if ((chains < 0 && J >= kSchedSixteenMinJ) || chains == 16) K16 = schedule_chains(a.parents, J, 16, s16, true);
        int use = 0;
        if (chains == 2) use = K2 ? 2 : 0;
        else if (chains == 4) use = K4 ? 4 : 0;
        else if (chains == 8) use = K8 ? 8 : 0;
        else if (chains == 16) use = K16 ? 16 : 0;
        else {  // walk cost per frame in sixteenths of a step: J x 2 today (8 frames per wave)
            const int c1 = 2 * J, c2 = K2 ? 2 * K2 : 1 << 30, c4 = K4 ? 4 * K4 : 1 << 30;
            // near-ties go to four chains: the tile is half the size, twice as many waves are resident (measured, 2^18 frames:
            // J = 96 two / four chains 384 / 342 us, J = 65 235 / 228 us; the 52-joint SMPL-H tree, 26 vs 17 steps: 150 / 179 us)
            // ... and so do skeletons whose two-chain tile (8 frames) is too big for more than four waves per CU (narrow
            // 128-joint tree, two / four chains: 727 / 650 us)
            if (K4 && J >= 36 && ((20 * c4 <= 23 * c2 && 4 * c4 <= 3 * c1) || (J > 100 && c4 <= 2 * c2))) use = 4;
            else if (K2 && 4 * c2 <= 3 * c1) use = 2;
            if (use == 4 && K8 && 4 * (8 * K8) <= 5 * (4 * K4)) use = 8;
            if (use == 8 && K16 && 10 * (16 * K16) <= 14 * (8 * K8)) use = 16;
        }
        if (use) {
            sa.rot = rot; sa.root_pos = root_pos; sa.offsets = offsets; sa.dq = dq; sa.F = F; sa.J = J;
            sa.K = use == 2 ? K2 : (use == 4 ? K4 : (use == 8 ? K8 : K16));
            sa.depth = a.depth;
            memcpy(sa.sched, use == 2 ? s2 : (use == 4 ? s4 : (use == 8 ? s8 : s16)), (size_t)sa.K * use);
            for (int j = 0; j < J; ++j) sa.parent[j] = (int16_t)a.parents.p[j];
            if (sched_lds_bytes(J, sa.K, use) <= kMaxLds)
                return use == 2 ? launch_to_root_sched<2>(sa, vec, s) : (use == 4 ? launch_to_root_sched<4>(sa, vec, s) : (use == 8 ? launch_to_root_sched<8>(sa, vec, s) : launch_to_root_sched<16>(sa, vec, s)));
        }
    }
    if (const int v = tune_env("PM_DQ_FPW", 0); v == 16 || v == 8 || v == 4) pick = v;  // PM_TUNING build only
    while (pick > 4 && pick * per_frame + fixed > kMaxLds) pick >>= 1;
    if (pick * per_frame + fixed <= kMaxLds) {
        if (pick == 16) return launch_to_root<16>(a, vec, s);
        if (pick == 8) return launch_to_root<8>(a, vec, s);
        return launch_to_root<4>(a, vec, s);
    }
    set_error("to_root_dq: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
}

extern "C" int pm_to_root_dq_f32(const float *rot, const float *root_pos, const int32_t *parents,
                                 const float *offsets, int64_t F, int32_t J, float *dq, pm_stream_t stream) {
    return to_root_dq_impl(rot, root_pos, parents, offsets, F, J, dq, -1.0f, stream);
}
extern "C" int pm_to_root_dq_hint_f32(const float *rot, const float *root_pos, const int32_t *parents, const float *offsets, int64_t F,
                                      int32_t J, float *dq, float offsets_abs_max, pm_stream_t stream) {
    return to_root_dq_impl(rot, root_pos, parents, offsets, F, J, dq, offsets_abs_max, stream);
}

extern "C" int pm_from_root_dq_f32(const float *dq, const int32_t *parents, int64_t F, int32_t J, float *trans,
                                   float *rot, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "from_root_dq: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(dq && parents && trans && rot, "from_root_dq: null pointer");
    GatherArgs a;
    a.in = dq; a.out0 = trans; a.out1 = rot; a.F = F; a.J = J;
    if (int e = pack_parents(parents, J, a.parents)) return e;
    return launch_gather<0>(a, aligned16(dq) && aligned16(trans) && aligned16(rot), static_cast<hipStream_t>(stream));
}

extern "C" int pm_from_global_rotations_f32(const float *global_quats, const int32_t *parents, int64_t F,
                                            int32_t J, float *local_quats, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "from_global_rotations: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(global_quats && parents && local_quats, "from_global_rotations: null pointer");
    GatherArgs a;
    a.in = global_quats; a.out0 = local_quats; a.out1 = nullptr; a.F = F; a.J = J;
    if (int e = pack_parents(parents, J, a.parents)) return e;
    return launch_gather<1>(a, aligned16(global_quats) && aligned16(local_quats), static_cast<hipStream_t>(stream));
}
