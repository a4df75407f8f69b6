// dqwide.hip -- to_root_dual_quat with the joints of a frame walked SIXTEEN / W AT A TIME from a step list (reference: pymotion/ops/skeleton.py:207-244;
// the loop there has no topology cliff).
//
// The scheduled walk of dq.hip (to_root_dq_sched_kernel) puts C chains on a frame too, but pays for it per tile: a joint table (48 J bytes) and a
// "program" (16 bytes per step and chain) in LDS next to the image, both rebuilt by every workgroup, a ds_read_b128 of program and three table reads
// per step -- ~45 instructions a step, and at sixteen chains 25 KB of LDS a wave at 250 joints (six waves a CU).  fk got out of the same corner in
// round 5 (tree_walk_w4 / fk_wide_kernel): the step list in REGISTERS, nothing in LDS but the image.  The same shape for the quaternion + translation
// payload:
//   * a wave owns FPW = 1, 2, 4 or 8 frames and W = 16 / FPW quads per frame; the host list-schedules the tree into steps of up to W joints whose
//     parents are finished (fk_wide_plan: critical path first, a joint at the earliest one step after its parent; the root takes no step and its
//     children wait for nobody -- they stay local, skeleton.py:236-237);
//   * slot j of the image is the joint's OUTPUT record (32 bytes: the image leaves as one dwordx4 stream).  Until the joint's step it holds the
//     inputs -- (q_local | 0, offset) --, afterwards (q_root | 0, t_root) (the translation as fixed-point words, its zero component the packed
//     residuals of the quaternion, on tiles that take the precise step: dqstep.hpp); slot J is the identity the root's children compose with,
//     slot J + 1 what idle quads read and write; the root's slot is parked as (q_0 | 0, root position): skeleton.py:232, and nobody reads it;
//   * lane c of a quad reads component c of the parent's quaternion and translation at the top of the step (after the writes of the step before:
//     in-order DS), the joint's own (q_c, offset_c) one step ahead (its slot is written by its own step only); 2 v_next / 2 v_nextnext of the
//     rotate-a-vector formula are two DPP multiplies of the offset register -- there is no joint table;
//   * the step words (own slot | parent slot << 16, in bytes) ride in the kernarg segment and are loaded into registers once per workgroup, which
//     takes `nt` consecutive tiles; the next tile's quaternions are in flight while a tile walks (the walk waits for LDS only).
// ~31 instructions a step for sixteen joint-frames.  Same products in the same order as the other walks (dq_step_math / dq_step_precise): their
// results to the bit on finite data.  (A non-finite ROOT quaternion is copied like the reference copies it; the kernels of dq.hip multiply it by the
// identity, which spreads an Inf component as NaN over the other three -- tests/test_gpu_large_magnitude.py names which kernel does which.)
#include "common.hpp"
#include "dqstep.hpp"

namespace pm {

constexpr int kDwSteps = 48, kDwGroups = kDwSteps / 4;  // steps the list holds, four to a register
constexpr int kDwStride = kDwSteps + 8;                 // words per quad in the kernarg segment: its steps, then idle words for the look-ahead
struct DqWideArgs {
    const float *rot, *root_pos, *offsets;
    float *dq;
    int64_t F;
    int32_t J, depth, nsteps, ablate;  // ablate: PM_TUNING build only (PM_DQ_ABLATE): 1 = skip the walk, 2 = skip phase C
    uint32_t jobs[16 * kDwStride];     // [quad of a frame][step]: own slot | parent slot << 16, both in BYTES from the frame's image
};

// dwords between the images of a tile's frames: J + 2 slots, and an odd number of 16-byte units so that the frames' quads start on different banks
__host__ __device__ constexpr int dq_wide_frame_stride(const int J, const int fpw) { return (J + 2) * 8 + (fpw > 1 ? 4 : 0); }

template <bool PRECISE, bool DEEP>
__device__ __forceinline__ void dw_walk(float *fD, const uint32_t (&JW)[kDwGroups + 1], const int nsteps, const int c, const FxScaleD fx) {
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f;   // S[c][1]:  - + - +
    const float s2 = (c == 0 || c == 3) ? -1.0f : 1.0f;   // S[c][2]:  - + + -
    const float s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;   // S[c][3]:  - - + +
    const float live = (c == 0) ? 0.0f : 1.0f;            // lane 0 carries the zero scalar part of (0, t)
    float two = 2.0f;
    asm volatile("" : "+v"(two));                         // (a DPP multiply takes registers only)
    char *bq = reinterpret_cast<char *>(fD + c);          // component c of a slot's quaternion; of its translation 16 bytes on
    const char *bl = reinterpret_cast<const char *>(fD + 4);  // precise: a slot's packed residuals (the zero component of its translation)
    auto word = [](const uint32_t v, auto t) __attribute__((always_inline)) {
        constexpr int T = decltype(t)::value;
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, T * 0x55, 0xf, 0xf, true);  // quad_perm:[T,T,T,T]
    };
    uint32_t w = word(JW[0], IntC<0>{});
    unsigned own = w & 0xffffu;
    float b = *reinterpret_cast<const float *>(bq + own), vc = *reinterpret_cast<const float *>(bq + own + 16);
    auto step = [&](const uint32_t wn) __attribute__((always_inline)) {
        const unsigned par = w >> 16, ownn = wn & 0xffffu;
        const float peq = *reinterpret_cast<const float *>(bq + par), pet = *reinterpret_cast<const float *>(bq + par + 16);
        int pel = 0;
        // (read as the float it was stored as: an int-typed load may be moved above the float-typed store of the step before -- the two types do not alias for
        // the compiler -- and then reads the slot's parked zero instead of the residuals: 5-6e-8 on the quaternions at four frames a wave, found by a soak run)
        if constexpr (PRECISE) pel = __float_as_int(*reinterpret_cast<const float *>(bl + par));
        const float bn = *reinterpret_cast<const float *>(bq + ownn), vn = *reinterpret_cast<const float *>(bq + ownn + 16);
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(b, s2), sb3 = quad_perm_mul<3, 2, 1, 0>(b, s3);
        const float w1 = quad_perm_mul<0, 3, 1, 2>(vc, two), w2 = quad_perm_mul<0, 2, 3, 1>(vc, two);  // 2 v_nextnext, 2 v_next (lane 0: 2 x 0)
        float q, t;
        if constexpr (PRECISE) {
            const double pqd = dq_parent_f64(peq, pel, c);
            int ti, tw;
            dq_step_precise<DEEP>(pqd, __float_as_int(pet), b, sb1, sb2, sb3, vc, w1, w2, live, fx.S, c, q, ti, tw);
            t = __int_as_float(tw);
        } else {
            dq_step_math(peq, vc + pet, b, sb1, sb2, sb3, w1, w2, live, q, t);
        }
        *reinterpret_cast<float *>(bq + own) = q;
        *reinterpret_cast<float *>(bq + own + 16) = t;
        own = ownn; b = bn; vc = vn; w = wn;
    };
    // (unrolled over the groups, with an exit per group: JW[g] picked by a loop counter is an indexed register read, behind which the compiler waits for
    // vector memory -- the next tile's quaternions, which are meant to be in flight across the walk -- at the top of every group)
    // (the list is padded with idle words: a last group of fewer than four steps runs idle ones)
#define PM_DW_GROUP(g)                     \
    if ((g) * 4 >= nsteps) return;         \
    step(word(JW[(g)], IntC<1>{}));        \
    step(word(JW[(g)], IntC<2>{}));        \
    step(word(JW[(g)], IntC<3>{}));        \
    step(word(JW[(g) + 1], IntC<0>{}));
    PM_DW_GROUP(0) PM_DW_GROUP(1) PM_DW_GROUP(2) PM_DW_GROUP(3) PM_DW_GROUP(4) PM_DW_GROUP(5)
    PM_DW_GROUP(6) PM_DW_GROUP(7) PM_DW_GROUP(8) PM_DW_GROUP(9) PM_DW_GROUP(10) PM_DW_GROUP(11)
#undef PM_DW_GROUP
    static_assert(kDwGroups == 12, "dw_walk spells out its groups");
}

// FPW frames a wave (W = 16 / FPW joints of a frame a step), NB batches of 64 records a tile (FPW J <= 64 NB).  A workgroup (one wave) takes `nt`
// consecutive tiles.  DEEP: see dqstep.hpp (kDqF64RotMinDepth).
template <int FPW, int NB, bool DEEP>
__global__ __launch_bounds__(PM_WAVE, NB <= 3 ? 5 : (NB <= 4 ? 4 : 3)) void to_root_dq_wide_kernel(const DqWideArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int W = 16 / FPW, NG = kDwGroups;
    const int lane = threadIdx.x, J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW, ngroups = (ntiles + nt - 1) / nt;
    const int64_t grp = xcd_tile_chunked(ngroups, kXcdChunk);
    if (grp < 0) return;
    const int FS = dq_wide_frame_stride(J, FPW), ne = FPW * J;
    const int quad = lane >> 2, f = quad / W, k = quad % W, c = lane & 3;
    float *fD = smem + f * FS;
    const int64_t t0 = grp * nt, t1 = (t0 + nt < ntiles) ? t0 + nt : ntiles;

    // a tile's global loads: one quaternion per lane and batch, the frames' root positions (lane c of every quad of frame f: component c of (0, root_pos))
    v4f q[NB];
    float rp = 0.0f;
    auto issue = [&](const int64_t tile) __attribute__((always_inline)) {
        const int64_t f0 = tile * FPW;
        const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW), n = nf * J;
        const v4f *src = reinterpret_cast<const v4f *>(a.rot) + f0 * J;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = u * PM_WAVE + lane;  // (no branch per batch: loads are clamped, stores guarded -- a wave-uniform test here made the compiler
            q[u] = __builtin_nontemporal_load(src + (e < n ? e : n - 1));  //  keep a copy of every register array per path: 236 VGPRs at NB = 8)
        }
        rp = (c > 0 && f < nf) ? a.root_pos[(f0 + f) * 3 + c - 1] : 0.0f;
    };
    issue(t0);

    // once per workgroup: this quad's step words, where a batch's record goes in the image, its joint's offset
    uint32_t JW[NG + 1];  // lane (k, t): word of step 4 g + t for quad k of every frame
#pragma unroll
    for (int g = 0; g <= NG; ++g) JW[g] = a.jobs[k * kDwStride + 4 * g + (lane & 3)];
    int so[NB];
    float o[NB][3];
    bool is_root[NB], once[NB];  // the record is a root's; a record of the tile's first frame (every joint once)
    const float invJ = 1.0f / (float)J;
    bool tbig = false;                 // a bone of a metre or more (or NaN) in the table: see kBigOffset
    float tsum = 0.0f, tmx = 0.0f;     // sum / max of the bones' lengths (this lane's share; NaN sticks)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int e = u * PM_WAVE + lane, ec = e < ne ? e : ne - 1;
        const int ef = (FPW == 1) ? 0 : (int)(((float)ec + 0.5f) * invJ);  // ec / J, exact for ec < 2^22
        const int ej = ec - ef * J;
        so[u] = ef * FS + ej * 8;
        is_root[u] = ej == 0; once[u] = e < J && ej > 0;
        o[u][0] = a.offsets[3 * ej]; o[u][1] = a.offsets[3 * ej + 1]; o[u][2] = a.offsets[3 * ej + 2];
    }
#pragma unroll
    for (int g = 0; g <= NG; ++g) asm volatile("" : "+v"(JW[g]));  // settle the list here, not inside the walk (behind the next tile's loads)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        o[u][0] = is_root[u] ? 0.0f : o[u][0]; o[u][1] = is_root[u] ? 0.0f : o[u][1]; o[u][2] = is_root[u] ? 0.0f : o[u][2];  // offsets[0] is ignored (skeleton.py:231-232: the root's translation is its position)
        const float l1 = fsqrt(__builtin_fmaf(o[u][0], o[u][0], __builtin_fmaf(o[u][1], o[u][1], o[u][2] * o[u][2]))) * 1.000001f;  // the bone's LENGTH (fx_scale_exact), rounded up
        const bool big = !(fabsf(o[u][0]) < kBigOffset) || !(fabsf(o[u][1]) < kBigOffset) || !(fabsf(o[u][2]) < kBigOffset);
        tbig = tbig || (once[u] && big);  // every joint once: the first frame's records
        tsum += once[u] ? l1 : 0.0f;
        tmx = (once[u] && (l1 > tmx || l1 != l1)) ? l1 : tmx;
    }
    const bool table_big = __builtin_amdgcn_ballot_w64(tbig) != 0;
    const float bsum = wave_sum(tsum), bmax = (float)a.depth * wave_max(tmx);  // (NaN sticks in both)
    const float tbound = (bmax < bsum) ? bmax : bsum;
    if (k == 0) {  // the identity slot (1,0,0,0 | 0,0,0,0) and the idle slot (zeros; idle steps keep it at zeros)
        fD[J * 8 + c] = (c == 0) ? 1.0f : 0.0f;
        fD[J * 8 + 4 + c] = 0.0f;
        fD[(J + 1) * 8 + c] = 0.0f;
        fD[(J + 1) * 8 + 4 + c] = 0.0f;
    }

    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t f0 = tile * FPW;
        const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW), n = nf * J;
        // park: (q_local | 0, offset) into every record's slot.  A quaternion off unit length anywhere in the tile: the fp32 step (see to_root_dq_kernel)
        bool offunit = false;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            {
                const int e = u * PM_WAVE + lane;
                const float n2 = __builtin_fmaf(q[u].w, q[u].w, __builtin_fmaf(q[u].z, q[u].z, __builtin_fmaf(q[u].y, q[u].y, q[u].x * q[u].x)));
                offunit = offunit || (fabsf(n2 - 1.0f) >= 1e-3f && n2 < 3e38f);  // (NaN / Inf: the float64 chain propagates them; records past the tile's end repeat its last one)
                if (e < n) {
                    *reinterpret_cast<v4f *>(smem + so[u]) = q[u];
                    *reinterpret_cast<v4f *>(smem + so[u] + 4) = v4f{0.0f, o[u][0], o[u][1], o[u][2]};
                }
            }
        }
        const float rpf = rp;  // this tile's root positions (rp is refilled by the next tile's loads)
        // which arithmetic this tile gets (wave-uniform; "Big-magnitude tiles", dqstep.hpp)
        bool precise = false;
        FxScaleD fx = {1.0, 1.0};
        if ((table_big || __builtin_amdgcn_ballot_w64(!(fabsf(rpf) < kBigRoot)) != 0) && __builtin_amdgcn_ballot_w64(offunit) == 0)
            precise = fx_scale_for<DEEP>(tbound, (k == 0) ? fabsf(rpf) : 0.0f, fx);  // false for a non-finite bound: fp32 step
        if (k == 0) {  // the root: (q_0 | 0, root position) -- what its step would have made of it (identity (x) q_0, 0 + root position)
            float tw = rpf;
            if (precise) tw = (c == 0) ? 0.0f : __int_as_float(DEEP ? (int)__builtin_rint((double)rpf * fx.S) : (int)__builtin_rintf(rpf * (float)fx.S));
            fD[4 + c] = tw;
        }
        wave_sync();
        if (tile + 1 < t1) issue(tile + 1);  // in flight while this tile walks
        if (!PM_ABLATED(a, 1)) {
            if (precise) dw_walk<true, DEEP>(fD, JW, a.nsteps, c, fx);
            else dw_walk<false, DEEP>(fD, JW, a.nsteps, c, fx);
        }
        wave_sync();
        // phase C, lane per record: (q, t) -> [q, 0.5 (0,t) (x) q]  (dual_quat.py:28-36)
        if (!PM_ABLATED(a, 2)) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                {
                    const int e = u * PM_WAVE + lane;
                    float *slot = smem + so[u];
                    const v4f qq = *reinterpret_cast<const v4f *>(slot), tt = *reinterpret_cast<const v4f *>(slot + 4);
                    float t[3] = {tt.y, tt.z, tt.w};
                    if (precise) {  // wave-uniform: the translation words are fixed point
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            t[i] = DEEP ? (float)((double)__float_as_int(t[i]) * fx.invS) : (float)__float_as_int(t[i]) * (float)fx.invS;
                    }
                    const float qa[4] = {qq.x, qq.y, qq.z, qq.w};
                    float d[8];
                    rt2dq(qa, t, d);
                    if (e < n) *reinterpret_cast<v4f *>(slot + 4) = v4f{d[4], d[5], d[6], d[7]};
                }
                if (u & 1) asm volatile("" ::: "memory");  // two records at a time: all NB at once (their sixteen reads hoisted to the top) set the kernel's register budget
            }
        }
        wave_sync();
        // copy-out: dwordx4 i of the tile is chunk i % 2J of frame i / 2J
        v4f *gout = reinterpret_cast<v4f *>(a.dq) + f0 * J * 2;
        const int n4 = n * 2, J2 = 2 * J;
        const float invJ2 = 1.0f / (float)J2;
        for (int i = lane; i < n4; i += PM_WAVE) {
            const int ff = (FPW == 1) ? 0 : (int)(((float)i + 0.5f) * invJ2);
            __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(smem + ff * FS + (i - ff * J2) * 4), gout + i);
        }
        wave_sync();  // the image is the next tile's
    }
}

template <int FPW, int NB>
static int launch_dq_wide(const DqWideArgs &a, const int nt, hipStream_t s) {
    const size_t lds = (size_t)FPW * dq_wide_frame_stride(a.J, FPW) * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW, ngroups = (ntiles + nt - 1) / nt, grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("to_root_dq: grid too large"); return PM_EUNSUPPORTED; }
    const bool deep = a.depth >= kDqF64RotMinDepth;
    set_kernel_name("void pm::to_root_dq_wide_kernel<%d, %d, %s>(pm::DqWideArgs, int)", FPW, NB, tf(deep));
    auto go = [&](auto kf) {
        if (int e = allow_lds(kf, lds)) return e;
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
        return (int)PM_OK;
    };
    if (int e = deep ? go(to_root_dq_wide_kernel<FPW, NB, true>) : go(to_root_dq_wide_kernel<FPW, NB, false>)) return e;
    return PM_AFTER_LAUNCH("to_root_dq launch");
}

template <int FPW>
static int launch_dq_wide_nb(const DqWideArgs &a, const int nt, hipStream_t s) {
    const int ne = FPW * a.J;  // (batches past the tile's records are clamped loads and guarded stores: five sizes bound that waste)
    if (ne <= 2 * PM_WAVE) return launch_dq_wide<FPW, 2>(a, nt, s);
    if (ne <= 3 * PM_WAVE) return launch_dq_wide<FPW, 3>(a, nt, s);
    if (ne <= 4 * PM_WAVE) return launch_dq_wide<FPW, 4>(a, nt, s);
    if (ne <= 6 * PM_WAVE) return launch_dq_wide<FPW, 6>(a, nt, s);
    return launch_dq_wide<FPW, 8>(a, nt, s);
}

// The kernel's step words for this tree at 16 / fpw joints a step: jobs[k * kDwStride + step] = own slot | parent slot << 16, in bytes of the 32-byte slots (the root's
// children compose with the identity slot J, skeleton.py:236-237; idle: slot J + 1 composed with the identity).  Returns the number of steps, -1 when the tree needs
// more than kDwSteps.
int dq_wide_words(const Parents &par, const int J, const int fpw, uint32_t *jobs) {
    const int W = 16 / fpw;
    uint32_t list[(kDwSteps + 2) * 16];
    const int nsteps = (J == 1) ? 0 : fk_wide_plan(par, J, W, kDwSteps, false, list);
    if (nsteps < 0) return -1;
    const uint32_t idle = (uint32_t)((J + 1) * 32) | ((uint32_t)(J * 32) << 16);
    for (int k = 0; k < 16; ++k)
        for (int st = 0; st < kDwStride; ++st) {
            uint32_t w = idle;
            if (k < W && st < nsteps) {
                const uint32_t j = list[st * W + k] & 0xffffu, p = list[st * W + k] >> 16;
                if ((int)j < J) w = (j * 32u) | ((p == 0 ? (uint32_t)J : p) * 32u) << 16;
            }
            jobs[k * kDwStride + st] = w;
        }
    return nsteps;
}

// to_root_dual_quat on 16-byte aligned arrays with `fpw` = 1, 2, 4 or 8 frames a wave.  Returns false (nothing launched) when fpw x J records do not fit
// eight batches or the tree needs more than kDwSteps steps of 16 / fpw joints, or more than max_quad_steps_per_joint_x10 / 10 quad-steps per joint
// (0: no such bound); true with rc set otherwise.
bool try_to_root_dq_wide(const int fpw, const float *rot, const float *root_pos, const float *offsets, float *dq, const int64_t F, const int32_t J,
                         const int32_t depth, const Parents &par, const int ablate, const int max_quad_steps_per_joint_x10, hipStream_t s, int &rc) {
    if ((fpw != 1 && fpw != 2 && fpw != 4 && fpw != 8) || fpw * J > 8 * PM_WAVE) return false;
    const int W = 16 / fpw;
    DqWideArgs a;
    a.nsteps = dq_wide_words(par, J, fpw, a.jobs);
    if (a.nsteps < 0) return false;
    if (max_quad_steps_per_joint_x10 > 0 && a.nsteps * W * 10 > max_quad_steps_per_joint_x10 * J) return false;
    a.rot = rot; a.root_pos = root_pos; a.offsets = offsets; a.dq = dq; a.F = F; a.J = J; a.depth = depth; a.ablate = ablate;
    // tiles per workgroup: the words and the offsets are loaded once and the next tile's quaternions are requested before a tile's walk, but a launch wants
    // many more workgroups than the chip has wave slots (the last round of a launch runs part empty).  Same-box sweep at 2^18...2^20 frames, one / two / four
    // tiles: 22 joints 193 / 190 / 199 us, 32 joints 255 / 276 / 299, 64 joints 263 / 278 / 308, 128 joints 570 / 532 / 605, 250 joints 549 / 546 / 603,
    // 512 joints 70.3 / 75.2 / 67.4 % of the HBM spec.  (PM_DQW_NT, PM_TUNING build only)
    const int64_t ntiles = (F + fpw - 1) / fpw;
    int nt = (J > 100 && ntiles >= 65536) ? 2 : 1;
    nt = tune_env("PM_DQW_NT", nt);
    if (nt < 1) nt = 1;
    rc = fpw == 1 ? launch_dq_wide_nb<1>(a, nt, s) : (fpw == 2 ? launch_dq_wide_nb<2>(a, nt, s) : (fpw == 4 ? launch_dq_wide_nb<4>(a, nt, s) : launch_dq_wide_nb<8>(a, nt, s)));
    return true;
}

}  // namespace pm

extern "C" int pm_step_list_plan_debug(const int32_t *parents, int32_t J, int32_t op, int32_t fpw, uint32_t *jobs) {
    PM_CHECK_ARGS(parents && jobs && J >= 1 && J <= PM_MAX_JOINTS, "step_list_plan: need parents, jobs and 1 <= J <= PM_MAX_JOINTS");
    PM_CHECK_ARGS((op == 0 || op == 1) && (fpw == 1 || fpw == 2 || fpw == 4 || fpw == 8), "step_list_plan: op 0 (to_root_dual_quat) or 1 (mirror), fpw 1 / 2 / 4 / 8");
    pm::Parents par;
    if (int e = pm::pack_parents(parents, J, par)) return e < 0 ? e : -e;
    const int n = op == 0 ? pm::dq_wide_words(par, J, fpw, jobs) : pm::mirror_wide_words(par, J, fpw, jobs);
    return n < 0 ? PM_EUNSUPPORTED : n;
}
