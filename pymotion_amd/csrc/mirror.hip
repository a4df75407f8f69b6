// mirror.hip -- skeleton mirroring, rotation part, for gfx950.
//   ops/skeleton.py:247-344 `mirror` (modes 'all' / 'symmetry'), :347-418 `_true_mirror`
// The reference chains  fk -> quat.from_matrix -> [joint permutation] -> negate two components ->
// from_global_rotations  through four full-size arrays.  World ROTATIONS depend neither on offsets nor on
// the root position, and  from_matrix(to_matrix(g)) = sign(g[dom]) g / |g|  where `dom` is the component
// its four-way branch (quat.py:85-156) makes dominant -- so no matrix is ever formed here:
//   phase A  lane per (frame, joint): quaternion straight from HBM, normalised like fk does (skeleton.py:45),
//            parked in a 16-byte slot of the frame's LDS image (16 J B per frame: 27 waves per CU at J = 22);
//   walk     four lanes per frame compose world quaternions down the tree, g_j = g_parent (x) q_j, with the
//            quad exchanges folded into DPP operands (quad_qmul): ~12 instructions a step for 16 frames, branch
//            free, the parent read one step ahead unless it is the previous joint (register chain);
//   finish   lane per (frame, joint): the reference's sign (the from_matrix branch, decided from the matrix
//            diagonal the quaternion implies) and normalisation, the joint permutation, the two negated
//            components, local'_j = conj(g'_parent(j)) (x) g'_j (skeleton.py:322-331 / :410-416), stored
//            straight from registers (one record per lane = a contiguous dwordx4 stream).
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace pm {

struct Map16 { int16_t m[PM_MAX_JOINTS]; };
constexpr int kMirrorDeepMinJ = 66;   // from here on mode 'all' walks one lane per frame where the topology allows (chain-like skeletons at 2^19 frames, lane per
                                      // frame / scheduled walk, % of the HBM spec: J = 32 75 / 69, 44 69 / 64, 48 67 / 67, 50 56 / 62, 56 61 / 65, 64 60 / 63,
                                      // 65 54 / 58, 66 57 / 55, 72 60 / 58, 80 60 / 55.5, 96 60 / 53, 128 61 / 45; the SMPL-H tree at 2^18: 68 / 60)
constexpr int kMirrorWideMinJ = 40;   // from here on the step-list walk (mirror_wide_kernel) goes first
constexpr int kMirrorSchedMinJ = 40;  // from here on the scheduled walk is considered (measured: see pm_mirror_rotations_f32)

struct MirrorArgs {
    const float *rot;   // [F,J,4] local rotations
    float *out;         // [F,J,4] mirrored local rotations
    int64_t F;
    int32_t J;
    int32_t c0, c1;     // quaternion components to negate (X: 2,3  Y: 1,3  Z: 1,2)
    int32_t K;          // C > 1 chains per frame: steps of the schedule
    Parents parents;
    Map16 mapping;      // identity for mode 'all'
    uint8_t sched[kSchedMax];  // [K][C]: joint index, 255 = idle (C > 1)
};

// J slots + the identity slot (+ an idle slot for the scheduled walk), padded so that (stride / 4) is odd: the quads of 8 frames
// hit 8 distinct bank groups
__host__ __device__ constexpr int mirror_frame_stride(const int J, const int C) { return 4 * ((J + (C > 1 ? 2 : 1)) | 1); }

// from_matrix(to_matrix(g)) without the matrices: quat.py:276-317 gives the diagonal, quat.py:85-156 the branch,
// and each branch's candidate is 4 g[dom] g; then its normalize (:155, quat.py:411-423).
// The branch predicates are evaluated on the quaternion itself, where nothing cancels against 1 (round 2 formed the diagonal
// 1 - 2 (yy + zz) ... in fp32 and compared those): with n = |g|^2,
//     r22 < 0      <=>  1 - 2 (xx + yy) / n < 0          <=>  xx + yy > ww + zz
//     r00 > r11    <=>  xx > yy                           <=>  |x| > |y|
//     r00 < -r11   <=>  2 - 2 (xx + yy) / n - 4 zz / n < 0  <=>  ww < zz  <=>  |w| < |z|
// so the decision is as good as g (exact comparisons of its components) and only flips against the float64 reference where
// two components of g tie to within g's own chain error.
__device__ __forceinline__ void canonical_sign(const float (&g)[4], float (&o)[4]) {
    const float ww = g[0] * g[0], xx = g[1] * g[1], yy = g[2] * g[2], zz = g[3] * g[3];
    const float dom = (xx + yy > ww + zz) ? ((xx > yy) ? g[1] : g[2]) : ((ww < zz) ? g[3] : g[0]);
    const float sg = (dom < 0.0f) ? -1.0f : 1.0f;
    const float c[4] = {sg * g[0], sg * g[1], sg * g[2], sg * g[3]};
    qnormalize(c, 1e-8f, o);
}

// C = chains per frame.  C = 1: 4 lanes per frame walk the joints in index order (below).  C = 2 / 4: the joints are list-scheduled
// onto C quads per frame as in to_root_dq_sched_kernel (dq.hip) -- subtrees are independent once their common ancestor is done, so
// the walk is K ~ max(J / C, depth) steps instead of J, at 16 / C frames per wave: the 52-joint SMPL-H tree walks 27 steps with
// two chains, 17 with four.  FPW = 16 / C there.
template <int FPW, bool VEC, int C>
__global__ __launch_bounds__(PM_WAVE) void mirror_kernel(const MirrorArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile_chunked(ntiles, kXcdChunk);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int n = nf * J;
    const int FS = mirror_frame_stride(J, C);
    float *sQ = smem;                                          // [FPW * FS]
    int *sPar = reinterpret_cast<int *>(sQ + FPW * FS);        // [J+1]  walk parent (the root: the identity slot J); entry J repeats
    int *sMap = sPar + (J + 1);                                // [2J]   {mapping[j], mapping[parents[j]]}
    typedef int v2i __attribute__((ext_vector_type(2)));
    v2i *sProg = reinterpret_cast<v2i *>(sMap + 2 * J + ((J & 1) ^ 1));  // C > 1: [(K + 2) * C] {own slot | on-chain << 31, parent slot} in bytes; 8-byte aligned
    const float invJ = 1.0f / (float)J;
    if constexpr (C > 1) {
        for (int i = lane; i < (a.K + 2) * C; i += PM_WAVE) {  // two idle steps of slack for the look-ahead
            const int st = i / C, kk = i - st * C;
            const int j = (st < a.K) ? a.sched[i] : 255;
            v2i e = v2i{(J + 1) * 16, J * 16};  // idle: the scratch slot, composed with the identity
            if (j != 255) {
                const int p = (j == 0) ? J : a.parents.p[j];
                const int prev = (st > 0) ? a.sched[(st - 1) * C + kk] : 255;
                e = v2i{j * 16 | ((p != J && prev == p) ? (int)0x80000000 : 0), p * 16};
            }
            sProg[i] = e;
        }
    }

    for (int j = lane; j <= J; j += PM_WAVE) {
        const int jc = j < J ? j : J - 1;
        sPar[j] = (jc == 0) ? J : a.parents.p[jc];
        if (j < J) { sMap[2 * j] = a.mapping.m[j]; sMap[2 * j + 1] = a.mapping.m[j == 0 ? 0 : a.parents.p[j]]; }
    }
    for_each_record4<VEC>(a.rot + f0 * J * 4, n, lane, [&](const int e, const v4f q, const bool valid) {
        const float qi[4] = {q.x, q.y, q.z, q.w};
        float u[4];
        qnormalize(qi, 1e-8f, u);  // skeleton.py:45
        const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
        const int j = e - f * J;
        if (valid) *reinterpret_cast<v4f *>(sQ + f * FS + j * 4) = v4f{u[0], u[1], u[2], u[3]};
    });
  if constexpr (C == 1) {
    const int wl = lane % (4 * FPW);  // lanes >= 4*FPW shadow lanes 0..; frames past a partial tile walk their own slots
    const int fq = wl >> 2, c = wl & 3;
    float *fD = sQ + fq * FS;
    fD[J * 4 + c] = (c == 0) ? 1.0f : 0.0f;  // the identity slot
    wave_sync();

    // ---- the walk (see to_root_dq_kernel for the same scheme with a translation on top) ------------------
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f;   // S[c][1]:  - + - +
    const float s2 = (c == 0 || c == 3) ? -1.0f : 1.0f;   // S[c][2]:  - + + -
    const float s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;   // S[c][3]:  - - + +
    const float *fDq = fD + c;
    float *oq = fD + c;                                   // slot of the current pair's first joint
    float bA = oq[0], bB = oq[4];                         // own component of joints j, j+1 (requested two steps ahead)
    float gq = 0.0f;                                      // previous joint's world quaternion (component c)
    float peA = (c == 0) ? 1.0f : 0.0f, peB = 0.0f;       // parent read one step ahead; the root composes with the identity
    int par = J;
    auto step = [&](const int j, const int o, const int parn, float &b, const float pe, float &pen, const bool may_be_dummy) {
        pen = fDq[parn * 4];  // parent of joint j+1, if it is not joint j itself (then: a stale value, unused)
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(b, s2), sb3 = quad_perm_mul<3, 2, 1, 0>(b, s3);
        const float pq = (par == j - 1) ? gq : pe;  // wave-uniform
        const float q = quad_qmul(pq, b, sb1, sb2, sb3);
        if (!may_be_dummy || j < J) oq[o] = q;
        b = oq[o + 8];                              // joint j+2: its slot still holds the local quaternion
        gq = q; par = parn;
    };
    for (int jb = 0; jb < J; jb += PM_WAVE) {
        const int i0 = jb + 1 + lane;
        const int pv = sPar[i0 < J ? i0 : J];       // parents of joints jb+1 .. jb+64 across the lanes
        const int jend = (J - jb) < PM_WAVE ? (J - jb) : PM_WAVE;
        asm volatile("" ::"v"(pv));                 // settle the window load here, not as an lgkmcnt(0) inside the loop
        for (int jj = 0; jj < jend; jj += 2) {      // pairs; for odd J the very last step is a dummy that stores nothing
            step(jb + jj, 0, __builtin_amdgcn_readlane(pv, jj), bA, peA, peB, false);
            step(jb + jj + 1, 4, __builtin_amdgcn_readlane(pv, jj + 1), bB, peB, peA, true);
            oq += 8;
        }
    }
    wave_sync();

  } else {
    // ---- the scheduled walk: quad (frame fq, chain k) runs its column of the program ---------------------------------------
    const int fq = lane / (4 * C), k = (lane >> 2) % C, c = lane & 3;
    float *fD = sQ + fq * FS;
    if (k == 0) { fD[J * 4 + c] = (c == 0) ? 1.0f : 0.0f; fD[(J + 1) * 4 + c] = (c == 0) ? 1.0f : 0.0f; }  // identity and idle slots
    wave_sync();
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f, s2 = (c == 0 || c == 3) ? -1.0f : 1.0f, s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;
    const char *bq = reinterpret_cast<const char *>(fD + c);
    const v2i *prog = sProg + k;
    // operands of step st + 1 are requested before step st computes: a parent finished at step st - 1 or earlier is in its slot
    // by now (in-order DS), one finished at step st is this quad's own register chain (the scheduler guarantees it)
    v2i e = prog[0], en = prog[C];
    float b = *reinterpret_cast<const float *>(bq + (e.x & 0x7fffffff)), pe = *reinterpret_cast<const float *>(bq + e.y);
    float gq = 0.0f;
    for (int st = 0; st < a.K; ++st) {
        const v2i enn = prog[(st + 2) * C];
        const float bn = *reinterpret_cast<const float *>(bq + (en.x & 0x7fffffff));  // its slot holds the local quaternion until its own step
        const float pen = *reinterpret_cast<const float *>(bq + en.y);
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(b, s2), sb3 = quad_perm_mul<3, 2, 1, 0>(b, s3);
        const float pq = (e.x < 0) ? gq : pe;  // uniform within the quad
        const float q = quad_qmul(pq, b, sb1, sb2, sb3);
        *reinterpret_cast<float *>(const_cast<char *>(bq) + (e.x & 0x7fffffff)) = q;
        gq = q; e = en; en = enn; b = bn; pe = pen;
    }
    wave_sync();
  }

    // ---- finish, lane per (frame, joint) --------------------------------------------------------------------
    // (this kernel is VALU-bound -- SQ_ACTIVE_INST_VALU accounts for every SIMD cycle -- so each world quaternion
    // gets the reference's sign and normalisation ONCE, in place, not once as a child and once per child it has)
    for_each_slot<2>(n, lane, [&](const int e, const bool valid) {
        const int fr = (int)(((float)e + 0.5f) * invJ);
        const int j = e - fr * J;
        float *slot = sQ + fr * FS + 4 * j;
        float g[4], cg[4];
        lds_get<4>(slot, 0, g);
        canonical_sign(g, cg);
        if (valid) *reinterpret_cast<v4f *>(slot) = v4f{cg[0], cg[1], cg[2], cg[3]};
    });
    wave_sync();
    const float f1 = (a.c0 == 1 || a.c1 == 1) ? -1.0f : 1.0f, f2 = (a.c0 == 2 || a.c1 == 2) ? -1.0f : 1.0f,
                f3 = (a.c0 == 3 || a.c1 == 3) ? -1.0f : 1.0f;
    float *gout = a.out + f0 * J * 4;
    for_each_slot<2>(n, lane, [&](const int e, const bool valid) {
        const int fr = (int)(((float)e + 0.5f) * invJ);
        const int j = e - fr * J;
        const float *fq_ = sQ + fr * FS;
        float cg[4], cp[4], o[4];
        lds_get<4>(fq_, sMap[2 * j], cg);
        lds_get<4>(fq_, sMap[2 * j + 1], cp);
        cg[1] *= f1; cg[2] *= f2; cg[3] *= f3;  // skeleton.py:310-318
        const float inv[4] = {cp[0], -cp[1] * f1, -cp[2] * f2, -cp[3] * f3};
        qmul(inv, cg, o);                       // skeleton.py:85-91 on the mirrored world rotations
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (j == 0) ? cg[k] : o[k];  // the root has no parent (select, not branch)
        if (valid) {
            if (VEC) __builtin_nontemporal_store(v4f{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4f *>(gout) + e);
            else { gout[4 * e] = o[0]; gout[4 * e + 1] = o[1]; gout[4 * e + 2] = o[2]; gout[4 * e + 3] = o[3]; }
        }
    });
}


// ---------------------------------------------------------------------------------------------------------------------------
// Mode 'all' (no joint permutation) on skeletons whose open branch points fit deep.hip's six register slots: ONE LANE PER FRAME,
// the joints streamed through a ring of sixteen 16-byte LDS slots per frame (a quaternion comes in, the mirrored local rotation goes
// out through the same slot; groups of eight records cut at the 128-byte lines of both arrays, a group's slots read into registers
// before the next group is parked over them and stored after: the structure of from_root_positions_order_kernel, ik.hip).  A lane's
// state is the world quaternion of the joint before and its mirrored, sign-fixed form; a parent that is not the previous joint
// comes from a saved register set.  Nothing grows with J, and a chain-like skeleton -- where the scheduled walk has nothing to
// schedule -- costs what a bushy one costs.
// ---------------------------------------------------------------------------------------------------------------------------
struct MirrorDeepArgs {
    const float *rot;
    float *out;
    int64_t F;
    int32_t J;
    int32_t c0, c1;
    DeepTopo topo;
};
struct MirrorSaves { float g[kDeepSlots][4], c[kDeepSlots][4]; };
template <int K>
__device__ __forceinline__ void mirror_slot_load(const int ld, const MirrorSaves &sv, float (&g)[4], float (&c)[4]) {
    if constexpr (K < kDeepSlots) {
        int code = ld;
        asm volatile("" : "+s"(code));  // an opaque copy per test (deep.hip: an indexed array would live in scratch memory)
        if (code == K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { g[i] = sv.g[K][i]; c[i] = sv.c[K][i]; }
        }
        mirror_slot_load<K + 1>(ld, sv, g, c);
    }
}
template <int K>
__device__ __forceinline__ void mirror_slot_save(const int st, MirrorSaves &sv, const float (&g)[4], const float (&c)[4]) {
    if constexpr (K < kDeepSlots) {
        int code = st;
        asm volatile("" : "+s"(code));
        if (code == K) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv.g[K][i] = g[i]; sv.c[K][i] = c[i]; }
        }
        mirror_slot_save<K + 1>(st, sv, g, c);
    }
}

constexpr int kMirrorDeepRow = 16 * 4 + 4;  // sixteen 16-byte slots + 16 bytes: (row / 4) odd

__global__ __launch_bounds__(PM_WAVE) void mirror_deep_kernel(const MirrorDeepArgs a) {
    // the ring leaves room for eight waves on a CU (this kernel runs from 66 joints on: the tables take the ninth's share), which the dispatcher
    // spreads evenly only if no SIMD can take a third: a register count beyond 512 / 3 makes sure (146 used; chain-like 2^19 x 72 / 96 / 128:
    // 254 / 341 / 441 us without, 249 / 334 / 427 us with -- see from_root_positions_order_kernel)
    asm volatile("; two waves per SIMD" ::: "v183");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int G = 8, RS = kMirrorDeepRow;
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t tile = xcd_tile((a.F + PM_WAVE - 1) / PM_WAVE);
    if (tile < 0) return;
    float *sImg = smem;  // [64][RS]
    const int64_t f0 = tile * PM_WAVE;  // a multiple of 64: (f0 + fr) J & 7 == fr J & 7
    const int nf = (int)((a.F - f0) < PM_WAVE ? (a.F - f0) : PM_WAVE);
    const int ngroups = ((J + 6) >> 3) + 1;
    const float *gin = a.rot + f0 * J * 4;
    float *gout = a.out + f0 * J * 4;
    const float f1 = (a.c0 == 1 || a.c1 == 1) ? -1.0f : 1.0f, f2 = (a.c0 == 2 || a.c1 == 2) ? -1.0f : 1.0f, f3 = (a.c0 == 3 || a.c1 == 3) ? -1.0f : 1.0f;

    // lane = (frl, place) = (lane >> 3, lane & 7) for loads and stores; instruction u covers frame 8 u + frl, whose shift is that of frl
    const int l_frl = lane >> 3, l_d = (lane & 7) - ((l_frl * J) & 7);  // joint of this lane's place in group c: 8 c + l_d
    v4f pre[G], pre1[G];
    auto issue = [&](const int c, v4f (&pre)[G]) {
        int j = 8 * c + l_d;
        j = j < 0 ? 0 : (j > J - 1 ? J - 1 : j);  // outside the frame: a valid record again, parked where nobody reads
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int fr = 8 * u + l_frl, fc = fr < nf ? fr : nf - 1;
            pre[u] = *reinterpret_cast<const v4f *>(gin + ((int64_t)fc * J + j) * 4);  // (not nontemporal: a misaligned frame's line is shared with the next group)
        }
    };
    auto park = [&](const int c, const v4f (&pre)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            float *p = sImg + (8 * u + l_frl) * RS + ((c & 1) * G + (lane & 7)) * 4;
            if (PM_LDS_OK(p, 16u)) *reinterpret_cast<v4f *>(p) = pre[u];
        }
    };
    issue(0, pre);
    if (ngroups > 1) issue(1, pre1);
    float *row = sImg + lane * RS;
    const int sf = (lane * J) & 7;
    float g[4] = {1.0f, 0.0f, 0.0f, 0.0f}, cm[4] = {1.0f, 0.0f, 0.0f, 0.0f};  // world quaternion of the previous joint; its mirrored, sign-fixed form
    MirrorSaves sv = {};
    auto walk = [&](const int c) {
        const int jlo = 8 * c - 7 < 0 ? 0 : 8 * c - 7, jhi = 8 * c > J - 1 ? J - 1 : 8 * c;
#pragma unroll 1
        for (int j = jlo; j <= jhi; ++j) {
            const int code = __builtin_amdgcn_readfirstlane(a.topo.code[j]), ld = code & 0xff, st = code >> 8;  // wave-uniform (kernarg)
            float *slot = row + ((j + sf) & 15) * 4;
            float qi[4], q[4];
            lds_get<4>(slot, 0, qi);
            qnormalize(qi, 1e-8f, q);  // skeleton.py:45
            float gp[4] = {g[0], g[1], g[2], g[3]}, cp[4] = {cm[0], cm[1], cm[2], cm[3]};
            mirror_slot_load<0>(ld, sv, gp, cp);
            if (ld == DEEP_ROOT) { gp[0] = 1.0f; gp[1] = 0.0f; gp[2] = 0.0f; gp[3] = 0.0f; }
            qmul(gp, q, g);
            float cg[4];
            canonical_sign(g, cg);
            cm[0] = cg[0]; cm[1] = cg[1] * f1; cm[2] = cg[2] * f2; cm[3] = cg[3] * f3;  // skeleton.py:310-318
            const float inv[4] = {cp[0], -cp[1], -cp[2], -cp[3]};
            float o[4];
            qmul(inv, cm, o);                                                          // skeleton.py:85-91 on the mirrored world rotations
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (ld == DEEP_ROOT) ? cm[k] : o[k];       // the root has no parent
            lds_put<4>(slot, 0, o);
            mirror_slot_save<0>(st, sv, g, cm);
        }
    };
    v4f outr[G];
    auto read_group = [&](const int k) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const float *p = sImg + (8 * u + l_frl) * RS + ((k & 1) * G + (lane & 7)) * 4;
            outr[u] = PM_LDS_OK(p, 16u) ? *reinterpret_cast<const v4f *>(p) : v4f{0, 0, 0, 0};
        }
    };
    auto store_group = [&](const int k) {
        const int j = 8 * k + l_d;
        const bool jok = j >= 0 && j < J;
        float *g0 = gout + (l_frl * J + (jok ? j : 0)) * 4;
#pragma unroll
        for (int u = 0; u < G; ++u)
            if (jok && 8 * u + l_frl < nf) __builtin_nontemporal_store(outr[u], reinterpret_cast<v4f *>(g0 + 8 * u * J * 4));
    };
    park(0, pre);
    if (ngroups > 1) park(1, pre1);
    if (ngroups > 2) issue(2, pre);
    for (int c = 0; c <= ngroups; ++c) {
        wave_sync();
        walk(c);
        wave_sync();
        if (c >= 1) {
            read_group(c - 1);
            wave_sync();
            if (c + 1 < ngroups) park(c + 1, pre);   // over the slots just read; requested a whole step ago
            if (c + 2 < ngroups) issue(c + 2, pre);  // in flight during the next step
            store_group(c - 1);
        }
    }
}

static int launch_mirror_deep(const MirrorDeepArgs &a, hipStream_t s) {
    const size_t lds = (size_t)PM_WAVE * kMirrorDeepRow * sizeof(float);
    const int64_t ntiles = (a.F + PM_WAVE - 1) / PM_WAVE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("mirror: grid too large"); return PM_EUNSUPPORTED; }
    set_kernel_name("pm::mirror_deep_kernel(pm::MirrorDeepArgs)");
    if (int e = allow_lds(mirror_deep_kernel, lds)) return e;
    hipLaunchKernelGGL(mirror_deep_kernel, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH("mirror (lane per frame) launch");
}

// ---------------------------------------------------------------------------------------------------------------------------
// (round 6) The walk from a STEP LIST in registers, 16 / FPW joints of a frame a step -- the shape of to_root_dq_wide_kernel (dqwide.hip; fk's
// tree_walk_w4 / fk_wide_kernel before it): the scheduled walk above keeps a program (8 bytes a step and chain) and two tables in LDS, rebuilt by
// every workgroup, and its schedule stops at kSchedMaxJoints -- beyond, a wide tree fell to the one-chain walk, J dependent steps on four frames a
// wave.  Here nothing is in LDS but the image (16 bytes a joint, + the identity and the idle slot), the list is fk_wide_plan's (a joint at the earliest
// one step after its parent; the root takes no step: its slot holds q_0 as parked, which is identity (x) q_0), a quad reads its parent's component at
// the top of the step and its own local quaternion one step ahead, and a record's place in the image, its mapped slot and its parent's mapped slot
// (skeleton.py:322-331) are registers of the lane that finishes it.  Same products in the same order as the walks above: their bits.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kMwSteps = 48, kMwGroups = kMwSteps / 4, kMwStride = kMwSteps + 8;
struct MirrorWideArgs {
    const float *rot;
    float *out;
    int64_t F;
    int32_t J, c0, c1, nsteps;
    Parents parents;
    Map16 mapping;
    uint32_t jobs[16 * kMwStride];  // [quad of a frame][step]: own slot | parent slot << 16, in BYTES from the frame's image
};
__host__ __device__ constexpr int mirror_wide_frame_stride(const int J) { return 4 * ((J + 2) | 1); }  // an odd number of 16-byte slots: the frames' quads start on different banks

__device__ __forceinline__ void mw_walk(float *fD, const uint32_t (&JW)[kMwGroups + 1], const int nsteps, const int c) {
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f, s2 = (c == 0 || c == 3) ? -1.0f : 1.0f, s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;
    char *bq = reinterpret_cast<char *>(fD + c);
    auto word = [](const uint32_t v, auto t) __attribute__((always_inline)) {
        constexpr int T = decltype(t)::value;
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, T * 0x55, 0xf, 0xf, true);  // quad_perm:[T,T,T,T]
    };
    uint32_t w = word(JW[0], IntC<0>{});
    unsigned own = w & 0xffffu;
    float b = *reinterpret_cast<const float *>(bq + own);
    auto step = [&](const uint32_t wn) __attribute__((always_inline)) {
        const unsigned par = w >> 16, ownn = wn & 0xffffu;
        const float pq = *reinterpret_cast<const float *>(bq + par);   // finished a step ago or earlier (in-order DS)
        const float bn = *reinterpret_cast<const float *>(bq + ownn);  // the next step's joint: its slot holds the local quaternion until its own step
        const float sb1 = quad_perm_mul<1, 0, 3, 2>(b, s1), sb2 = quad_perm_mul<2, 3, 0, 1>(b, s2), sb3 = quad_perm_mul<3, 2, 1, 0>(b, s3);
        const float q = quad_qmul(pq, b, sb1, sb2, sb3);
        *reinterpret_cast<float *>(bq + own) = q;
        own = ownn; b = bn; w = wn;
    };
    // (spelled out group by group, an exit per group: see dw_walk, dqwide.hip)
#define PM_MW_GROUP(g)                     \
    if ((g) * 4 >= nsteps) return;         \
    step(word(JW[(g)], IntC<1>{}));        \
    step(word(JW[(g)], IntC<2>{}));        \
    step(word(JW[(g)], IntC<3>{}));        \
    step(word(JW[(g) + 1], IntC<0>{}));
    PM_MW_GROUP(0) PM_MW_GROUP(1) PM_MW_GROUP(2) PM_MW_GROUP(3) PM_MW_GROUP(4) PM_MW_GROUP(5)
    PM_MW_GROUP(6) PM_MW_GROUP(7) PM_MW_GROUP(8) PM_MW_GROUP(9) PM_MW_GROUP(10) PM_MW_GROUP(11)
#undef PM_MW_GROUP
    static_assert(kMwGroups == 12, "mw_walk spells out its groups");
}

// FPW frames a wave (W = 16 / FPW joints of a frame a step), NB batches of 64 records a tile (FPW J <= 64 NB); a workgroup (one wave) takes `nt` tiles.
template <int FPW, int NB>
__global__ __launch_bounds__(PM_WAVE) void mirror_wide_kernel(const MirrorWideArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int W = 16 / FPW, NG = kMwGroups;
    const int lane = threadIdx.x, J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW, ngroups = (ntiles + nt - 1) / nt;
    const int64_t grp = xcd_tile_chunked(ngroups, kXcdChunk);
    if (grp < 0) return;
    const int FS = mirror_wide_frame_stride(J), ne = FPW * J;
    const int quad = lane >> 2, f = quad / W, k = quad % W, c = lane & 3;
    float *fD = smem + f * FS;
    const int64_t t0 = grp * nt, t1 = (t0 + nt < ntiles) ? t0 + nt : ntiles;
    v4f q[NB];
    auto issue = [&](const int64_t tile) __attribute__((always_inline)) {
        const int64_t f0 = tile * FPW;
        const int n = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW) * J;
        const v4f *src = reinterpret_cast<const v4f *>(a.rot) + f0 * J;
#pragma unroll
        for (int u = 0; u < NB; ++u) {  // (no branch per batch: loads are clamped, stores guarded)
            const int e = u * PM_WAVE + lane;
            q[u] = __builtin_nontemporal_load(src + (e < n ? e : n - 1));
        }
    };
    issue(t0);
    uint32_t JW[NG + 1];  // lane (k, t): word of step 4 g + t for quad k of every frame
#pragma unroll
    for (int g = 0; g <= NG; ++g) JW[g] = a.jobs[k * kMwStride + 4 * g + (lane & 3)];
    // a record's slot, the slot of the joint it is mapped to and of that joint's... the reference reads mapping[j] and mapping[parents[j]] (skeleton.py:322-331)
    int so[NB], mo[NB], mp[NB];
    bool is_root[NB];
    const float invJ = 1.0f / (float)J;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int e = u * PM_WAVE + lane, ec = e < ne ? e : ne - 1;
        const int ef = (FPW == 1) ? 0 : (int)(((float)ec + 0.5f) * invJ);  // ec / J, exact for ec < 2^22
        const int ej = ec - ef * J;
        so[u] = ef * FS + ej * 4;
        mo[u] = ef * FS + 4 * a.mapping.m[ej];
        mp[u] = ef * FS + 4 * a.mapping.m[ej == 0 ? 0 : a.parents.p[ej]];
        is_root[u] = ej == 0;
    }
#pragma unroll
    for (int g = 0; g <= NG; ++g) asm volatile("" : "+v"(JW[g]));  // settle the list here, not inside the walk (behind the next tile's loads)
#pragma unroll
    for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(mo[u]), "+v"(mp[u]));
    if (k == 0) { fD[J * 4 + c] = (c == 0) ? 1.0f : 0.0f; fD[(J + 1) * 4 + c] = (c == 0) ? 1.0f : 0.0f; }  // identity and idle slots (idle steps keep the idle one at the identity)
    const float f1 = (a.c0 == 1 || a.c1 == 1) ? -1.0f : 1.0f, f2 = (a.c0 == 2 || a.c1 == 2) ? -1.0f : 1.0f, f3 = (a.c0 == 3 || a.c1 == 3) ? -1.0f : 1.0f;

    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t f0 = tile * FPW;
        const int n = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW) * J;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = u * PM_WAVE + lane;
            const float qi[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
            float un[4];
            qnormalize(qi, 1e-8f, un);  // skeleton.py:45
            if (e < n) *reinterpret_cast<v4f *>(smem + so[u]) = v4f{un[0], un[1], un[2], un[3]};
        }
        wave_sync();
        if (tile + 1 < t1) issue(tile + 1);  // in flight while this tile walks
        mw_walk(fD, JW, a.nsteps, c);
        wave_sync();
        // finish, lane per record: the reference's sign and normalisation ONCE per world quaternion, in place ...
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = u * PM_WAVE + lane;
            float *slot = smem + so[u];
            const v4f gv = *reinterpret_cast<const v4f *>(slot);
            const float g[4] = {gv.x, gv.y, gv.z, gv.w};
            float cg[4];
            canonical_sign(g, cg);
            if (e < n) *reinterpret_cast<v4f *>(slot) = v4f{cg[0], cg[1], cg[2], cg[3]};
            if (u & 1) asm volatile("" ::: "memory");
        }
        wave_sync();
        // ... then the joint permutation, the two negated components and local'_j = conj(g'_parent) (x) g'_j, stored straight from registers
        v4f *gout = reinterpret_cast<v4f *>(a.out) + f0 * J;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = u * PM_WAVE + lane;
            const v4f gv = *reinterpret_cast<const v4f *>(smem + mo[u]), pv = *reinterpret_cast<const v4f *>(smem + mp[u]);
            const float cg[4] = {gv.x, gv.y * f1, gv.z * f2, gv.w * f3};  // skeleton.py:310-318
            const float inv[4] = {pv.x, -pv.y * f1, -pv.z * f2, -pv.w * f3};
            float o[4];
            qmul(inv, cg, o);  // skeleton.py:85-91 on the mirrored world rotations
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = is_root[u] ? cg[i] : o[i];  // the root has no parent (select, not branch)
            if (e < n) __builtin_nontemporal_store(v4f{o[0], o[1], o[2], o[3]}, gout + e);
            if (u & 1) asm volatile("" ::: "memory");
        }
        wave_sync();  // the image is the next tile's
    }
}

template <int FPW, int NB>
static int launch_mirror_wide(const MirrorWideArgs &a, const int nt, hipStream_t s) {
    const size_t lds = (size_t)FPW * mirror_wide_frame_stride(a.J) * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW, ngroups = (ntiles + nt - 1) / nt, grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("mirror: grid too large"); return PM_EUNSUPPORTED; }
    set_kernel_name("void pm::mirror_wide_kernel<%d, %d>(pm::MirrorWideArgs, int)", FPW, NB);
    auto kf = mirror_wide_kernel<FPW, NB>;
    if (int e = allow_lds(kf, lds)) return e;
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
    return PM_AFTER_LAUNCH("mirror launch");
}
template <int FPW>
static int launch_mirror_wide_nb(const MirrorWideArgs &a, const int nt, hipStream_t s) {
    const int ne = FPW * a.J;
    if (ne <= 2 * PM_WAVE) return launch_mirror_wide<FPW, 2>(a, nt, s);
    if (ne <= 3 * PM_WAVE) return launch_mirror_wide<FPW, 3>(a, nt, s);
    if (ne <= 4 * PM_WAVE) return launch_mirror_wide<FPW, 4>(a, nt, s);
    if (ne <= 6 * PM_WAVE) return launch_mirror_wide<FPW, 6>(a, nt, s);
    return launch_mirror_wide<FPW, 8>(a, nt, s);
}
// The kernel's step words for this tree at 16 / fpw joints a step: jobs[k * kMwStride + step] = own slot | parent slot << 16, in bytes of the 16-byte slots (the root
// takes no step: its children compose with its slot as parked; idle: slot J + 1 composed with the identity slot J).  Returns the number of steps, -1 beyond kMwSteps.
int mirror_wide_words(const Parents &par, const int J, const int fpw, uint32_t *jobs) {
    const int W = 16 / fpw;
    uint32_t list[(kMwSteps + 2) * 16];
    const int nsteps = (J == 1) ? 0 : fk_wide_plan(par, J, W, kMwSteps, false, list);
    if (nsteps < 0) return -1;
    const uint32_t idle = (uint32_t)((J + 1) * 16) | ((uint32_t)(J * 16) << 16);
    for (int k = 0; k < 16; ++k)
        for (int st = 0; st < kMwStride; ++st) {
            uint32_t w = idle;
            if (k < W && st < nsteps) {
                const uint32_t j = list[st * W + k] & 0xffffu, p = list[st * W + k] >> 16;
                if ((int)j < J) w = (j * 16u) | (p * 16u) << 16;
            }
            jobs[k * kMwStride + st] = w;
        }
    return nsteps;
}

// 16-byte aligned arrays, `fpw` = 1, 2, 4 or 8 frames a wave.  false (nothing launched): fpw x J records do not fit eight batches, the tree needs more than
// kMwSteps steps of 16 / fpw joints, or more than max_quad_steps_per_joint_x10 / 10 quad-steps per joint (0: no such bound); true with rc set otherwise.
static bool try_mirror_wide(const int fpw, const MirrorArgs &m, const int max_quad_steps_per_joint_x10, hipStream_t s, int &rc) {
    const int J = m.J;
    if ((fpw != 1 && fpw != 2 && fpw != 4 && fpw != 8) || fpw * J > 8 * PM_WAVE) return false;
    const int W = 16 / fpw;
    MirrorWideArgs a;
    a.nsteps = mirror_wide_words(m.parents, J, fpw, a.jobs);
    if (a.nsteps < 0) return false;
    if (max_quad_steps_per_joint_x10 > 0 && a.nsteps * W * 10 > max_quad_steps_per_joint_x10 * J) return false;
    a.rot = m.rot; a.out = m.out; a.F = m.F; a.J = J; a.c0 = m.c0; a.c1 = m.c1; a.parents = m.parents; a.mapping = m.mapping;
    const int64_t ntiles = (m.F + fpw - 1) / fpw;
    // two tiles a workgroup: a tile is 16 J fpw bytes in and out (2-6 KB), and a workgroup that writes 4-12 KB in one place leaves fewer cache lines shared with
    // its neighbours -- same-box sweep, one / two tiles: 48 joints 140 / 135 us, SMPL-H 170 / 157, 64 joints 197 / 181, 65 joints 244 / 204, 96 joints 330 / 279,
    // 128 joints 400 / 374, 512 joints 961 / 753 (four: 767); a clip of real length keeps one (the launch wants its workgroups)
    int nt = ntiles >= 32768 ? 2 : 1;
    nt = tune_env("PM_MW_NT", nt);
    if (nt < 1) nt = 1;
    rc = fpw == 1 ? launch_mirror_wide_nb<1>(a, nt, s) : (fpw == 2 ? launch_mirror_wide_nb<2>(a, nt, s) : (fpw == 4 ? launch_mirror_wide_nb<4>(a, nt, s) : launch_mirror_wide_nb<8>(a, nt, s)));
    return true;
}

static size_t mirror_lds_bytes(const int FPW, const int J, const int K, const int C) {
    return ((size_t)FPW * mirror_frame_stride(J, C) + 3 * (size_t)J + 1 + 8) * sizeof(float) + (C > 1 ? (size_t)(K + 2) * C * 8 + 8 : 0);  // + slack for the walk's look-ahead
}

template <int FPW, int C = 1>
static int launch_mirror(const MirrorArgs &a, bool vec, hipStream_t s) {
    const size_t lds = mirror_lds_bytes(FPW, a.J, a.K, C);
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("mirror: grid too large"); return PM_EUNSUPPORTED; }
    set_kernel_name("void pm::mirror_kernel<%d, %s, %d>(pm::MirrorArgs)", FPW, tf(vec), C);
    if (vec) {
        auto k = mirror_kernel<FPW, true, C>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    } else {
        auto k = mirror_kernel<FPW, false, C>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    }
    return PM_AFTER_LAUNCH("mirror launch");
}

}  // namespace pm

extern "C" int pm_mirror_rotations_f32(const float *rot, const int32_t *parents, const int32_t *mapping, int axis,
                                       int64_t F, int32_t J, float *out, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "mirror: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    PM_CHECK_ARGS(axis >= 0 && axis <= 2, "mirror: axis must be 0 (X), 1 (Y) or 2 (Z)");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(rot && parents && out, "mirror: null pointer");
    MirrorArgs a;
    a.rot = rot; a.out = out; a.F = F; a.J = J;
    a.c0 = (axis == 0) ? 2 : 1;  // skeleton.py:310-318: X -> (2,3), Y -> (1,3), Z -> (1,2)
    a.c1 = (axis == 2) ? 2 : 3;
    if (int e = pack_parents(parents, J, a.parents)) return e;
    for (int32_t j = 0; j < J; ++j) {
        const int32_t m = mapping ? mapping[j] : j;
        if (m < 0 || m >= J) { set_error("mirror: joints_mapping[%d] = %d out of range", j, m); return PM_EINVAL; }
        a.mapping.m[j] = (int16_t)m;
    }
    const bool vec = aligned16(rot) && aligned16(out);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (round 6) From kMirrorWideMinJ joints on: the walk from a step list in registers, 16 / fpw joints of a frame a step (mirror_wide_kernel), with or without a joint
    // mapping.  Same-box sweep (profiles/r06_mirror_wide_sweep.txt), this kernel / what ran before, us: random trees of 48 / 64 / 96 / 128 / 250 / 512 joints
    // 141 / 151, 196 / 210, 294 / 343, 374 / 462, 383 / 707, 753 / 4411 (beyond kSchedMaxJoints a wide tree fell to the one-chain walk); humanoids of 56 / 64 / 128 /
    // 250 / 512 joints against the lane-per-frame kernel 180 / 193, 191 / 225, 381 / 436, 373 / 467, 797 / 913; SMPL-H 168 / 183; chain-like skeletons a draw
    // (56: 194 / 194, 72: 260 / 250, 96: 326 / 335); 40 joints 116-124 / 125-131; below, sixteen frames a wave on the one-chain walk are a draw or ahead (22 joints
    // 124-137 / 128-132, 32: 191-203 / 185, 36: 111-113 / 112-115).
    // Frames a wave: four up to 100 joints, one beyond (96 joints: 279 us with four or two, 128 joints 374 with one against 391); a narrow tree (under a third of its quad-steps busy, or more steps than the list holds)
    // takes more frames and fewer joints a step, and what no width holds goes on to the kernels below.  PM_MIRROR_WIDE (PM_TUNING build only): 0 never,
    // 1 / 2 / 4 / 8 force that many frames a wave.
    if (const int wide = tune_env("PM_MIRROR_WIDE", -1); vec && wide != 0 && (wide > 0 || (J >= kMirrorWideMinJ && tune_env("PM_MIRROR_DEEP", -1) != 1 && tune_env("PM_MIRROR_CHAINS", -1) < 0 && tune_env("PM_MIRROR_FPW", 0) == 0))) {
        int rc = PM_OK;
        if (wide > 0) { if (try_mirror_wide(wide, a, 0, s, rc)) return rc; }
        else
            for (const int bound : {15, 30})  // (a width with two thirds of its quad-steps busy first, then one with a third: to_root_dq_impl, dq.hip)
                for (int fpw = J <= 100 ? 4 : 1; fpw <= 8; fpw *= 2)
                    if (try_mirror_wide(fpw, a, bound, s, rc)) return rc;
    }
    // mode 'all' on long skeletons: one lane per frame, joints streamed (mirror_deep_kernel), where the call has the joint-frames to fill the
    // chip (common.hpp).  From 66 joints on, and from 52 when the row is a whole number of 64-byte pieces (round 4, with the kernel's eight
    // waves two to a SIMD: SMPL-H as stored and a chain-like 52 at 2^18 / 2^20 frames 81 / 324 us against 89-95 / 339 us for the scheduled
    // walk; J = 40 a draw, J = 50 -- 8-byte row ends -- 9 % slower).  PM_MIRROR_DEEP (PM_TUNING build only): 0 = never, 1 = whenever eligible
    if (const int deep = tune_env("PM_MIRROR_DEEP", -1); vec && mapping == nullptr && deep != 0 && (deep == 1 || ((J >= kMirrorDeepMinJ || (J >= 52 && J % 4 == 0)) && lane_per_frame_pays(F, J, kMirrorDeepMinJointFrames)))) {
        MirrorDeepArgs da;
        if (deep_plan(a.parents, J, false, da.topo) >= 0) {
            da.rot = rot; da.out = out; da.F = F; da.J = J; da.c0 = a.c0; da.c1 = a.c1;
            return launch_mirror_deep(da, s);
        }
    }
    const size_t per_frame = (size_t)mirror_frame_stride(J, 1) * sizeof(float), fixed = (3 * (size_t)J + 9) * sizeof(float) + 256;
    int pick = (7 * (16 * per_frame + fixed) <= kMaxLds) ? 16 : 8;  // 4 lanes per frame; keep >= 7 waves per CU if possible
    // Several chains per frame where the tree is wide enough for the shorter walk to pay (walk cost per frame ~ steps x chains / 16),
    // the rule of to_root_dual_quat (dq.hip).  PM_MIRROR_CHAINS (PM_TUNING build only): 0 = one chain, 2 / 4.
    a.K = 0;
    if (const int chains = tune_env("PM_MIRROR_CHAINS", -1); (chains < 0 ? J >= kMirrorSchedMinJ : (chains == 2 || chains == 4)) && J <= kSchedMaxJoints) {
        uint8_t s2[kSchedMax], s4[kSchedMax];
        const int K2 = chains == 4 ? 0 : schedule_chains(a.parents, J, 2, s2, false), K4 = chains == 2 ? 0 : schedule_chains(a.parents, J, 4, s4, false);
        int use = 0;
        if (chains == 2) use = K2 ? 2 : 0;
        else if (chains == 4) use = K4 ? 4 : 0;
        else {
            const int c1 = 2 * J, c2 = K2 ? 2 * K2 : 1 << 30, c4 = K4 ? 4 * K4 : 1 << 30;
            // measured at 2^19 frames, one / two / four chains (% of the HBM spec): chain-like skeletons J = 36 66 / 63 / 47, 40 64 / 66.5 / 50,
            // 52 57 / 63 / 53, 64 55 / 62 / 53, 72 49 / 58 / 48, 96 50 / 53 / 47, 128 40 / 45 / 43; bushy random trees J = 40 64 / 66 / 58,
            // 52 57 / 63 / 62, 64 55 / 62 / 62, 72 49 / 57 / 56, 96 50 / 53 / 59, 128 41 / 46 / 58: two chains from 40 joints on, four
            // only for big trees wide enough to halve the walk again
            if (K4 && J >= 80 && 10 * c4 <= 12 * c2) use = 4;
            else if (K2 && 4 * c2 <= 3 * c1) use = 2;
        }
        if (use) {
            a.K = use == 2 ? K2 : K4;
            memcpy(a.sched, use == 2 ? s2 : s4, (size_t)a.K * use);
            if (mirror_lds_bytes(16 / use, J, a.K, use) <= kMaxLds) return use == 2 ? launch_mirror<8, 2>(a, vec, s) : launch_mirror<4, 4>(a, vec, s);
            a.K = 0;
        }
    }
    if (const int v = tune_env("PM_MIRROR_FPW", 0); v == 16 || v == 8 || v == 4) pick = v;  // PM_TUNING build only
    while (pick > 4 && pick * per_frame + fixed > kMaxLds) pick >>= 1;
    if (pick * per_frame + fixed <= kMaxLds) {
        if (pick == 16) return launch_mirror<16>(a, vec, s);
        if (pick == 8) return launch_mirror<8>(a, vec, s);
        return launch_mirror<4>(a, vec, s);
    }
    set_error("mirror: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
}
