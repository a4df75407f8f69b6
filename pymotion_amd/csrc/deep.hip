// deep.hip -- long skeletons: ONE LANE PER FRAME, the skeleton streamed through LDS a few joints at a time.
//
// The tile kernels of fk.hip / dq.hip keep a whole frame's output in LDS while its tree is walked (32-64 B per joint), and
// spread a frame over 3-16 lanes so that a wave's few frames still fill it.  That is the right shape up to ~60 joints.
// Beyond, the image of ONE frame is 2-8 KB, a CU holds 20-40 frames, and a frame's walk is a dependent chain of J / 2 ... J
// steps that nothing hides: 2^19 frames x 128 joints of a chain-like skeleton ran at 31 % (to_root_dual_quat) and 46 %
// (fk) of the HBM spec, and no schedule over more chains helps a skeleton that IS a chain.
//
// Here a frame is one lane's sequential loop -- 64 frames per wave whatever the topology -- and only a CHUNK of M = 8
// joints of the 64 frames is in LDS at a time (it is the transposition buffer between "lane = frame" and coalesced rows):
//     load   the chunk's M quaternions of 64 frames (rows of 128 B, one dwordx4 per lane and load) -- requested one chunk
//            ahead into registers, parked into the tail of the slots their results will overwrite;
//     walk   jj = 0 .. M-1: the lane reads its quaternion, composes it with its parent's state and writes the output
//            record into the slot.  The parent's state is the lane's own registers when parents[j] == j - 1; otherwise one of
//            kDeepSlots saved states, also registers: the host colours the live ranges [p, last child of p] of the
//            joints whose children do not follow them directly (a humanoid needs 1-3; a skeleton that needs more than
//            kDeepSlots stays on the tile kernels);
//     store  the chunk's rows (M x 32 B per frame, contiguous in HBM) leave as coalesced dwordx4.
// 17 KB of LDS per wave: eight waves per CU, 512 frames in flight per CU instead of 20-40.
//
// The state is FLOAT64 (quaternion and translation): the arithmetic of a lane-per-frame walk is a fifth of the chip time its
// bytes need even at the float64 rate, and it makes the result independent of the data's magnitude -- no big-tile test, no
// residuals, no fixed point: every output is the float32 rounding of the reference's own float64 chain
// (skeleton.py:207-244, dual_quat.py:12-36) up to 1e-15.
#include "common.hpp"

namespace pm {

// Colours the joints whose state a later, non-adjacent child needs onto kDeepSlots register sets.  `root_is_identity`:
// to_root_dual_quat's convention -- children of the root stay local (skeleton.py:236-237) and the root's own state is never a
// parent.  Returns the number of slots used, or -1 if the skeleton needs more than kDeepSlots.
int deep_plan(const Parents &par, const int J, const bool root_is_identity, DeepTopo &t) {
    int last_use[PM_MAX_JOINTS], slot_of[PM_MAX_JOINTS], busy_until[kDeepSlots], used = 0;
    for (int j = 0; j < J; ++j) { last_use[j] = -1; slot_of[j] = -1; }
    for (int j = 1; j < J; ++j) {
        const int p = par.p[j];
        if (p == j - 1 && !(root_is_identity && p == 0)) continue;
        if (root_is_identity && p == 0) continue;
        if (last_use[p] < j) last_use[p] = j;
    }
    for (int k = 0; k < kDeepSlots; ++k) busy_until[k] = -1;
    for (int j = 0; j < J; ++j) {
        const int p = (j == 0) ? -1 : par.p[j];
        int load, save = DEEP_NONE;
        if (j == 0) load = DEEP_ROOT;
        else if (root_is_identity && p == 0) load = DEEP_LOCAL;
        else if (p == j - 1) load = DEEP_CHAIN;
        else load = slot_of[p];
        if (last_use[j] >= 0) {  // a slot is free again once the joint that used it last has LOADED it: joint j may take it
            int k = 0;
            while (k < kDeepSlots && busy_until[k] > j) ++k;
            if (k == kDeepSlots) return -1;
            busy_until[k] = last_use[j];
            slot_of[j] = k;
            save = k;
            if (k + 1 > used) used = k + 1;
        }
        t.code[j] = load | (save << 8);
    }
    return used;
}

// ---- to_root_dual_quat -------------------------------------------------------------------------------------------------
struct DeepDqArgs {
    const float *rot;       // [F,J,4]
    const float *root_pos;  // [F,3]
    const float *offsets;   // [J,3]
    float *dq;              // [F,J,8]
    int64_t F;
    int32_t J;
    int32_t ablate;  // PM_TUNING build only (env PM_DEEP_ABLATE): 1 = the ring kernel skips its partial (edge) groups' stores
    DeepTopo topo;
};

constexpr int kDeepDqRow = kDeepM * 8 + 4;  // floats per frame and chunk; (row / 4) odd: the lanes (= frames) of a ds_*_b128 spread over all banks

// The saved parent states: kDeepSlots register sets picked by a wave-uniform index (compile-time recursion: an indexed array
// would live in scratch memory).
struct DeepSaves { double q[kDeepSlots][4], t[kDeepSlots][3]; };
template <int K>
__device__ __forceinline__ void deep_slot_load(const int ld, const DeepSaves &sv, double (&Qp)[4], double (&Tp)[3]) {
    if constexpr (K < kDeepSlots) {
        int code = ld;
        asm volatile("" : "+s"(code));  // an opaque copy per test: or the chain of tests is folded into sv.q[ld] -- an indexed array, i.e. scratch memory
        if (code == K) {
            Qp[0] = sv.q[K][0]; Qp[1] = sv.q[K][1]; Qp[2] = sv.q[K][2]; Qp[3] = sv.q[K][3];
            Tp[0] = sv.t[K][0]; Tp[1] = sv.t[K][1]; Tp[2] = sv.t[K][2];
        }
        deep_slot_load<K + 1>(ld, sv, Qp, Tp);
    }
}
template <int K>
__device__ __forceinline__ void deep_slot_save(const int st, DeepSaves &sv, const double (&Q)[4], const double (&T)[3]) {
    if constexpr (K < kDeepSlots) {
        int code = st;
        asm volatile("" : "+s"(code));
        if (code == K) {
            sv.q[K][0] = Q[0]; sv.q[K][1] = Q[1]; sv.q[K][2] = Q[2]; sv.q[K][3] = Q[3];
            sv.t[K][0] = T[0]; sv.t[K][1] = T[1]; sv.t[K][2] = T[2];
        }
        deep_slot_save<K + 1>(st, sv, Q, T);
    }
}

// One joint of the lane's frame: (Q, T) <- parent (x) (q, v), `out` = the joint's dual quaternion.  ld / st: DeepTopo codes
// (wave-uniform), rp = the frame's root position, sv = the saved parent states.
__device__ __forceinline__ void deep_dq_joint(const float (&q)[4], const float (&o)[4], const double (&rp)[3], const int ld, const int st,
                                              double (&Q)[4], double (&T)[3], DeepSaves &sv, float (&out)[8]) {
    double Qp[4], Tp[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) Qp[i] = Q[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Tp[i] = T[i];
    deep_slot_load<0>(ld, sv, Qp, Tp);
    double v[3] = {(double)o[0], (double)o[1], (double)o[2]};
    if (ld == DEEP_ROOT || ld == DEEP_LOCAL) {  // (rot_j, offsets[j]) as they are -- the root: (rot_0, global_pos): the identity as parent is exact
        Qp[0] = 1.0; Qp[1] = 0.0; Qp[2] = 0.0; Qp[3] = 0.0; Tp[0] = 0.0; Tp[1] = 0.0; Tp[2] = 0.0;
        if (ld == DEEP_ROOT) { v[0] = rp[0]; v[1] = rp[1]; v[2] = rp[2]; }
    }
    const double b[4] = {(double)q[0], (double)q[1], (double)q[2], (double)q[3]};
    // rotations[j] = quat.mul(rotations[parent], rotations[j])                                     skeleton.py:241
    Q[0] = __builtin_fma(-Qp[3], b[3], __builtin_fma(-Qp[2], b[2], __builtin_fma(-Qp[1], b[1], Qp[0] * b[0])));
    Q[1] = __builtin_fma(-Qp[3], b[2], __builtin_fma(Qp[2], b[3], __builtin_fma(b[0], Qp[1], Qp[0] * b[1])));
    Q[2] = __builtin_fma(-Qp[1], b[3], __builtin_fma(Qp[3], b[1], __builtin_fma(b[0], Qp[2], Qp[0] * b[2])));
    Q[3] = __builtin_fma(-Qp[2], b[1], __builtin_fma(Qp[1], b[2], __builtin_fma(b[0], Qp[3], Qp[0] * b[3])));
    // translations[j] = quat.mul_vec(rotations[parent], translations[j]) + translations[parent]      :238-240, quat.py:320-334
    double t[3];
    t[0] = __builtin_fma(Qp[2], v[2], -(Qp[3] * v[1]));
    t[1] = __builtin_fma(Qp[3], v[0], -(Qp[1] * v[2]));
    t[2] = __builtin_fma(Qp[1], v[1], -(Qp[2] * v[0]));
    t[0] += t[0]; t[1] += t[1]; t[2] += t[2];
    T[0] = (__builtin_fma(Qp[0], t[0], v[0]) + __builtin_fma(Qp[2], t[2], -(Qp[3] * t[1]))) + Tp[0];
    T[1] = (__builtin_fma(Qp[0], t[1], v[1]) + __builtin_fma(Qp[3], t[0], -(Qp[1] * t[2]))) + Tp[1];
    T[2] = (__builtin_fma(Qp[0], t[2], v[2]) + __builtin_fma(Qp[1], t[1], -(Qp[2] * t[0]))) + Tp[2];
    // [q_r, 0.5 (0, t) (x) q_r]                                                                       dual_quat.py:28-36
    // (quat.mul with the scalar part 0.0 of the pure quaternion written out, quat.py:337-361: 0 x Inf is what makes the reference's
    // dual part NaN where its rotation has overflowed, and the pattern is part of the contract)
    const double h[4] = {0.0, 0.5 * T[0], 0.5 * T[1], 0.5 * T[2]};
    const float res[8] = {
        (float)Q[0], (float)Q[1], (float)Q[2], (float)Q[3],
        (float)__builtin_fma(-h[3], Q[3], __builtin_fma(-h[2], Q[2], __builtin_fma(-h[1], Q[1], h[0] * Q[0]))),
        (float)__builtin_fma(-h[3], Q[2], __builtin_fma(h[2], Q[3], __builtin_fma(Q[0], h[1], h[0] * Q[1]))),
        (float)__builtin_fma(-h[1], Q[3], __builtin_fma(h[3], Q[1], __builtin_fma(Q[0], h[2], h[0] * Q[2]))),
        (float)__builtin_fma(-h[2], Q[1], __builtin_fma(h[1], Q[2], __builtin_fma(Q[0], h[3], h[0] * Q[3])))};
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = res[i];
    deep_slot_save<0>(st, sv, Q, T);
}

__global__ __launch_bounds__(PM_WAVE) void to_root_dq_deep_kernel(const DeepDqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int M = kDeepM, RS = kDeepDqRow;
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t tile = xcd_tile((a.F + PM_WAVE - 1) / PM_WAVE);
    if (tile < 0) return;
    float *sImg = smem;                 // [64][RS]
    float *sOff = smem + PM_WAVE * RS;  // [J][4]
    for (int j = lane; j < J; j += PM_WAVE) {
        float *o = sOff + 4 * j;
        if (PM_LDS_OK(o, 16u)) { o[0] = a.offsets[3 * j]; o[1] = a.offsets[3 * j + 1]; o[2] = a.offsets[3 * j + 2]; o[3] = 0.0f; }
    }
    const int64_t f0 = tile * PM_WAVE;
    const int nf = (int)((a.F - f0) < PM_WAVE ? (a.F - f0) : PM_WAVE);
    const int nchunks = (J + M - 1) / M;
    const float *grot = a.rot + f0 * J * 4;
    float *gout = a.dq + f0 * J * 8;

    // the chunk's quaternions, one 128-byte row segment per frame: lane (8 fr' + jj) of load u reads joint j0 + jj of frame 8 u + fr'
    v4f pre[M];
    auto issue = [&](const int c) {
        const int j0 = c * M, mj = (J - j0) < M ? (J - j0) : M;
        const int jj = lane & (M - 1), jc = jj < mj ? jj : mj - 1;  // past the skeleton's end: a valid record again, parked where nobody reads
#pragma unroll
        for (int u = 0; u < M; ++u) {
            const int fr = u * (PM_WAVE / M) + (lane >> 3), fc = fr < nf ? fr : nf - 1;
            pre[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(grot + ((int64_t)fc * J + j0 + jc) * 4));
        }
    };
    const int fl = lane < nf ? lane : nf - 1;
    double rp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rp[k] = (double)a.root_pos[(f0 + fl) * 3 + k];
    asm volatile("" : "+v"(rp[0]), "+v"(rp[1]), "+v"(rp[2]));  // settled HERE: its first use is inside the walk, where the wait would be one for every prefetch in flight
    issue(0);

    double Q[4] = {1.0, 0.0, 0.0, 0.0}, T[3] = {0.0, 0.0, 0.0};  // state of the joint this lane composed last
    DeepSaves sv = {};
    float *row = sImg + lane * RS;
    wave_sync();

    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < M; ++u) {
            float *p = sImg + (u * (PM_WAVE / M) + (lane >> 3)) * RS + (lane & (M - 1)) * 8 + 4;
            if (PM_LDS_OK(p, 16u)) *reinterpret_cast<v4f *>(p) = pre[u];
        }
    };
    auto walk = [&](const int j0, const int mj) {
#pragma unroll 1
        for (int jj = 0; jj < mj; ++jj) {
            const int j = j0 + jj;
            const int code = __builtin_amdgcn_readfirstlane(a.topo.code[j]), ld = code & 0xff, st = code >> 8;  // wave-uniform (kernarg)
            float *slot = row + jj * 8;
            float q[4], o[4];
            lds_get<4>(slot, 1, q);
            lds_get<4>(sOff, j, o);
            float out[8];
            deep_dq_joint(q, o, rp, ld, st, Q, T, sv, out);
            lds_put<8>(slot, 0, out);
        }
    };
    // Rows of mj x 32 B per frame, contiguous in HBM.  Unit = 16 bytes = half a slot; lane (fr', k) of store u writes place k of
    // frame 4 u + fr'.  Frames past a partial tile repeat the last frame (the same bytes to the same address), so that the stores
    // of a full chunk are UNCONDITIONAL: the wait for the next chunk's quaternions, requested before them, is then a counted
    // vmcnt(16 + ...) -- behind predicated stores (or a loop of them) it is a wait for every store of this chunk to be acknowledged.
    const int st_k = lane & 15, st_fr = lane >> 4;
    auto st_lds = [&](const int u) { const int fr = 4 * u + st_fr; return sImg + (fr < nf ? fr : nf - 1) * RS + 4 * st_k; };
    auto st_glb = [&](const int u, const int j0) { const int fr = 4 * u + st_fr; return gout + ((fr < nf ? fr : nf - 1) * J + j0) * 8 + 4 * st_k; };  // < 2^21 floats into the tile
    auto store_full = [&](const int j0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {  // four batches of four: the LDS reads of a batch are in flight together
            v4f r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float *p = st_lds(4 * h + u); r[u] = PM_LDS_OK(p, 16u) ? *reinterpret_cast<const v4f *>(p) : v4f{0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(r[u], reinterpret_cast<v4f *>(st_glb(4 * h + u, j0)));
        }
    };
    auto store_part = [&](const int j0, const int mj) {
#pragma unroll
        for (int u = 0; u < 2 * M; ++u) {
            const float *p = st_lds(u);
            if (st_k < 2 * mj && PM_LDS_OK(p, 16u)) __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(p), reinterpret_cast<v4f *>(st_glb(u, j0)));
        }
    };
    park();
    const int nfull = J / M;  // chunks of M joints; the last chunk of the skeleton may be shorter
    for (int c = 0; c + 1 < nchunks; ++c) {
        issue(c + 1);  // in flight while this chunk is walked and stored
        wave_sync();
        walk(c * M, M);
        wave_sync();
        store_full(c * M);
        wave_sync();  // the slots are reused
        park();
    }
    wave_sync();
    {
        const int c = nchunks - 1, mj = J - c * M;
        walk(c * M, mj);
        wave_sync();
        if (c < nfull) store_full(c * M);
        else store_part(c * M, mj);
    }
}

// ---- the same walk for rows that do not start on cache lines ----------------------------------------------------------
// A frame's rows start at f J 16 B (in) / f J 32 B (out): unless J is a multiple of 8, the chunk segments above straddle
// 128-byte lines on both sides and every line is requested and written in two pieces (measured at 2^19 frames, chain-like
// skeletons: J = 64 54 % of the HBM spec, 68 49 %, 66 44 %, 65 32 %).  Here the segments are cut where the LINES are: the
// tile is one flat array of records g = f J + j, a frame's GROUP k holds its records with g >> 2 == (f J >> 2) + k (four
// records = 128 B out, 64 B in; the first and last group of a frame are partial), and the walk stays wave-uniform by lagging:
//     step c:  park group c (requested a step ahead), walk joints 4 c - 3 .. 4 c, store group c - 1
// -- joint j of frame f sits in group (j + s_f) >> 2, s_f = f J & 3, so after group c is parked every frame has its joints up
// to 4 c, and group c - 1 is complete in every frame once joint 4 c - 1 is done.  The slots are a ring of two groups (eight
// slots: the LDS of the chunk kernel); a lane's slot for joint j is (j + s_f) & 7.
__global__ __launch_bounds__(PM_WAVE) void to_root_dq_ring_kernel(const DeepDqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int G = 4, RS = kDeepDqRow;
    static_assert(RS >= 2 * G * 8 + 4, "ring of two groups");
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t tile = xcd_tile((a.F + PM_WAVE - 1) / PM_WAVE);
    if (tile < 0) return;
    float *sImg = smem;                 // [64][RS]
    float *sOff = smem + PM_WAVE * RS;  // [J][4]
    for (int j = lane; j < J; j += PM_WAVE) {
        float *o = sOff + 4 * j;
        if (PM_LDS_OK(o, 16u)) { o[0] = a.offsets[3 * j]; o[1] = a.offsets[3 * j + 1]; o[2] = a.offsets[3 * j + 2]; o[3] = 0.0f; }
    }
    const int64_t f0 = tile * PM_WAVE;  // a multiple of 64: (f0 + fr) J & 3 == fr J & 3
    const int nf = (int)((a.F - f0) < PM_WAVE ? (a.F - f0) : PM_WAVE);
    const int ngroups = ((J + 2) >> 2) + 1;
    const float *grot = a.rot + f0 * J * 4;
    float *gout = a.dq + f0 * J * 8;

    // loads: lane = (frl, jj) = (lane >> 2, lane & 3); load u covers frame 16 u + frl, whose s is that of frl (16 J = 0 mod 4)
    const int l_frl = lane >> 2, l_d = (lane & 3) - ((l_frl * J) & 3);  // joint of this lane's position in group c: 4 c + l_d
    v4f pre[G];
    auto issue = [&](const int c) {
        int j = 4 * c + l_d;
        j = j < 0 ? 0 : (j > J - 1 ? J - 1 : j);  // outside the frame: a valid record again, parked where nobody reads
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int fr = 16 * u + l_frl, fc = fr < nf ? fr : nf - 1;
            pre[u] = *reinterpret_cast<const v4f *>(grot + ((int64_t)fc * J + j) * 4);  // (not nontemporal: the other half of the line is the next step's)
        }
    };
    // stores: lane = (frl, k) = (lane >> 3, lane & 7), 16-byte half k & 1 of position k >> 1; unit u covers frame 8 u + frl
    const int s_frl = lane >> 3, s_d = ((lane & 7) >> 1) - ((s_frl * J) & 3);
    const int fl = lane < nf ? lane : nf - 1;
    double rp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rp[k] = (double)a.root_pos[(f0 + fl) * 3 + k];
    asm volatile("" : "+v"(rp[0]), "+v"(rp[1]), "+v"(rp[2]));  // settled before the first prefetch (see the chunk kernel)
    issue(0);
    double Q[4] = {1.0, 0.0, 0.0, 0.0}, T[3] = {0.0, 0.0, 0.0};
    DeepSaves sv = {};
    float *row = sImg + lane * RS;
    const int sf = (lane * J) & 3;
    wave_sync();

    auto park = [&](const int c) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            float *p = sImg + (16 * u + l_frl) * RS + ((c & 1) * G + (lane & 3)) * 8 + 4;
            if (PM_LDS_OK(p, 16u)) *reinterpret_cast<v4f *>(p) = pre[u];
        }
    };
    auto walk = [&](const int c) {
        const int jlo = 4 * c - 3 < 0 ? 0 : 4 * c - 3, jhi = 4 * c > J - 1 ? J - 1 : 4 * c;
#pragma unroll 1
        for (int j = jlo; j <= jhi; ++j) {
            const int code = __builtin_amdgcn_readfirstlane(a.topo.code[j]), ld = code & 0xff, st = code >> 8;  // wave-uniform (kernarg)
            float *slot = row + ((j + sf) & 7) * 8;
            float q[4], o[4];
            lds_get<4>(slot, 1, q);
            lds_get<4>(sOff, j, o);
            float out[8];
            deep_dq_joint(q, o, rp, ld, st, Q, T, sv, out);
            lds_put<8>(slot, 0, out);
        }
    };
    // group k leaves: lane (frl, place) of store u writes 16 bytes of frame 8 u + frl.  In a FULL tile the interior groups (every
    // place a joint of its frame, whatever the frame's shift) are stored unconditionally -- see the chunk kernel: the wait for the
    // next group's quaternions stays a counted one.  (A partial tile -- the last of a launch -- predicates everything: a frame past
    // its end has no stand-in here, its neighbour's rows are shifted differently.)
    auto st_lds = [&](const int u, const int k) { return sImg + (8 * u + s_frl) * RS + ((k & 1) * G) * 8 + (lane & 7) * 4; };
    auto st_glb = [&](const int u, const int j) { return gout + ((8 * u + s_frl) * J + j) * 8 + (lane & 1) * 4; };
    auto store_interior = [&](const int k) {
        const int j = 4 * k + s_d;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4f r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float *p = st_lds(4 * h + u, k); r[u] = PM_LDS_OK(p, 16u) ? *reinterpret_cast<const v4f *>(p) : v4f{0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(r[u], reinterpret_cast<v4f *>(st_glb(4 * h + u, j)));
        }
    };
    auto store_edge = [&](const int k) {
        const int j = 4 * k + s_d;
        const bool jok = j >= 0 && j < J;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float *p = st_lds(u, k);
            if (jok && !PM_ABLATED(a, 1) && 8 * u + s_frl < nf && PM_LDS_OK(p, 16u)) *reinterpret_cast<v4f *>(st_glb(u, jok ? j : 0)) = *reinterpret_cast<const v4f *>(p);  // (not nontemporal: the rest of a partial line belongs to the neighbouring frame and comes a tile's walk later)
        }
    };
    const int kint = (J - 4) >> 2;  // groups 1 .. kint are interior
    int cend = kint + 2 < ngroups - 1 ? kint + 2 : ngroups - 1;  // steps 2 .. cend - 1: interior store, another group to fetch
    if (nf < PM_WAVE) cend = 0;
    park(0);
    int c = 0;
    for (; c < 2 && c <= ngroups; ++c) {  // the first two steps: nothing, then the frames' partial first groups to store
        if (c + 1 < ngroups) issue(c + 1);
        wave_sync();
        walk(c);
        wave_sync();
        if (c >= 1) store_edge(c - 1);
        wave_sync();
        if (c + 1 < ngroups) park(c + 1);
    }
    for (; c < cend; ++c) {
        issue(c + 1);  // in flight while this step walks and stores
        wave_sync();
        walk(c);
        wave_sync();
        store_interior(c - 1);
        wave_sync();  // the ring half of group c - 1 is parked again now
        park(c + 1);
    }
    for (; c <= ngroups; ++c) {
        if (c + 1 < ngroups) issue(c + 1);
        wave_sync();
        walk(c);
        wave_sync();
        if (nf == PM_WAVE && c - 1 >= 1 && c - 1 <= kint) store_interior(c - 1);
        else store_edge(c - 1);
        wave_sync();
        if (c + 1 < ngroups) park(c + 1);
    }
}

size_t deep_dq_lds_bytes(const int J) { return ((size_t)PM_WAVE * kDeepDqRow + 4 * (size_t)J) * sizeof(float); }

// The caller has checked: pointers 16-byte aligned, deep_plan(...) >= 0.
int launch_to_root_deep(const float *rot, const float *root_pos, const float *offsets, float *dq, const int64_t F, const int32_t J,
                        const DeepTopo &topo, hipStream_t s) {
    DeepDqArgs a;
    a.rot = rot; a.root_pos = root_pos; a.offsets = offsets; a.dq = dq; a.F = F; a.J = J; a.topo = topo;
    a.ablate = tune_env("PM_DEEP_ABLATE", 0);
    const int64_t ntiles = (F + PM_WAVE - 1) / PM_WAVE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("to_root_dq: grid too large"); return PM_EUNSUPPORTED; }
    const size_t lds = deep_dq_lds_bytes(J);
    // rows that start on cache lines in every frame (J a multiple of 8): plain chunks; otherwise the line-aligned ring
    const bool ring = tune_env("PM_DQ_DEEP_RING", (J % kDeepM) != 0) != 0;  // PM_TUNING build only
    set_kernel_name(ring ? "pm::to_root_dq_ring_kernel(pm::DeepDqArgs)" : "pm::to_root_dq_deep_kernel(pm::DeepDqArgs)");
    auto kf = ring ? to_root_dq_ring_kernel : to_root_dq_deep_kernel;
    if (int e = allow_lds(to_root_dq_ring_kernel, lds)) return e;
    if (int e = allow_lds(to_root_dq_deep_kernel, lds)) return e;
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH("to_root_dq (deep) launch");
}

// (fk in the same shape was built and measured in round 3, and removed: its 12- and 36-byte records leave a chunk of eight joints as
// 96- and 288-byte PIECES of cache lines -- no joint count aligns them -- and the chip writes such pieces at 1.7-2.7 TB/s: 2^19 x 128
// frames 1438 us against 1163 us for the pipelined tile kernel, 367 us with the stores ablated.  Line-sized pieces need 32 records of
// a frame in LDS at once: 1.5 KB per frame, the tile kernels' problem again.  fk beyond 64 joints got a six-records-per-lane
// variant of the pipelined tile kernel instead, fk.hip.)

}  // namespace pm
