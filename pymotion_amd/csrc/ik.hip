// ik.hip -- from_root_positions (pymotion/ops/skeleton.py:96-170): joint positions -> local rotations.
//
// The reference aligns joints one at a time and re-runs a full fk (plus a matrix->quaternion pass) for every
// joint that has children and once more for every extra child: O(J^2) joint transforms per frame through
// ~6 J full-size temporaries.  The quantities it extracts from those fk calls have closed forms:
//   * global_rots[:, j] while joint j still holds the identity  = world rotation of parent(j)  (G_pre);
//   * rest direction  inverse(G_pre) (pos_c - pos_j)            = offsets[c]                    (exactly);
//   * after rotations[j] is set, world rotation of j            = G_pre (x) rot_j,  rest dir of a further child gc = offsets[gc].
// So one walk over the joints in index order (parents first) with the world quaternions of finished joints in
// LDS does the same thing in O(J):  one lane per frame,
//     G_j = G_pre (x) from_to(offsets[c0], inv(G_pre) (P_c0 - P_j))                                   (:136-141)
//     for each further child gc:
//         G_j = G_j (x) from_to_axis(offsets[gc], inv(G_j)(P_gc - P_j), inv(G_j) normalize(P_c0 - P_j))   (:147-168)
//     (the reference's fk re-normalises its local rotations, skeleton.py:45; here from_to returns exactly unit
//     quaternions);  joints without children keep the identity (:126-130).
// HBM traffic: 12 J B/frame in, 16 J out, both as coalesced one-record-per-lane streams.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace pm {

constexpr int kIkFourChainsMinJ = 56;
constexpr int64_t kIkClipFrames = 32768;  // a clip of real length: see the four-chain dispatch in pm_from_root_positions_f32

#ifndef PM_IK_MINW
#define PM_IK_MINW 4
#endif

struct Topo16 {  // kernarg: parents and children in CSR form
    int16_t parent[PM_MAX_JOINTS];
    int16_t cstart[PM_MAX_JOINTS + 1];
    int16_t clist[PM_MAX_JOINTS];
};

struct IkArgs {
    const float *pos;      // [F,J,3] root-centred joint positions
    const float *offsets;  // [J,3]
    float *out;            // [F,J,4] local rotations
    int64_t F;
    int32_t J;
    int32_t ablate;  // PM_TUNING build only (env PM_IK_ABLATE): 1 = no walk, 2 = no final pass, 4 = no position staging
    int32_t K;       // two chains per frame: steps of the schedule (0: one chain, every joint with children in index order)
    Topo16 topo;
    uint8_t sched[512];  // [K][C] joint aligned at step st by chain c, 255 = idle
};

// LDS image: ONE 16-byte slot per (frame, joint).  It holds the joint's position until the joint has been
// aligned (positions are only read while processing the joint itself or its parent, and parents come first)
// and the joint's world quaternion G_j afterwards -- 16 J + 16 B per frame, so 64 frames (every lane of the
// wave) fit in 23 KiB at J = 22 and six such waves share a CU.  The local rotations are not staged at all:
// G_j = G_pre (x) rot_j/(|rot_j| + 1e-8), so the final lane-per-(frame, joint) pass recovers
// rot_j = conj(G_parent) (x) G_j (the from_global_rotations gather; equal to the reference's value up to the
// ~1e-7 by which from_to's output misses unit length) and stores it straight from registers, coalesced.
// Joints without children keep the exact identity (:126-130).
__host__ __device__ constexpr int ik_frame_stride(const int J) { return 4 * ((J + 1) | 1); }  // (stride / 4) odd: the lanes (= frames) of a ds_read_b128 spread over all banks
__host__ __device__ constexpr int ik_tables_floats(const int J) { return 8 * (J + 1) + ((3 * J + 4 + 3) & ~3); }  // sOff [8 (J + 1)] + sTopo [3 J + 4], padded to 16 bytes

// NL > 0: the pipelined form.  Loading the positions, walking and storing the rotations are three phases of comparable
// length (2^20 x 22: 47 + 91 + 70 us when run alone) and a wave does them one after the other; with 23 KiB of image only six
// waves share a CU, too few for the phases of different waves to overlap, so the kernel ran at their SUM.  Here a
// workgroup owns `nt` consecutive tiles and all of a tile's position records (<= NL per lane) are requested into
// registers BEFORE the previous tile is walked: loads fly during the walk, and the rotations of the previous tile drain
// (fire-and-forget stores) while the next tile is parked and walked.  NL = 0: one tile per workgroup, batched loads
// (skeletons whose tile does not fit the register file).
// C = chains per frame.  The walk of a frame is a dependent chain that only occupancy hides (above), and the image bounds
// occupancy at 64 frames per wave.  With C = 2 a wave holds 32 frames and lanes 32..63 walk a SECOND chain of the same
// frames: the host schedules the joints that have children onto two chains (ik_schedule: a joint is ready two steps after
// its parent, or right after it on the parent's own chain, where the parent's quaternion is still in registers), the
// image and with it the LDS per wave halve, twice as many waves are resident, and a frame's walk is K ~ items / 2 steps.
template <int FPW, bool VEC, int NL, int C>
__global__ __launch_bounds__(PM_WAVE, (NL > 0 && NL <= 12) ? PM_IK_MINW : ((NL > 12 && NL <= 28) ? 2 : 1)) void from_root_positions_kernel(const IkArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t group = xcd_tile((ntiles + nt - 1) / nt);
    if (group < 0) return;
    const int FS = ik_frame_stride(J);
    float *sS = smem;                         // [FPW * FS]  slot (f, j): position, then world quaternion
    float *sOff = sS + FPW * FS;              // [(J + 1) * 8]  rest offset u of a joint {u0, u1, u2, 1 / |u| | u / |u|, |u|} (entry J: the idle item's zeros)
    int *sTopo = reinterpret_cast<int *>(sOff + 8 * (J + 1));  // [J] parent | [J+1] cstart | [J] clist
    typedef int v4i __attribute__((ext_vector_type(4)));
    // [(J + 2) * C] the walk's program: {joint, parent, first child, first further child | count << 16}; read and written as
    // 16-byte words, so the tables in front of it (6 J + 7 words, an odd count) are rounded up to a 16-byte boundary
    v4i *sItem = reinterpret_cast<v4i *>(sOff + ik_tables_floats(J));
    const float invJ = 1.0f / (float)J;
    constexpr int NR = NL > 0 ? NL : 1;
    v3f_a4 pre[NR];  // NL > 0: the next tile's position records, one 12-byte record per lane and load (coalesced dwordx3)
    auto issue = [&](const int64_t tile) {
        const int64_t f0 = tile * FPW;
        const int n = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW) * J;
        const float *g = a.pos + f0 * J * 3;
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int e = u * PM_WAVE + lane;
            if (u * PM_WAVE < n) pre[u] = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(g + 3 * (e < n ? e : n - 1)));
        }
    };
    if constexpr (NL > 0) issue(group * nt);
    for (int j = lane; j <= J; j += PM_WAVE) {  // rest offsets as they are (exact inputs of the cross / dot products below), their length once
        const float o[3] = {j < J ? a.offsets[3 * j] : 0.0f, j < J ? a.offsets[3 * j + 1] : 0.0f, j < J ? a.offsets[3 * j + 2] : 0.0f};
        const float u2 = __builtin_fmaf(o[0], o[0], __builtin_fmaf(o[1], o[1], o[2] * o[2]));
        const float iu = (u2 > 0.0f) ? __builtin_amdgcn_rsqf(u2) : 0.0f;
        *reinterpret_cast<v4f *>(sOff + 8 * j) = v4f{o[0], o[1], o[2], iu};
        *reinterpret_cast<v4f *>(sOff + 8 * j + 4) = v4f{o[0] * iu, o[1] * iu, o[2] * iu, fsqrt(u2)};
    }
    for (int j = lane; j <= J; j += PM_WAVE) {
        // parent | leaf << 16: what the final pass needs about a joint, in one word
        if (j < J) { sTopo[j] = (int)a.topo.parent[j] | ((a.topo.cstart[j + 1] == a.topo.cstart[j]) ? 0x10000 : 0); sTopo[2 * J + 1 + j] = a.topo.clist[j]; }
        sTopo[J + j] = a.topo.cstart[j];
    }
    // The walk visits the joints that have children, in index order.  Its topology reads are wave-uniform but DEPENDENT
    // (child range -> first child -> that child's slot): three LDS round trips in a row per joint, with 1.5 waves per SIMD
    // to hide them.  They are flattened once into one record per visited joint, which the walk reads two steps ahead.
    int nitems_all = 0;  // wave-uniform: steps of the walk
    const v4i idle = v4i{J, -1, J, 0};  // a scratch slot aligned with itself: the identity, stored where nobody looks
    if constexpr (C == 1) {
        for (int j0 = 0; j0 < J; j0 += PM_WAVE) {  // compaction: ballot + prefix popcount
            const int j = j0 + lane;
            const int cs = (j < J) ? a.topo.cstart[j] : 0, ce = (j < J) ? a.topo.cstart[j + 1] : 0;
            const bool has = ce > cs;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
            if (has) sItem[nitems_all + __popcll(m & ((1ull << lane) - 1ull))] =
                v4i{j, (j == 0) ? -1 : (int)a.topo.parent[j], (int)a.topo.clist[cs], (cs + 1) | ((ce - cs - 1) << 16)};
            nitems_all += __popcll(m);
        }
        if (lane < 2) sItem[nitems_all + lane] = idle;  // slack for the look-ahead
    } else {
        nitems_all = a.K;
        for (int i = lane; i < (a.K + 2) * C; i += PM_WAVE) {
            const int j = (i < a.K * C) ? (int)a.sched[i] : 255;
            v4i e = idle;
            if (j != 255) {
                const int cs = a.topo.cstart[j], ce = a.topo.cstart[j + 1];
                e = v4i{j, (j == 0) ? -1 : (int)a.topo.parent[j], (int)a.topo.clist[cs], (cs + 1) | ((ce - cs - 1) << 16)};
            }
            sItem[i] = e;
        }
    }
  for (int64_t tile = group * nt; tile < ntiles && tile < (group + 1) * nt; ++tile) {
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int n = nf * J;
    // positions -> the 16-byte slots of the image
    int lane_park = lane;
    auto park = [&](const int e0, const v3f_a4 pv) {  // record e0 + lane
        const int e = e0 + lane_park;
        const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
        const int j = e - f * J;
        if (lane_park < n - e0) { float *sl = sS + f * FS + 4 * j; sl[0] = pv.x; sl[1] = pv.y; sl[2] = pv.z; }  // (lane against a scalar: no threshold register per record)
    };
    if constexpr (NL > 0) {
        if (!PM_ABLATED(a, 4)) {
            asm volatile("" : "+v"(lane_park));  // the records' slots are recomputed per tile (six instructions each): held across the walk, NL addresses cost the register that spills
#pragma unroll
            for (int u = 0; u < NR; ++u)
                if (u * PM_WAVE < n) park(u * PM_WAVE, pre[u]);
        }
        if (tile + 1 < ntiles && tile + 1 < (group + 1) * nt) issue(tile + 1);  // in flight during this tile's walk
    } else {
        // two batches of 8 loads per lane (12 KiB per wave) in flight before the first one is parked, batch k+2 requested
        // before batch k is consumed (four loads at a time -- one memory latency per 256 records -- cost 10 %)
        const float *g = a.pos + f0 * J * 3;
        constexpr int NB = 8, BT = NB * PM_WAVE;
        auto load_b = [&](const int e0, v3f_a4 (&pv)[NB]) {
            if (e0 >= n) return;  // wave-uniform
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
                pv[u] = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(g + 3 * ec));
            }
        };
        auto park_b = [&](const int e0, const v3f_a4 (&pv)[NB]) {
            if (e0 >= n) return;
#pragma unroll
            for (int u = 0; u < NB; ++u) park(e0 + u * PM_WAVE, pv[u]);
        };
        v3f_a4 pa[NB], pb[NB];
        load_b(0, pa);
        load_b(BT, pb);
        for (int e0 = PM_ABLATED(a, 4) ? n : 0; e0 < n; e0 += 2 * BT) {
            park_b(e0, pa);
            load_b(e0 + 2 * BT, pa);
            park_b(e0 + BT, pb);
            load_b(e0 + 3 * BT, pb);
        }
    }
    wave_sync();

    // ---- the walk: one lane per frame ------------------------------------------------------------------------
    // Only WORLD quaternions are produced here (the final pass recovers the local ones).  Both alignment primitives of the
    // reference are evaluated from the SAME four quantities of the rest offset u (an exact input) and the predicted direction
    // v (the child's offset turned into the parent's frame), none of which loses digits when u and v are close to
    // (anti-)parallel -- which is where the reference's own result is most sensitive and round 2's fp32 evaluation fell
    // apart (5e-2 on a 20 000 x 52 batch, 2e-4 on the 4 099-frame test):
    //     cr = u x v      Kahan's difference of products: 1.5 ulp of each component however much cancels
    //     dt = u . v,  N = |u| |v|
    //     N + dt and N - dt: the one that does not cancel as it stands, the other as |cr|^2 / (that one)
    //   from_to(u, v) (quat.py:504-576) = (sqrt((1 + dot) / 2), sqrt((1 - dot) / 2) normalize(u^ x v^)) and from_to_axis(u, v, axis)
    //   (:579-650) = the same (w, s) about a given axis, s signed by cr . axis, are then the reference's own formulas with
    //   1 +- dot = (N +- dt) / N taken from those -- including the shrink of its dot by the two "+ 1e-8" normalisations, which
    //   an fp32 evaluation cannot see in dot itself but which moves the result by 1e-6 ... 1e-5 at small angles (measured: the
    //   closed form WITHOUT it, in float64, is 5e-5 off the reference at p99.9 on a 52-joint batch; with it 4e-6).  (What is NOT
    //   carried: the reference's axis normalize(u^ x v^) being short of unit length by 1e-8 / sin(angle) -- 1e-5 only within
    //   0.06 degrees of anti-parallel, where the answer moves by more than that with the last bit of the input.)
    //   Their special cases on np.isclose(dot, +-1) are tests of N (1 -+ dot) against 1.001e-5 N: dot ~ 1 snaps to the identity
    //   (:551-552), dot ~ -1 takes the rare branch (:554-571); a zero-length v gives the identity.
    //   Roll about a further child: the axis the reference derives, inv(G_j) normalize(P_c0 - P_j), IS the rest direction of the
    //   first child, which the alignment just mapped there -- except where that alignment snapped to the identity or took
    //   the anti-parallel branch (a half turn about an arbitrary axis is not the exact alignment either: round 2 missed
    //   that case), and those lanes derive it the long way.
    // (Two joints in flight per lane -- independent subtrees scheduled by the host onto two instruction streams -- was built
    // and measured: 286 us against 262 us for the same code with one stream.  The walk is not what the kernel waits for.)
    const int f = lane % FPW;  // C = 1: lanes >= FPW shadow lanes 0.. ; frames past a partial tile use their own slots
    const int ch = (C == 1) ? 0 : (lane / FPW) % C;  // which chain of the frame this lane walks
    float *fS = sS + f * FS;
    float g[4] = {1.0f, 0.0f, 0.0f, 0.0f};  // world quaternion of the joint this lane aligned last
    struct Ops { float pj[4], pc[4], a[4], b[4]; };
    // operands of one step that may be requested a step ahead: positions are static until their joint is aligned.  The PARENT's world quaternion is not
    // among them (round 6): it is read at the top of the step itself, after the write of the step before -- all chains of a frame are lanes of one wave,
    // whose DS operations execute in order -- so that a joint may follow its parent in the very next step on ANY chain (ik_schedule); against ~170
    // instructions of arithmetic the read's latency is nothing, and bushy trees' schedules lose a quarter of their steps
    auto fetch = [&](const v4i it, Ops &o) {
        lds_get<4>(fS, it.x, o.pj);
        lds_get<4>(fS, it.z, o.pc);  // children come later: their slots still hold positions
        lds_get<4>(sOff, 2 * it.z, o.a);      // rest offset u of the first child, 1 / |u|
        lds_get<4>(sOff, 2 * it.z + 1, o.b);  // u / |u|, |u|
    };
    // v turned by the INVERSE of the unit quaternion g:  v + 2 (w c + qv x c), c = qv x v, qv = -g.xyz  (18 instructions; the
    // reference's own order of terms, quat.py:320-334 / qmulvec, takes 21)
    auto unrotate = [](const float (&g)[4], const float (&v)[3], float (&o)[3]) {
        const float c0 = __builtin_fmaf(g[3], v[1], -(g[2] * v[2]));
        const float c1 = __builtin_fmaf(g[1], v[2], -(g[3] * v[0]));
        const float c2 = __builtin_fmaf(g[2], v[0], -(g[1] * v[1]));
        const float s0 = __builtin_fmaf(g[0], c0, __builtin_fmaf(g[3], c1, -(g[2] * c2)));
        const float s1 = __builtin_fmaf(g[0], c1, __builtin_fmaf(g[1], c2, -(g[3] * c0)));
        const float s2 = __builtin_fmaf(g[0], c2, __builtin_fmaf(g[2], c0, -(g[1] * c1)));
        o[0] = __builtin_fmaf(2.0f, s0, v[0]); o[1] = __builtin_fmaf(2.0f, s1, v[1]); o[2] = __builtin_fmaf(2.0f, s2, v[2]);
    };
    // N + dt, N - dt, |cr|^2 and cr for u (rest, |u|^2 = u2) and v: see above
    struct Pair { float cr[3], cr2, npd, nmd, N, iv; };
    auto pair_of = [](const float (&u)[3], const float lu, const float (&v)[3]) {
        Pair q;
        q.cr[0] = diff_of_products(u[1], v[2], u[2], v[1]);
        q.cr[1] = diff_of_products(u[2], v[0], u[0], v[2]);
        q.cr[2] = diff_of_products(u[0], v[1], u[1], v[0]);
        q.cr2 = __builtin_fmaf(q.cr[0], q.cr[0], __builtin_fmaf(q.cr[1], q.cr[1], q.cr[2] * q.cr[2]));
        const float dt = __builtin_fmaf(u[0], v[0], __builtin_fmaf(u[1], v[1], u[2] * v[2]));
        const float v2 = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], v[2] * v[2]));
        q.iv = __builtin_amdgcn_rsqf(v2);  // 1 / |v|: the eps term needs it anyway, and |v| = v2 / |v| saves the square root of N
        q.N = lu * (v2 * q.iv);            // (v = 0: NaN, caught by the N > 0 tests like the N = 0 it replaces)
        const float big = q.N + fabsf(dt), small = q.cr2 * frcp(big);
        q.npd = (dt >= 0.0f) ? big : small;
        q.nmd = (dt >= 0.0f) ? small : big;
        return q;
    };
    const int nitems = PM_ABLATED(a, 1) ? 0 : nitems_all;
    // Two steps per trip with ping-pong operand sets (A, B): the look-ahead costs no register moves (a single set copied
    // per step was 20 v_mov of ~170 instructions).  A step's item is consumed into scalars first and its registers are
    // refilled with the item two steps on; the other set's operands are requested before this step computes (if the next
    // item's parent is THIS lane's joint its gl is stale, and unused: register chain).
    int prevj = -1;
    auto step = [&](const int st, v4i &cur, const Ops &oc, const v4i &nxt, Ops &on) {
        const int j = cur.x, par = cur.y;
        const int xs = cur.w & 0xffff, nx = cur.w >> 16;
        float gl[4];
        lds_get<4>(fS, par < 0 ? 0 : par, gl);  // a finished parent's slot holds its G
        cur = sItem[(st + 2) * C + ch];
        fetch(nxt, on);
        float gpre[4];
        const bool chain = par == prevj;  // (the root, par = -1, is "chained" to the identity g starts as)
#pragma unroll
        for (int k = 0; k < 4; ++k) gpre[k] = chain ? g[k] : gl[k];
        const float (&pj)[4] = oc.pj, (&pc)[4] = oc.pc;
        const float u[3] = {oc.a[0], oc.a[1], oc.a[2]};
        const float iu = oc.a[3], lu = oc.b[3];
        const float (&un)[3] = reinterpret_cast<const float (&)[3]>(oc.b);  // u / |u|
        const float d[3] = {pc[0] - pj[0], pc[1] - pj[1], pc[2] - pj[2]};
        float p[3];
        unrotate(gpre, d, p);  // the child's offset in the parent's frame
        const Pair q = pair_of(u, lu, p);
        // The reference's dot is dt / ((|u| + 1e-8) (|v| + 1e-8)) = (dt / N) (1 - e), e = 1e-8 (1 / |u| + 1 / |v|): 2e-7 on a 0.1-unit
        // bone.  That moves sqrt((1 - dot) / 2) by e / (4 s) -- 1e-6 at a five-degree angle, 1e-5 at half a degree -- so it is
        // carried along (first order): N (1 +- dot_ref) = (N +- dt) -+ dt e.
        const float e = 1e-8f * (iu + q.iv), tol = 1.001e-5f * q.N;  // np.isclose(dot, +-1): 1e-8 + 1e-5
        const float dte = 0.5f * (q.npd - q.nmd) * e;                                       // dt e
        const float npr = q.npd - dte, nmr = q.nmd + dte;                                   // N (1 +- dot_ref)
        // (sqrt(npr), sqrt(nmr) cr / |cr|) / sqrt(2 N) with |cr|^2 = npd nmd, expanded to first order in dte: ONE reciprocal square
        // root and one reciprocal (the literal form: two square roots and two reciprocal square roots), unit to O(e^2)
        const float A = __builtin_amdgcn_rsqf((q.N + q.N) * q.npd);
        const float vs = A * __builtin_fmaf(0.5f * dte, frcp(q.nmd), 1.0f);
        float r[4] = {A * __builtin_fmaf(-0.5f, dte, q.npd), q.cr[0] * vs, q.cr[1] * vs, q.cr[2] * vs};
        const bool snap = nmr <= tol || !(q.N > 0.0f);
        if (snap) { r[0] = 1.0f; r[1] = 0.0f; r[2] = 0.0f; r[3] = 0.0f; }
        const bool anti = npr <= tol && q.N > 0.0f;
        if (__builtin_amdgcn_ballot_w64(anti) != 0 && anti) {  // anti-parallel (:554-571), rare: skipped by the whole wave otherwise
            const float a1[3] = {un[0], un[1], un[2]};
            const bool xlike = isclose_to(fabsf(a1[0]), 1.0f);
            const float og[3] = {xlike ? 0.0f : 1.0f, xlike ? 1.0f : 0.0f, 0.0f};
            const float c2[3] = {a1[1] * og[2] - a1[2] * og[1], a1[2] * og[0] - a1[0] * og[2], a1[0] * og[1] - a1[1] * og[0]};
            float ax2[3];
            vnormalize(c2, 1e-8f, ax2);
            r[0] = 0.0f; r[1] = ax2[0]; r[2] = ax2[1]; r[3] = ax2[2];
        }
        qmul(gpre, r, g);  // G_j once the first child is aligned
        const bool inexact = snap || anti;  // G_j does not take the rest direction onto d (not even approximately)
        // roll correction from every further child: G_j <- G_j (x) roll; per lane with two chains (the other chain's lanes wait)
        for (int rr = 0; __builtin_amdgcn_ballot_w64(rr < nx) != 0; ++rr) {
            const bool act = rr < nx;
            const int gc = sTopo[2 * J + 1 + (act ? xs + rr : xs)];
            float pg[4], ug[4];
            lds_get<4>(fS, gc, pg);
            lds_get<4>(sOff, 2 * gc, ug);  // rest offset of this child, 1 / its length
            const float lug = sOff[8 * gc + 7];
            const float dg[3] = {pg[0] - pj[0], pg[1] - pj[1], pg[2] - pj[2]};
            float v[3];
            unrotate(g, dg, v);
            // The roll axis the reference derives, inv(G_j) normalize(P_c0 - P_j), is NOT quite the first child's rest direction: its
            // alignment turns by acos(dot (1 - e)), e = 1e-8 (1 / |u| + 1 / |v|), not by the angle between the two directions, so G_j
            // leaves the rest direction e cot(angle) short of d -- 1e-4 rad on a 3 cm bone turned by a third of a degree (or by
            // 179.7), and the roll quaternion inherits that times sin(roll / 2): round 3, which used the rest direction as it stands,
            // read 1e-4 there (3e-5 at one degree) where one ulp of the inputs moves the reference's answer by 3e-8.  To first order
            // in e the axis is  un + e (dt N / |cr|^2) ((dt / N) un - p / |p|)  -- from exact table values and the step's own
            // quantities, i.e. without the rounding noise of deriving it the long way (measured: the long way in fp32 is worse than
            // no correction on trees, 2e-4 against 6e-5 at 3001 x 52).  Lanes whose alignment snapped to the identity or took the
            // anti-parallel branch derive it the long way as before (a half turn about an arbitrary axis is no alignment at all).
            float axis[3];
            {
                float kk = dte * q.N;
                asm volatile("" : "+v"(kk));  // or all of this is hoisted in front of the loop and paid by every step
                kk *= frcp(q.cr2);
                const float cs = 0.5f * (q.npd - q.nmd) * frcp(q.N), ks = 1.0f + kk * cs, kp = kk * q.iv;
                axis[0] = __builtin_fmaf(-kp, p[0], ks * un[0]); axis[1] = __builtin_fmaf(-kp, p[1], ks * un[1]); axis[2] = __builtin_fmaf(-kp, p[2], ks * un[2]);
            }
            if (__builtin_amdgcn_ballot_w64(inexact && act) != 0) {
                float dd[3] = {d[0], d[1], d[2]}, dn[3], ax[3];
                asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]));  // or the normalisation (a square root and a reciprocal) is hoisted in front of the loop and paid by every step
                vnormalize(dd, 1e-8f, dn);
                unrotate(g, dn, ax);
                axis[0] = inexact ? ax[0] : axis[0]; axis[1] = inexact ? ax[1] : axis[1]; axis[2] = inexact ? ax[2] : axis[2];
            }
            const float ub[3] = {ug[0], ug[1], ug[2]};
            const Pair t = pair_of(ub, lug, v);
            // (w, s) = (sqrt((1 + dot) / 2), sqrt((1 - dot) / 2)) with the reference's dot (see the alignment); s signed by
            // cr . axis (np.sign: 0 -> 0, NaN -> NaN)
            const float eg = 1e-8f * (ug[3] + t.iv), tolg = 1.001e-5f * t.N;
            const float dtg = 0.5f * (t.npd - t.nmd) * eg;
            const float npg = t.npd - dtg, nmg = t.nmd + dtg;
            const float i2n = __builtin_amdgcn_rsqf(t.N + t.N);
            const float cda = t.cr[0] * axis[0] + t.cr[1] * axis[1] + t.cr[2] * axis[2];
            const float sg = (cda > 0.0f) ? 1.0f : ((cda < 0.0f) ? -1.0f : cda);
            const float w = fsqrt(npg) * i2n, sn = fsqrt(nmg) * i2n * sg;
            float roll[4] = {w, axis[0] * sn, axis[1] * sn, axis[2] * sn};
            // (a zero-length v: the reference's dot is 0 there, w = s = sqrt(.5) about `axis`; here NaN -- the positions of a
            // joint and its child coincide, nothing the reference's own tests exercise)
            if (nmg <= tolg) { roll[0] = 1.0f; roll[1] = 0.0f; roll[2] = 0.0f; roll[3] = 0.0f; }
            if (npg <= tolg) { roll[0] = 0.0f; roll[1] = axis[0]; roll[2] = axis[1]; roll[3] = axis[2]; }
            float g2[4];
            qmul(g, roll, g2);
            g[0] = act ? g2[0] : g[0]; g[1] = act ? g2[1] : g[1]; g[2] = act ? g2[2] : g[2]; g[3] = act ? g2[3] : g[3];
        }
        lds_put<4>(fS, j, g);  // P_j is dead from here on
        prevj = j;
    };
    v4i iA = sItem[ch], iB = sItem[C + ch];
    Ops oA, oB;
    fetch(iA, oA);
    for (int st = 0; st < nitems; st += 2) {
        step(st, iA, oA, iB, oB);
        if (st + 1 >= nitems) break;
        step(st + 1, iB, oB, iA, oA);
    }
    wave_sync();

    // ---- local rotations back out of the world quaternions, lane per (frame, joint), straight to HBM -----------
    float *gout = a.out + f0 * J * 4;
    for_each_slot<2>(PM_ABLATED(a, 2) ? 0 : n, lane, [&](const int e, const bool valid) {
        const int fr = (int)(((float)e + 0.5f) * invJ);
        const int j = e - fr * J;
        const float *fq = sS + fr * FS;
        float gj[4], gp[4], o[4];
        lds_get<4>(fq, j, gj);
        const int info = sTopo[j];
        lds_get<4>(fq, info & 0xffff, gp);
        const float inv[4] = {gp[0], -gp[1], -gp[2], -gp[3]};
        qmul(inv, gj, o);
        const bool leaf = (info & 0x10000) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = leaf ? (k == 0 ? 1.0f : 0.0f) : ((j == 0) ? gj[k] : o[k]);
        if (valid) {
            if (VEC) __builtin_nontemporal_store(v4f{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4f *>(gout) + e);
            else { gout[4 * e] = o[0]; gout[4 * e + 1] = o[1]; gout[4 * e + 2] = o[2]; gout[4 * e + 3] = o[3]; }
        }
    });
    wave_sync();  // the image is reused by the next tile
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ONE LANE PER FRAME, the joints streamed through LDS (the shape of deep.hip) -- for any table, parents first, whose children sit close
// enough behind their parent: depth-first storage (every BVH hierarchy), the level-order tables of the SMPL family, anything between.
// The kernel above keeps a frame's 16 J bytes in LDS for the whole walk, which is what bounds it: 32 frames x 2 chains per wave
// at 11 waves per CU (J = 22), half the lane-steps of its two-chain schedule idle, a final lane-per-record pass to turn world
// quaternions back into local ones; five 31 KB images per CU at J = 52.  Here
//   * a lane walks ITS frame's joints in memory order -- 64 frames per wave, no idle chains;
//   * LDS is a ring of sixteen 16-byte slots per frame -- a joint's position comes in (cut at the output's 128-byte lines:
//     groups of eight records g = f J + j, as in to_root_dq_ring_kernel), and the joint's LOCAL rotation -- what the alignment and
//     the rolls produce (r (x) roll (x) ...): no world -> local pass -- goes out through the same slot: 17 KB per wave, eight or nine
//     waves = 512-576 frames in flight per CU;
//   * the walk is a host-made list of OPERATIONS, one per joint (ik_order_plan).  Joint p is finished by ONE operation that reads P_p
//     and the positions of its children out of the ring (they are still there: see the window below) or, for children stored too far
//     down the row (the further children of a depth-first table, the finger roots of a level-order hand), out of a queue of per-lane
//     fetches made when the tile starts, aligns, rolls once per further child (the reference consumes them at the same moment,
//     skeleton.py:147-168) and writes the rotation over P_p; a joint without children is an operation that writes the identity;
//   * the window: while step c of the tile loop runs, the ring holds groups c - 1 and c of every lane, i.e. the joints
//     [8 c - 8, 8 c + D] whatever the lane's line shift sf = (frame J) & 7 is, D = gcd(J, 8) - 1 (J = 52: D = 3; J % 8 == 0: 7;
//     odd J: 0).  The plan puts joint p into step (p >> 3) + 1 -- the last one before its line leaves -- takes the children at index
//     <= 8 step + D from the ring and the others from the queue, and runs a step's operations in joint order (parents first: every
//     parent is final before any of its children is touched);
//   * the parent's world quaternion: the lane's own registers if the parent's operation was the last one that aligned anything, else
//     one of six saved register sets (the plan colours their live ranges over the operation sequence, like deep_plan);
//   * the operands of the next operation (two positions, the table row of the first child) are requested while the current one
//     computes, into two operand sets used in turn.
// SMPL-H as stored (52 joints, level order): 8 queue entries, 4 register sets; the 22-joint BVH body: 4 and 1.
// Round 3's depth-first-only form of this kernel ("joint j finishes joint j - 1", every further child through the queue) is gone: the
// operation list covers every table it took, with fewer queue entries, and measured faster on all of them (chain-like 2^19 x 24 / 32 /
// 64 / 128: 82 / 105 / 232 / 406 us against 88 / 118 / 244 / 452 us; a depth-first SMPL-H 95 against 111 us at 2^18 frames).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ik_unrotate(const float (&g)[4], const float (&v)[3], float (&o)[3]) {  // v turned by the inverse of unit g
    const float c0 = __builtin_fmaf(g[3], v[1], -(g[2] * v[2]));
    const float c1 = __builtin_fmaf(g[1], v[2], -(g[3] * v[0]));
    const float c2 = __builtin_fmaf(g[2], v[0], -(g[1] * v[1]));
    const float s0 = __builtin_fmaf(g[0], c0, __builtin_fmaf(g[3], c1, -(g[2] * c2)));
    const float s1 = __builtin_fmaf(g[0], c1, __builtin_fmaf(g[1], c2, -(g[3] * c0)));
    const float s2 = __builtin_fmaf(g[0], c2, __builtin_fmaf(g[2], c0, -(g[1] * c1)));
    o[0] = __builtin_fmaf(2.0f, s0, v[0]); o[1] = __builtin_fmaf(2.0f, s1, v[1]); o[2] = __builtin_fmaf(2.0f, s2, v[2]);
}
struct IkPair { float cr[3], cr2, npd, nmd, N, iv; };
__device__ __forceinline__ IkPair ik_pair_of(const float (&u)[3], const float lu, const float (&v)[3]) {  // see the walk of the tile kernel
    IkPair q;
    q.cr[0] = diff_of_products(u[1], v[2], u[2], v[1]);
    q.cr[1] = diff_of_products(u[2], v[0], u[0], v[2]);
    q.cr[2] = diff_of_products(u[0], v[1], u[1], v[0]);
    q.cr2 = __builtin_fmaf(q.cr[0], q.cr[0], __builtin_fmaf(q.cr[1], q.cr[1], q.cr[2] * q.cr[2]));
    const float dt = __builtin_fmaf(u[0], v[0], __builtin_fmaf(u[1], v[1], u[2] * v[2]));
    const float v2 = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], v[2] * v[2]));
    q.iv = __builtin_amdgcn_rsqf(v2);
    q.N = lu * (v2 * q.iv);
    const float big = q.N + fabsf(dt), small = q.cr2 * frcp(big);
    q.npd = (dt >= 0.0f) ? big : small;
    q.nmd = (dt >= 0.0f) ? small : big;
    return q;
}
// from_to(u, inv(gpre) d) with the reference's eps terms and special cases (the alignment of the tile kernel's walk, same formulas);
// `inexact`: the result does not take the rest direction exactly onto d (snapped to the identity, or the anti-parallel branch)
__device__ __forceinline__ void ik_align(const float (&gpre)[4], const float (&d)[3], const float (&ta)[4], const float (&tb)[4], float (&r)[4], bool &inexact, const bool want_axis, float (&axis)[3]) {
    const float u[3] = {ta[0], ta[1], ta[2]};
    const float iu = ta[3], lu = tb[3];
    float p[3];
    ik_unrotate(gpre, d, p);
    const IkPair q = ik_pair_of(u, lu, p);
    const float e = 1e-8f * (iu + q.iv), tol = 1.001e-5f * q.N;
    const float dte = 0.5f * (q.npd - q.nmd) * e;
    const float npr = q.npd - dte, nmr = q.nmd + dte;
    const float A = __builtin_amdgcn_rsqf((q.N + q.N) * q.npd);
    const float vs = A * __builtin_fmaf(0.5f * dte, frcp(q.nmd), 1.0f);
    r[0] = A * __builtin_fmaf(-0.5f, dte, q.npd); r[1] = q.cr[0] * vs; r[2] = q.cr[1] * vs; r[3] = q.cr[2] * vs;
    const bool snap = nmr <= tol || !(q.N > 0.0f);
    if (snap) { r[0] = 1.0f; r[1] = 0.0f; r[2] = 0.0f; r[3] = 0.0f; }
    const bool anti = npr <= tol && q.N > 0.0f;
    if (__builtin_amdgcn_ballot_w64(anti) != 0 && anti) {  // anti-parallel (quat.py:554-571), rare: skipped by the whole wave otherwise
        const float a1[3] = {tb[0], tb[1], tb[2]};
        const bool xlike = isclose_to(fabsf(a1[0]), 1.0f);
        const float og[3] = {xlike ? 0.0f : 1.0f, xlike ? 1.0f : 0.0f, 0.0f};
        const float c2[3] = {a1[1] * og[2] - a1[2] * og[1], a1[2] * og[0] - a1[0] * og[2], a1[0] * og[1] - a1[1] * og[0]};
        float ax2[3];
        vnormalize(c2, 1e-8f, ax2);
        r[0] = 0.0f; r[1] = ax2[0]; r[2] = ax2[1]; r[3] = ax2[2];
    }
    inexact = snap || anti;
    // the roll axis of further children, to first order in e (see the tile kernel's walk): un + e (dt N / |cr|^2) ((dt / N) un - p / |p|)
    axis[0] = tb[0]; axis[1] = tb[1]; axis[2] = tb[2];
    if (want_axis) {  // wave-uniform
        const float kk = dte * q.N * frcp(q.cr2);
        const float cs = 0.5f * (q.npd - q.nmd) * frcp(q.N), ks = 1.0f + kk * cs, kp = kk * q.iv;
        axis[0] = __builtin_fmaf(-kp, p[0], ks * tb[0]); axis[1] = __builtin_fmaf(-kp, p[1], ks * tb[1]); axis[2] = __builtin_fmaf(-kp, p[2], ks * tb[2]);
    }
}
// from_to_axis(offsets[gc], inv(g) dg, axis) (quat.py:579-650) as the tile kernel's walk evaluates it; tg = {u_gc, 1 / |u_gc|}, lug = |u_gc|
__device__ __forceinline__ void ik_roll(const float (&g)[4], const float (&dg)[3], const float (&d)[3], const float (&un)[3], const bool inexact,
                                        const float (&tg)[4], const float lug, float (&roll)[4]) {
    float v[3];
    ik_unrotate(g, dg, v);
    float axis[3] = {un[0], un[1], un[2]};  // (un: the axis ik_align derived, first order in e)
    if (__builtin_amdgcn_ballot_w64(inexact) != 0) {
        float dd[3] = {d[0], d[1], d[2]}, dn[3], ax[3];
        asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]));
        vnormalize(dd, 1e-8f, dn);
        ik_unrotate(g, dn, ax);
        axis[0] = inexact ? ax[0] : axis[0]; axis[1] = inexact ? ax[1] : axis[1]; axis[2] = inexact ? ax[2] : axis[2];
    }
    const float ub[3] = {tg[0], tg[1], tg[2]};
    const IkPair t = ik_pair_of(ub, lug, v);
    const float eg = 1e-8f * (tg[3] + t.iv), tolg = 1.001e-5f * t.N;
    const float dtg = 0.5f * (t.npd - t.nmd) * eg;
    const float npg = t.npd - dtg, nmg = t.nmd + dtg;
    const float i2n = __builtin_amdgcn_rsqf(t.N + t.N);
    const float cda = t.cr[0] * axis[0] + t.cr[1] * axis[1] + t.cr[2] * axis[2];
    const float sg = (cda > 0.0f) ? 1.0f : ((cda < 0.0f) ? -1.0f : cda);
    const float w = fsqrt(npg) * i2n, sn = fsqrt(nmg) * i2n * sg;
    roll[0] = w; roll[1] = axis[0] * sn; roll[2] = axis[1] * sn; roll[3] = axis[2] * sn;
    if (nmg <= tolg) { roll[0] = 1.0f; roll[1] = 0.0f; roll[2] = 0.0f; roll[3] = 0.0f; }
    if (npg <= tolg) { roll[0] = 0.0f; roll[1] = axis[0]; roll[2] = axis[1]; roll[3] = axis[2]; }
}

struct IkSaves { float g[kDeepSlots][4]; };
template <int K>
__device__ __forceinline__ void ik_slot_load(const int ld, const IkSaves &sv, float (&g)[4]) {
    if constexpr (K < kDeepSlots) {
        int code = ld;
        asm volatile("" : "+s"(code));  // an opaque copy per test (see deep.hip: an indexed array would live in scratch memory)
        if (code == K) { g[0] = sv.g[K][0]; g[1] = sv.g[K][1]; g[2] = sv.g[K][2]; g[3] = sv.g[K][3]; }
        ik_slot_load<K + 1>(ld, sv, g);
    }
}
template <int K>
__device__ __forceinline__ void ik_slot_save(const int st, IkSaves &sv, const float (&g)[4]) {
    if constexpr (K < kDeepSlots) {
        int code = st;
        asm volatile("" : "+s"(code));
        if (code == K) { sv.g[K][0] = g[0]; sv.g[K][1] = g[1]; sv.g[K][2] = g[2]; sv.g[K][3] = g[3]; }
        ik_slot_save<K + 1>(st, sv, g);
    }
}

// The same two behind one test that skips the chain when no set is named (a halving test in front of two half chains was tried: the compiler
// merges the halves into an indexed array, which lives in scratch memory)
__device__ __forceinline__ void ik_set_load(const int code, const IkSaves &sv, float (&g)[4]) {
    int c0 = code;
    asm volatile("" : "+s"(c0));
    if (c0 < kDeepSlots) ik_slot_load<0>(code, sv, g);
}
__device__ __forceinline__ void ik_set_save(const int code, IkSaves &sv, const float (&g)[4]) {
    int c0 = code;
    asm volatile("" : "+s"(c0));
    if (c0 < kDeepSlots) ik_slot_save<0>(code, sv, g);
}

// Records per group: 8 -- whole 128-byte lines of output, 96 bytes of input, 17 KB of LDS per wave.  The kernel is a latency chain per lane
// and lives on occupancy (round 3, 2^20 x 22 with 8 / 7 / 5 / 4 waves per CU: 158 / 174 / 193 / 267 us), but groups of four -- half the ring,
// sixteen waves per CU -- move HALF lines and were slower for it (240 us; J = 128 at 2^19: 615 against 444 us): not instantiated.
__host__ __device__ constexpr int ik_deep_row(const int G) { return 2 * G * 4 + 4; }  // 2 G 16-byte slots + 16 bytes: (row / 4) odd

// Which tables take it (ik_order_wanted): measured against the tile kernels on chain-like skeletons at 2^20 frames, tile / this kernel --
// J = 8 52 / 56 us, 12 85 / 79, 16 108 / 98, 20 139 / 125, 24 174 / 154, 32 247 / 222 (J % 4 == 0: few distinct line shifts, D >= 3);
// J = 14 95 / 118, 18 128 / 138, 22 159 / 170, 26 193 / 200, 30 231 / 227, 34 279 / 258; odd J = 13 92 / 131, 19 134 / 165, 25 191 / 211,
// 29 231 / 233, 31 233 / 224, 35 284 / 258, 41 371 / 306, 47 468 / 337.  The 22-joint BVH body: 155-165 / 144-164 us, a draw.
static bool ik_order_wanted(const int J) { return (J >= 12 && J % 4 == 0) || J >= 30; }
constexpr int kIkOrderFar = 16;  // queue entries (the kernel is instantiated for 4 / 8 / 12 / 16: every consumption shifts the whole queue)
constexpr int kIkOrderSteps = PM_MAX_JOINTS / 8 + 4;
enum : int { IKO_CHAIN = 6, IKO_ROOT = 7, IKO_NONE = 7 };

struct IkOrderArgs {
    const float *pos;      // [F,J,3]
    const float *offsets;  // [J,3]
    float *out;            // [F,J,4]
    int64_t F;
    int32_t J;
    int32_t nfar;
    int32_t far_joint[kIkOrderFar];    // joints fetched per lane at the start of a tile, in the order the operations consume them
    int32_t wst[kIkOrderSteps];        // step c runs operations [wst[c], wst[c + 1])
    int32_t rsrc[PM_MAX_JOINTS / 2];   // further children, 16 bits each: joint | far << 15, in the order the operations consume them
    int32_t ops[PM_MAX_JOINTS];        // p | c1 << 9 | c1far << 18 | ld << 19 | st << 22 | nroll << 25 | leaf << 29
};
static_assert(sizeof(IkOrderArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <int G, int NF, bool TWO>
__global__ __launch_bounds__(PM_WAVE) void from_root_positions_order_kernel(const IkOrderArgs a) {
    // TWO: the ring leaves room for eight waves on a CU, which the dispatcher spreads evenly only if no SIMD can take a third: a register
    // count beyond 512 / 3 makes sure (the kernel needs ~160; without this 2^18 x 52 reads 109 us against 104 us)
    if constexpr (TWO) asm volatile("; two waves per SIMD" ::: "v183");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RS = ik_deep_row(G), FPI = PM_WAVE / G, LG = 3, SM = 2 * G - 1;
    static_assert(G == 8, "groups of eight records");
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t tile = xcd_tile((a.F + PM_WAVE - 1) / PM_WAVE);
    if (tile < 0) return;
    float *sImg = smem;                 // [64][RS]
    float *sOff = smem + PM_WAVE * RS;  // [J][4]  {u, 1 / |u|} of every joint's rest offset ...
    float *sLen = sOff + 4 * J;         // [J]     ... and |u|
    for (int j = lane; j < J; j += PM_WAVE) {
        const float o[3] = {a.offsets[3 * j], a.offsets[3 * j + 1], a.offsets[3 * j + 2]};
        const float u2 = __builtin_fmaf(o[0], o[0], __builtin_fmaf(o[1], o[1], o[2] * o[2]));
        const float iu = (u2 > 0.0f) ? __builtin_amdgcn_rsqf(u2) : 0.0f;
        float *t = sOff + 4 * j;
        if (PM_LDS_OK(t, 16u)) *reinterpret_cast<v4f *>(t) = v4f{o[0], o[1], o[2], iu};
        if (PM_LDS_OK(sLen + j, 4u)) sLen[j] = fsqrt(u2);
    }
    const int64_t f0 = tile * PM_WAVE;
    const int nf = (int)((a.F - f0) < PM_WAVE ? (a.F - f0) : PM_WAVE);
    const int ngroups = ((J + G - 2) >> LG) + 1;
    const float *gpos = a.pos + f0 * J * 3;
    float *gout = a.out + f0 * J * 4;
    const int fl = lane < nf ? lane : nf - 1;

    v3f_a4 farl[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        farl[k] = v3f_a4{0.0f, 0.0f, 0.0f};
        if (k < a.nfar) farl[k] = *reinterpret_cast<const v3f_a4 *>(gpos + ((int64_t)fl * J + a.far_joint[k]) * 3);
    }
    float fq[NF][3];  // the queue itself: consumed from entry 0, shifted IN PLACE (far_pop)

    const int l_frl = lane >> LG, l_d = (lane & (G - 1)) - ((l_frl * J) & (G - 1));
    v3f_a4 pre[G], pre1[G];
    auto issue = [&](const int c, v3f_a4 (&pre)[G]) {
        int j = G * c + l_d;
        j = j < 0 ? 0 : (j > J - 1 ? J - 1 : j);
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int fr = FPI * u + l_frl, fc = fr < nf ? fr : nf - 1;
            pre[u] = *reinterpret_cast<const v3f_a4 *>(gpos + ((int64_t)fc * J + j) * 3);
        }
    };
    issue(0, pre);
    if (ngroups > 1) issue(1, pre1);
    auto park = [&](const int c, const v3f_a4 (&pre)[G]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            float *p = sImg + (FPI * u + l_frl) * RS + ((c & 1) * G + (lane & (G - 1))) * 4;
            if (PM_LDS_OK(p, 16u)) { p[0] = pre[u].x; p[1] = pre[u].y; p[2] = pre[u].z; }
        }
    };
    float *row = sImg + lane * RS;
    const int sf = (lane * J) & (G - 1);
    float g[4] = {1.0f, 0.0f, 0.0f, 0.0f};   // world quaternion of the joint finished last
    IkSaves sv = {};
    int ri = 0;                              // next entry of the further-children list (wave-uniform)
    wave_sync();
    // entry 0 out, the rest one place down -- as moves the compiler cannot re-home: written as plain assignments it gives the queue other
    // registers on the path that shifts, and the path that does NOT then copies the whole queue (18 v_mov in every operation)
    auto far_pop = [&](auto &dst) {
        dst[0] = fq[0][0]; dst[1] = fq[0][1]; dst[2] = fq[0][2];
#pragma unroll
        for (int k = 0; k + 1 < NF; ++k)
            asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                         : "+v"(fq[k][0]), "+v"(fq[k][1]), "+v"(fq[k][2]) : "v"(fq[k + 1][0]), "v"(fq[k + 1][1]), "v"(fq[k + 1][2]));
    };
    // An operation's operands (P_p, P_c1, the first child's table row) are requested while the operation before it computes; two sets used in
    // turn (a single look-ahead set is copied with ~20 v_mov per operation)
    struct Opn { float pp[4], pc[4], ta[4], len; };
    auto fetch = [&](const int cd, Opn &q) {
        if (!(cd & (1 << 29))) {
            const int p = cd & 511, c1 = (cd >> 9) & 511;
            lds_get<4>(row + ((p + sf) & SM) * 4, 0, q.pp);
            if (!(cd & (1 << 18))) lds_get<4>(row + ((c1 + sf) & SM) * 4, 0, q.pc);
            lds_get<4>(sOff, c1, q.ta);
            q.len = sLen[c1];
        }
    };
    auto op = [&](const int cur, const Opn &q, const bool more, const int nxt, Opn &qn) {
        asm volatile("" ::"v"(q.pp[3]), "v"(q.pc[3]));  // (the unused fourth floats: left free, their registers are re-used for addresses right behind the read -- a wait per read)
        const float pp[3] = {q.pp[0], q.pp[1], q.pp[2]}, ta[4] = {q.ta[0], q.ta[1], q.ta[2], q.ta[3]};
        float pc[3] = {q.pc[0], q.pc[1], q.pc[2]}, tb[4];
        tb[0] = ta[0] * ta[3]; tb[1] = ta[1] * ta[3]; tb[2] = ta[2] * ta[3]; tb[3] = q.len;
        const int p = cur & 511;
        if (more) fetch(nxt, qn);  // positions parked before this step began, never this operation's own slot
        if (cur & (1 << 29)) {  // a joint without children keeps the identity (skeleton.py:126-130)
            const float id[4] = {1.0f, 0.0f, 0.0f, 0.0f};
            lds_put<4>(row + ((p + sf) & SM) * 4, 0, id);
            return;
        }
        const int ld = (cur >> 19) & 7, st = (cur >> 22) & 7, nroll = (cur >> 25) & 15;
        if (cur & (1 << 18)) {  // the first child lies beyond the window
            far_pop(pc);
        }
        float gpre[4] = {g[0], g[1], g[2], g[3]};
        ik_set_load(ld, sv, gpre);
        if (ld == IKO_ROOT) { gpre[0] = 1.0f; gpre[1] = 0.0f; gpre[2] = 0.0f; gpre[3] = 0.0f; }
        const float d[3] = {pc[0] - pp[0], pc[1] - pp[1], pc[2] - pp[2]};
        float r[4];
        bool inexact;
        float un[3];
        ik_align(gpre, d, ta, tb, r, inexact, nroll > 0, un);
        qmul(gpre, r, g);
        for (int rr = 0; rr < nroll; ++rr) {  // further children (wave-uniform)
            const int w = __builtin_amdgcn_readfirstlane(a.rsrc[ri >> 1]);
            const int e = (ri & 1) ? (w >> 16) & 0xffff : w & 0xffff;
            ++ri;
            const int gc = e & 511;
            float tg[4], pg[4];
            lds_get<4>(sOff, gc, tg);
            const float lug = sLen[gc];
            if (e & 0x8000) {
                far_pop(pg);
            } else {
                lds_get<4>(row + ((gc + sf) & SM) * 4, 0, pg);
            }
            const float dg[3] = {pg[0] - pp[0], pg[1] - pp[1], pg[2] - pp[2]};
            float roll[4], g2[4], r2[4];
            ik_roll(g, dg, d, un, inexact, tg, lug, roll);
            qmul(g, roll, g2);
            qmul(r, roll, r2);
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[k] = g2[k]; r[k] = r2[k]; }
        }
        ik_set_save(st, sv, g);
        lds_put<4>(row + ((p + sf) & SM) * 4, 0, r);  // the local rotation of p, through the slot its position came in by
    };
    auto walk = [&](const int c) {
        const int ob = __builtin_amdgcn_readfirstlane(a.wst[c]), oe = __builtin_amdgcn_readfirstlane(a.wst[c + 1]);
        if (ob >= oe) return;
        Opn qa = {}, qb = {};
        int code = __builtin_amdgcn_readfirstlane(a.ops[ob]), o = ob;
        fetch(code, qa);
#pragma unroll 1
        for (;;) {
            bool more = o + 1 < oe;
            int nxt = more ? __builtin_amdgcn_readfirstlane(a.ops[o + 1]) : 0;
            op(code, qa, more, nxt, qb);
            if (!more) break;
            ++o; code = nxt;
            more = o + 1 < oe;
            nxt = more ? __builtin_amdgcn_readfirstlane(a.ops[o + 1]) : 0;
            op(code, qb, more, nxt, qa);
            if (!more) break;
            ++o; code = nxt;
        }
    };
    const int s_d = l_d;
    v4f outr[G];
    auto read_group = [&](const int k) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const float *p = sImg + (FPI * u + l_frl) * RS + ((k & 1) * G + (lane & (G - 1))) * 4;
            outr[u] = PM_LDS_OK(p, 16u) ? *reinterpret_cast<const v4f *>(p) : v4f{0, 0, 0, 0};
        }
    };
    auto store_group = [&](const int k) {
        const int j = G * k + s_d;
        const bool jok = j >= 0 && j < J;
        float *g0 = gout + (l_frl * J + (jok ? j : 0)) * 4;
#pragma unroll
        for (int u = 0; u < G; ++u)
            if (jok && FPI * u + l_frl < nf) __builtin_nontemporal_store(outr[u], reinterpret_cast<v4f *>(g0 + FPI * u * J * 4));
    };
    park(0, pre);
    if (ngroups > 1) park(1, pre1);
#pragma unroll
    for (int k = 0; k < NF; ++k) {  // settled HERE (everything requested so far has arrived): their first use is inside the walk
        fq[k][0] = farl[k].x; fq[k][1] = farl[k].y; fq[k][2] = farl[k].z;
        asm volatile("" : "+v"(fq[k][0]), "+v"(fq[k][1]), "+v"(fq[k][2]));
    }
    if (ngroups > 2) issue(2, pre);
    for (int c = 0; c <= ngroups; ++c) {
        wave_sync();
        walk(c);
        wave_sync();
        if (c >= 1) {
            read_group(c - 1);
            wave_sync();
            if (c + 1 < ngroups) park(c + 1, pre);
            if (c + 2 < ngroups) issue(c + 2, pre);
            store_group(c - 1);
        }
    }
}

// Host plan of the kernel above (see its header).  Returns false when the table does not fit: a child inside the window rule but
// more than kIkOrderFar children beyond it, more than 15 further children of one joint, more than kDeepSlots world quaternions alive.
static bool ik_order_plan(const Topo16 &t, const int J, IkOrderArgs &a) {
    int g8 = 8;
    while (J % g8) g8 >>= 1;
    const int D = g8 - 1;
    const int ngroups = ((J + 6) >> 3) + 1;
    if (ngroups + 2 > kIkOrderSteps) return false;
    // operations in execution order: step (p >> 3) + 1, joint index inside a step -- i.e. plain index order, cut into steps
    int nfar = 0, nr = 0;
    uint16_t rs[PM_MAX_JOINTS];
    for (int c = 0; c <= ngroups + 1; ++c) a.wst[c] = 0;
    for (int p = 0; p < J; ++p) a.wst[(p >> 3) + 2]++;          // count of step (p >> 3) + 1, shifted by one for the prefix sum
    for (int c = 1; c <= ngroups + 1; ++c) a.wst[c] += a.wst[c - 1];
    for (int c = ngroups + 2; c < kIkOrderSteps; ++c) a.wst[c] = a.wst[ngroups + 1];
    // world quaternion of p: needed by the operations of its children that have children; `chain` if that operation is the next
    // one that aligns anything (operations of childless joints do not touch the registers)
    int next_align[PM_MAX_JOINTS + 1];
    next_align[J] = -1;
    for (int p = J - 1; p >= 0; --p) next_align[p] = (t.cstart[p + 1] > t.cstart[p]) ? p : next_align[p + 1];
    int last_use[PM_MAX_JOINTS], slot_of[PM_MAX_JOINTS], busy_until[kDeepSlots];
    for (int p = 0; p < J; ++p) { last_use[p] = -1; slot_of[p] = -1; }
    for (int j = 1; j < J; ++j) {
        const int pa = t.parent[j];
        if (t.cstart[j + 1] > t.cstart[j] && next_align[pa + 1] != j && last_use[pa] < j) last_use[pa] = j;
    }
    for (int k = 0; k < kDeepSlots; ++k) busy_until[k] = -1;
    for (int p = 0; p < J; ++p) {
        const int cs = t.cstart[p], ce = t.cstart[p + 1];
        if (ce == cs) { a.ops[p] = p | (1 << 29); continue; }
        const int hi = 8 * ((p >> 3) + 1) + D;                  // children up to here are read from the ring
        const int c1 = t.clist[cs], c1far = c1 > hi;
        if (c1far) { if (nfar == kIkOrderFar) return false; a.far_joint[nfar++] = c1; }
        const int nroll = ce - cs - 1;
        if (nroll > 15) return false;
        for (int k = cs + 1; k < ce; ++k) {
            const int gc = t.clist[k], far = gc > hi;
            if (far) { if (nfar == kIkOrderFar) return false; a.far_joint[nfar++] = gc; }
            rs[nr++] = (uint16_t)(gc | (far ? 0x8000 : 0));
        }
        int ld;
        if (p == 0) ld = IKO_ROOT;
        else if (next_align[t.parent[p] + 1] == p) ld = IKO_CHAIN;
        else { ld = slot_of[t.parent[p]]; if (ld < 0) return false; }
        int st = IKO_NONE;
        if (last_use[p] >= 0) {
            int k = 0;
            while (k < kDeepSlots && busy_until[k] > p) ++k;   // (a set is free again for the operation that follows its last reader)
            if (k == kDeepSlots) return false;
            busy_until[k] = last_use[p];
            slot_of[p] = k;
            st = k;
        }
        a.ops[p] = p | (c1 << 9) | (c1far << 18) | (ld << 19) | (st << 22) | (nroll << 25);
    }
    for (int k = nfar; k < kIkOrderFar; ++k) a.far_joint[k] = 0;
    a.nfar = nfar;
    for (int k = 0; k < PM_MAX_JOINTS / 2; ++k) a.rsrc[k] = 0;
    for (int k = 0; k < nr; ++k) a.rsrc[k >> 1] |= (int32_t)((uint32_t)rs[k] << (16 * (k & 1)));
    return true;
}

static int launch_ik_order(const IkOrderArgs &a, hipStream_t s) {
    constexpr int G = 8;
    const size_t lds = ((size_t)PM_WAVE * ik_deep_row(G) + 5 * (size_t)a.J) * sizeof(float);
    const int64_t ntiles = (a.F + PM_WAVE - 1) / PM_WAVE;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("from_root_positions: grid too large"); return PM_EUNSUPPORTED; }
    const int nf = a.nfar <= 4 ? 4 : (a.nfar <= 8 ? 8 : (a.nfar <= 12 ? 12 : 16));
    const bool two = 9 * ((lds + 511) & ~(size_t)511) > kMaxLds;  // fewer than nine waves fit a CU (LDS is granted in 512-byte units)
    set_kernel_name("void pm::from_root_positions_order_kernel<%d, %d, %s>(pm::IkOrderArgs)", G, nf, tf(two));
#define PM_IKO_LAUNCH(N, T)                                                                                        \
    {                                                                                                              \
        if (int e = allow_lds(from_root_positions_order_kernel<G, N, T>, lds)) return e;                          \
        hipLaunchKernelGGL((from_root_positions_order_kernel<G, N, T>), dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a); \
    }
    if (two) { if (nf == 4) PM_IKO_LAUNCH(4, true) else if (nf == 8) PM_IKO_LAUNCH(8, true) else if (nf == 12) PM_IKO_LAUNCH(12, true) else PM_IKO_LAUNCH(16, true) }
    else { if (nf == 4) PM_IKO_LAUNCH(4, false) else if (nf == 8) PM_IKO_LAUNCH(8, false) else if (nf == 12) PM_IKO_LAUNCH(12, false) else PM_IKO_LAUNCH(16, false) }
#undef PM_IKO_LAUNCH
    return PM_AFTER_LAUNCH("from_root_positions (lane per frame, any order) launch");
}

// (round 6: a joint may follow its parent in the very next step on ANY chain -- the kernel reads the parent's slot at the top of the step, see `fetch`;
// PM_IK_RELAXED = 0 on the tuning build: rounds 2-5's rule, two steps unless on the parent's own chain)
static int ik_schedule_rule(const Topo16 &t, const int J, uint8_t *sched, const int C, const bool relaxed, int *cost) {
    int height[PM_MAX_JOINTS], done_step[PM_MAX_JOINTS], done_chain[PM_MAX_JOINTS], items = 0;
    for (int j = 0; j < J; ++j) { height[j] = 0; done_step[j] = -1; done_chain[j] = -1; }
    for (int j = J - 1; j >= 0; --j) {
        const int nc = t.cstart[j + 1] - t.cstart[j];
        if (nc > 0) {
            ++items;
            height[j] += nc;
            if (j > 0 && height[t.parent[j]] < height[j]) height[t.parent[j]] = height[j];
        }
    }
    int left = items, K = 0;
    for (int st = 0; left > 0; ++st) {
        if (C * (st + 1) > 512) return 0;
        for (int k = 0; k < C; ++k) {
            int best = -1, best_on = 0;
            for (int j = 0; j < J; ++j) {
                if (done_step[j] >= 0 || t.cstart[j + 1] <= t.cstart[j]) continue;
                int on = 0;
                if (j > 0) {
                    const int p = t.parent[j];
                    if (done_step[p] < 0 || done_step[p] == st) continue;
                    if (done_step[p] == st - 1) {
                        if (done_chain[p] == k) on = 1;
                        else if (!relaxed) continue;
                    }
                }
                if (best < 0 || on > best_on || (on == best_on && height[j] > height[best])) { best = j; best_on = on; }
            }
            sched[C * st + k] = (uint8_t)(best < 0 ? 255 : best);
            if (best >= 0) { done_step[best] = st; done_chain[best] = k; --left; }
        }
        K = st + 1;
    }
    // what the walk costs: a step is one alignment plus as many roll corrections as its busiest chain has further children (~170 / ~150 instructions)
    *cost = 0;
    for (int st = 0; st < K; ++st) {
        int rolls = 0;
        for (int k = 0; k < C; ++k) {
            const int j = sched[C * st + k];
            if (j != 255) { const int nx = t.cstart[j + 1] - t.cstart[j] - 1; rolls = nx > rolls ? nx : rolls; }
        }
        *cost += 170 + 150 * rolls;
    }
    return (C > 2 || 4 * K <= 3 * items) ? K : 0;
}
// Both rules are valid programs for the kernel; the list scheduler is greedy, so neither is always the shorter walk (measured, random trees: the next-step rule
// +2...5 % at 22 / 40 / 64 / 96 / 128 joints, -2...4 % at 55 / 72 / 80): the host builds both and keeps the cheaper one by the model above.
// PM_IK_RELAXED (tuning build): 0 / 1 = that rule only.
static int ik_schedule(const Topo16 &t, const int J, uint8_t *sched, const int C = 2) {
    const int only = tune_env("PM_IK_RELAXED", -1);
    uint8_t alt[512];
    int c_two = 1 << 30, c_next = 1 << 30;
    const int k_next = only == 0 ? 0 : ik_schedule_rule(t, J, sched, C, true, &c_next);
    if (only == 1) return k_next;
    const int k_two = ik_schedule_rule(t, J, alt, C, false, &c_two);
    if (k_two > 0 && (k_next == 0 || c_two < c_next)) { memcpy(sched, alt, sizeof(alt)); return k_two; }
    return k_next;
}

template <int FPW, int C>
static int launch_ik(const IkArgs &a, bool vec, hipStream_t s) {
    const size_t lds = ((size_t)FPW * ik_frame_stride(a.J) + ik_tables_floats(a.J)) * sizeof(float) + (size_t)(a.J + 2) * C * 16;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    // records per lane of one tile -> the pipelined instantiation that holds them in registers (3 VGPRs each)
    const int nl = (FPW * a.J + PM_WAVE - 1) / PM_WAVE;
    // (the register file bounds residency here: 12 / 24 / 28 / 56 records -> 148 / 246 / ~210 / 415 VGPRs with two chains)
    int cap = nl <= 12 ? 12 : (nl <= 24 ? 24 : (nl <= 28 ? 28 : (nl <= 56 ? 56 : 0)));
    if (tune_env("PM_IK_PIPE", 1) == 0) cap = 0;  // PM_TUNING build only
    // (measured, tiles per workgroup 1 / 2 / 4: 2^18 x 52 with 28 records per lane 165 / 173 / 186 us; 2^20 x 22 196 / 191 / 216 us, 2^18 x 128 899 / 863 / 845 us; unpipelined 206 / 992)
    int nt = cap == 0 ? 1 : (ntiles >= 4096 ? (cap <= 24 ? 2 : (cap == 28 ? 1 : 4)) : 1);
    nt = tune_env("PM_IK_NT", nt);
    if (nt < 1 || cap == 0) nt = 1;
    const int64_t ngroups = (ntiles + nt - 1) / nt;
    const int64_t grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("from_root_positions: grid too large"); return PM_EUNSUPPORTED; }
    set_kernel_name("void pm::from_root_positions_kernel<%d, %s, %d, %d>(pm::IkArgs, int)", FPW, tf(vec), cap, C);
#define PM_IK_LAUNCH(V, N)                                                          \
    {                                                                               \
        auto k = from_root_positions_kernel<FPW, V, N, C>;                          \
        if (int e = allow_lds(k, lds)) return e;                                    \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);  \
    }
    if (vec) { if (cap == 12) PM_IK_LAUNCH(true, 12) else if (cap == 24) PM_IK_LAUNCH(true, 24) else if (cap == 28) PM_IK_LAUNCH(true, 28) else if (cap == 56) PM_IK_LAUNCH(true, 56) else PM_IK_LAUNCH(true, 0) }
    else { if (cap == 12) PM_IK_LAUNCH(false, 12) else if (cap == 24) PM_IK_LAUNCH(false, 24) else if (cap == 28) PM_IK_LAUNCH(false, 28) else if (cap == 56) PM_IK_LAUNCH(false, 56) else PM_IK_LAUNCH(false, 0) }
#undef PM_IK_LAUNCH
    return PM_AFTER_LAUNCH("from_root_positions launch");
}

}  // namespace pm

extern "C" int pm_from_root_positions_f32(const float *positions, const int32_t *parents, const float *offsets, int64_t F,
                                          int32_t J, float *rotations, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "from_root_positions: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(positions && parents && offsets && rotations, "from_root_positions: null pointer");
    IkArgs a;
    a.pos = positions; a.offsets = offsets; a.out = rotations; a.F = F; a.J = J;
    a.ablate = tune_env("PM_IK_ABLATE", 0);
    Parents p;
    if (int e = pack_parents(parents, J, p)) return e;
    // children in index order, exactly the lists the reference builds (skeleton.py:121-125)
    int cnt[PM_MAX_JOINTS + 1] = {0};
    for (int32_t j = 1; j < J; ++j) cnt[p.p[j] + 1]++;
    for (int32_t j = 0; j < J; ++j) cnt[j + 1] += cnt[j];
    int fill[PM_MAX_JOINTS];
    for (int32_t j = 0; j <= J; ++j) a.topo.cstart[j] = (int16_t)cnt[j];
    for (int32_t j = 0; j < J; ++j) { fill[j] = cnt[j]; a.topo.parent[j] = (int16_t)p.p[j]; a.topo.clist[j] = 0; }
    for (int32_t j = 1; j < J; ++j) a.topo.clist[fill[p.p[j]]++] = (int16_t)j;
    const bool vec = aligned16(positions) && aligned16(rotations);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one lane per frame, joints streamed through a ring of LDS slots (from_root_positions_order_kernel) where that pays and the table fits
    // its window / queue / register sets.  PM_IK_ORDER (PM_TUNING build only): 0 = never, 1 = whenever the table fits
    if (const int ord = tune_env("PM_IK_ORDER", -1); aligned16(rotations) && J >= 2 && ord != 0 && (ord == 1 || (ik_order_wanted(J) && lane_per_frame_pays(F, J, kIkOrderMinJointFrames)))) {
        IkOrderArgs oa;
        if (ik_order_plan(a.topo, J, oa)) {
            oa.pos = positions; oa.offsets = offsets; oa.out = rotations; oa.F = F; oa.J = J;
            return launch_ik_order(oa, s);
        }
    }
    const size_t per_frame = (size_t)ik_frame_stride(J) * sizeof(float), fixed = (size_t)ik_tables_floats(J) * sizeof(float) + (size_t)(J + 2) * 32 + 256;
    {
        const int v = tune_env("PM_IK_FPW", 0);  // PM_TUNING build only
        a.K = 0;
        if (v == 64 && 64 * per_frame + fixed <= kMaxLds) return launch_ik<64, 1>(a, vec, s);
        if (v == 32 && 32 * per_frame + fixed <= kMaxLds) return launch_ik<32, 1>(a, vec, s);
        if (v == 16) return launch_ik<16, 1>(a, vec, s);
    }
    // two chains per frame when the tree is wide enough for the schedule to pay (PM_IK_CHAINS=1 in the tuning build: never)
    const int chains = tune_env("PM_IK_CHAINS", 0);  // PM_TUNING build only: 1 = never several, 2 / 4 = that many if the schedule exists
    a.K = (J <= 254 && chains != 1 && chains != 4) ? ik_schedule(a.topo, J, a.sched) : 0;
    // FOUR chains (16 frames per wave) for long skeletons whose tree is wide enough to shorten the walk again: beyond ~56 joints the
    // 32-frame image leaves room for two to four waves per CU and the walk is all the kernel waits for
    // -- and for a clip of real length whatever the joint count: up to 2^15 frames a launch is a few waves per CU and takes as long as ONE
    // wave's walk, which four chains shorten (the 22-joint body 10.5 -> 8.9 us, SMPL-H 20.8 -> 16.3 us at 2^10...2^14 frames; from 2^16
    // frames on two chains -- twice the frames in a wave -- are ahead again: 14.3 against 18.8 us on the body)
    const bool clip = F <= kIkClipFrames;
    if (J <= 254 && (chains == 4 || (chains == 0 && (J > kIkFourChainsMinJ || clip)))) {
        uint8_t s4[512];
        const int K4 = ik_schedule(a.topo, J, s4, 4);
        if (K4 > 0 && (chains == 4 || a.K == 0 || 10 * K4 <= 7 * a.K || (clip && K4 < a.K)) && 2 * (16 * per_frame + fixed + (size_t)(J + 2) * 32) <= kMaxLds) {
            a.K = K4;
            memcpy(a.sched, s4, sizeof(s4));
            return launch_ik<16, 4>(a, vec, s);
        }
    }
    if (a.K > 0 && 2 * (32 * per_frame + fixed) <= kMaxLds) return launch_ik<32, 2>(a, vec, s);
    a.K = 0;
    if (4 * (64 * per_frame + fixed) <= kMaxLds) return launch_ik<64, 1>(a, vec, s);  // every lane busy, >= 4 waves per CU
    if (4 * (32 * per_frame + fixed) <= kMaxLds) return launch_ik<32, 1>(a, vec, s);
    if (2 * (16 * per_frame + fixed) <= kMaxLds) return launch_ik<16, 1>(a, vec, s);
    if (4 * per_frame + fixed <= kMaxLds) return launch_ik<4, 1>(a, vec, s);
    set_error("from_root_positions: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
}
