// fkwide.hip -- fk for long skeletons whose tree is WIDE (reference: pymotion/ops/skeleton.py:16-61; no topology cliff there).
//
// The walks of fk.hip put 4-20 FRAMES in a wave and visit a frame's joints one after the other; beyond 128 joints that is a tile of
// 4 x 48 J bytes (one wave a CU at J = 512) walking J dependent steps, and the streamed walk that replaces it (fk_stream_kernel) needs the
// cross-chunk branch points of the tree to fit eight register sets -- which a bushy tree (parents[j] uniform in [0, j): depth ~ 2 ln J,
// almost every parent in an earlier chunk) does not: 34 % of the HBM spec at J = 129, 23 % at 256, 8.6 % at 512 (profiles/r04_fk_long_sweep.txt).
//
// Here a wave owns ONE frame and its lanes run over the JOINTS: the host list-schedules the tree into steps of up to 16 joints whose
// parents are finished (critical path first), a quad takes one joint of the step -- lane r row r of [R | p], the joint's local [L | t]
// shared across the quad through the DPP operand of the multiply-adds exactly as in tree_walk_q4 -- and the frame's image (48 J bytes,
// the output layout) is the only LDS a wave needs: 24 KB at J = 512, six waves a CU, ~J / 16 + depth steps instead of J.
//   * slot j of the rotation image holds L_j until its step overwrites it with R_j; slot j of the position image holds the offset t_j
//     until the step overwrites it with p_j (so [L | t] row r is what lane r reads, and nothing but the image is in LDS);
//   * the root takes no step (its slot holds L_0 = R_0 as parked, its position is the caller's); slots J (identity | 0) and J + 1 are what
//     idle quads of a step read and write;
//   * the step list rides in the kernarg segment (one word per quad and step: joint | parent << 16); a quad's words are loaded into
//     registers once per workgroup, which then takes a few consecutive frames: the walk waits for LDS only, and the next frame's
//     quaternions are in flight while a frame walks;
//   * the next step's [L | t] row is requested before the current step's arithmetic (its slot is written by nobody else);
//   * a frame's rows start wherever 36 J f / 12 J f bytes put them: the image sits at the same offset modulo 16 bytes in LDS, so the body
//     leaves as dwordx4 and only the first / last < 4 floats of a row go one by one (XCD-contiguous frames: neighbours merge in L2).
// The arithmetic is the other walks' to the bit (same products, same order, same local rotations, PREC_DYN per frame: float64 local
// rotations and the fixed-point translation chain when offsets or the root are big, see fk.hip).
#include "common.hpp"

namespace pm {

constexpr int kFwSteps = 48;               // steps the list holds
constexpr int kFwStride = kFwSteps + 8;    // words per quad in the kernarg segment: its steps, then idle words for the look-ahead (16 x 56 x 4 = 3584 B)
struct FkWideArgs {
    const float *rot, *root_pos, *offsets;  // rot: [F, J, 4] quaternions or [F, J, 3, 2] ortho6d records; offsets: [J, 3] or (PFO) [F, J, 3]
    float *pos, *rotmats, *quat_out;        // quat_out: [F, J, 4] or null (ortho6d source only)
    int64_t F;
    int32_t J, depth, nsteps, ablate;
    float eps;                              // ortho6d Gram-Schmidt floor
    int32_t pad_;
    uint32_t jobs[16 * kFwStride];  // [quad][step]: joint | parent << 16 -- lane t of a quad keeps the word of step 4 g + t in register g
};

// Host: list scheduling, at most 16 joints a step, a joint at the earliest one step after its parent; ready joints with the longest
// path below them first (a binary heap on (height, lower index first): J log J per call).  Returns the number of steps (-1: more than max_steps).
// `width` joints a step, at most `max_steps` steps; `jobs`: [steps + 2][width] words joint | parent << 16.  A slot without a joint holds the idle
// word J + 1 | J << 16 (fk_wide_kernel's spare slots) or, with `dup_idle`, the step's first joint again (tree_walk_w4 of fk.hip: same reads, same writes).
int fk_wide_plan(const Parents &par, const int J, const int width, const int max_steps, const bool dup_idle, uint32_t *jobs) {
    int height[PM_MAX_JOINTS], heap[PM_MAX_JOINTS], nheap = 0, fresh[PM_MAX_JOINTS], nfresh = 0;
    int first_child[PM_MAX_JOINTS], sibling[PM_MAX_JOINTS];
    for (int j = 0; j < J; ++j) { height[j] = 0; first_child[j] = -1; sibling[j] = -1; }
    for (int j = J - 1; j > 0; --j) {
        const int p = par.p[j];
        if (height[j] + 1 > height[p]) height[p] = height[j] + 1;
        sibling[j] = first_child[p];
        first_child[p] = j;
    }
    auto before = [&](const int x, const int y) { return height[x] > height[y] || (height[x] == height[y] && x < y); };
    auto push = [&](const int j) {
        int i = nheap++;
        heap[i] = j;
        while (i > 0 && before(heap[i], heap[(i - 1) / 2])) { const int t = heap[i]; heap[i] = heap[(i - 1) / 2]; heap[(i - 1) / 2] = t; i = (i - 1) / 2; }
    };
    auto pop = [&]() {
        const int top = heap[0];
        heap[0] = heap[--nheap];
        for (int i = 0;;) {
            int m = i;
            const int l = 2 * i + 1, r = 2 * i + 2;
            if (l < nheap && before(heap[l], heap[m])) m = l;
            if (r < nheap && before(heap[r], heap[m])) m = r;
            if (m == i) break;
            const int t = heap[i]; heap[i] = heap[m]; heap[m] = t;
            i = m;
        }
        return top;
    };
    // the root needs no step: its slot holds L_0 = R_0 as parked, its position is the caller's (skeleton.py:49 -- the reference copies both;
    // a product with an identity seed would turn a partly non-finite L_0 into NaN rows)
    for (int c = first_child[0]; c >= 0; c = sibling[c]) push(c);
    int steps = 0, done = 1;
    const uint32_t idle = (uint32_t)(J + 1) | ((uint32_t)J << 16);
    while (done < J) {
        if (steps == max_steps) return -1;
        nfresh = 0;
        for (int k = 0; k < width; ++k) {
            if (nheap > 0) {
                const int j = pop();
                jobs[steps * width + k] = (uint32_t)j | ((uint32_t)par.p[j] << 16);
                for (int c = first_child[j]; c >= 0; c = sibling[c]) fresh[nfresh++] = c;  // ready from the next step on
                ++done;
            } else {
                jobs[steps * width + k] = dup_idle ? jobs[steps * width] : idle;
            }
        }
        for (int i = 0; i < nfresh; ++i) push(fresh[i]);
        ++steps;
    }
    for (int k = 0; k < 2 * width; ++k) jobs[steps * width + k] = dup_idle ? 0u : idle;  // (two steps of padding: what a look-ahead may read past the list)
    return steps;
}

// LDS image (n floats, same offset modulo 16 bytes as its place in HBM) -> HBM: < 4 floats, dwordx4 ..., < 4 floats
__device__ __forceinline__ void fw_row_out(float *__restrict__ g, const float *lds, const int n, const int lane) {
    int h = (int)((4u - (unsigned)((reinterpret_cast<uintptr_t>(g) >> 2) & 3u)) & 3u);
    h = h < n ? h : n;
    if (lane < h) g[lane] = lds[lane];
    const int n4 = (n - h) >> 2;
    v4f *g4 = reinterpret_cast<v4f *>(g + h);
    const v4f *l4 = reinterpret_cast<const v4f *>(lds + h);
    int i = lane;
    for (; i + 3 * PM_WAVE < n4; i += 4 * PM_WAVE) {  // four LDS reads in flight, then their stores
        const v4f x0 = l4[i], x1 = l4[i + PM_WAVE], x2 = l4[i + 2 * PM_WAVE], x3 = l4[i + 3 * PM_WAVE];
        __builtin_nontemporal_store(x0, g4 + i);
        __builtin_nontemporal_store(x1, g4 + i + PM_WAVE);
        __builtin_nontemporal_store(x2, g4 + i + 2 * PM_WAVE);
        __builtin_nontemporal_store(x3, g4 + i + 3 * PM_WAVE);
    }
    for (; i < n4; i += PM_WAVE) __builtin_nontemporal_store(l4[i], g4 + i);
    const int k = h + (n4 << 2) + lane;
    if (k < n) g[k] = lds[k];
}

// The walk.  Lanes (k, r), r < 3: quad k's joint of every step, row r.  sRot / sPos: the frame's image.  JW: this quad's words -- lane t of the
// quad holds the word of step 4 g + t in JW[g], a quad broadcast (DPP) hands it out -- in REGISTERS for the life of the workgroup (loaded
// once, before the first frame; NG + 1 of them, picked by the group counter): the list is kernarg memory behind the vector
// L1, which the frames' streaming loads and stores keep cold -- a word requested one step ahead cost the walk an L2 round trip per step
// (J = 512 at 2^18 frames: 2450 us; four steps ahead: 2060 us), and a walk without vector-memory waits is what lets the NEXT frame's
// quaternions be in flight while this frame walks.
template <bool FX, int NG>
__device__ __forceinline__ void fw_walk(float *sRot, float *sPos, const uint32_t (&JW)[NG + 1], const int nsteps, const int lane, const float S, const bool poison) {
    const int r = (lane & 3) < 3 ? (lane & 3) : 2;  // lane 3 of a quad shadows lane 2 (same reads, same writes): it holds a quarter of the quad's words
    // out = p0 * bcast_0(l) + p1 * bcast_1(l) + p2 * bcast_2(l), l = this lane's element of its row of [L | t] (lane i of the quad: row i);
    // the four products of a step -- three rows of L and t against the parent's row -- as ONE block: the rows travel from step to step through
    // register copies, a VGPR written by the VALU needs two wait states before a DPP read, and nothing can be scheduled into the block
    auto dot4 = [](const float l0, const float l1, const float l2, const float tr, const float p0, const float p1, const float p2, float &g0, float &g1,
                   float &g2, float &dt) __attribute__((always_inline)) {
        asm volatile("s_nop 1\n\t"
                     "v_mul_f32_dpp %0, %4, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mul_f32_dpp %1, %5, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mul_f32_dpp %2, %6, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_mul_f32_dpp %3, %7, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %0, %4, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, %5, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %2, %6, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %3, %7, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %0, %4, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %1, %5, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %2, %6, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %3, %7, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
                     : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(dt)
                     : "v"(l0), "v"(l1), "v"(l2), "v"(tr), "v"(p0), "v"(p1), "v"(p2));
    };
    // word of step 4 g + T: lane T of the quad holds it in JW[g]
    auto word = [](const uint32_t v, auto t) __attribute__((always_inline)) {
        constexpr int T = decltype(t)::value;
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, T * 0x55, 0xf, 0xf, true);  // quad_perm:[T,T,T,T]
    };
    uint32_t w = word(JW[0], IntC<0>{});
    float *aj = sRot + __umul24(w & 0xffffu, 9u) + r * 3, *pj = sPos + __umul24(w & 0xffffu, 3u) + r;
    float l0 = aj[0], l1 = aj[1], l2 = aj[2], tr = *pj;
    // one step: the joint whose [L | t] row is in (l0, l1, l2, tr) -- parent `w >> 16` -- and the request for the row of the step after (`wn`)
    auto step = [&](const uint32_t wn) __attribute__((always_inline)) {
        const unsigned par = w >> 16;
        const float *ap = sRot + __umul24(par, 9u) + r * 3, *pp = sPos + __umul24(par, 3u) + r;
        const float p0 = ap[0], p1 = ap[1], p2 = ap[2], pt = *pp;
        // the next step's [L | t] row: its slot is written by that step only
        float *an = sRot + __umul24(wn & 0xffffu, 9u) + r * 3, *pn = sPos + __umul24(wn & 0xffffu, 3u) + r;
        const float n0 = an[0], n1 = an[1], n2 = an[2], nt = *pn;
        float g0, g1, g2, dt;
        dot4(l0, l1, l2, tr, p0, p1, p2, g0, g1, g2, dt);
        float gt;
        if (FX) gt = __int_as_float(__float_as_int(pt) + (int)__builtin_rintf(dt * S));
        else {
            gt = dt + pt;
            if (poison) poison_row(pt, g0, g1, g2);  // (wave-uniform) a NaN / Inf translation in the frame: the reference's p_parent[r] * 0, see fk.hip's tree_walk
        }
        aj[0] = g0; aj[1] = g1; aj[2] = g2;
        *pj = gt;
        aj = an; pj = pn;
        l0 = n0; l1 = n1; l2 = n2; tr = nt;
        w = wn;
    };
#pragma nounroll
    for (int g = 0; g * 4 < nsteps; ++g) {  // (the list is padded with idle words: a last group of fewer than four steps runs idle ones)
        const uint32_t cur = JW[g], nxt = JW[g + 1];
        step(word(cur, IntC<1>{}));
        step(word(cur, IntC<2>{}));
        step(word(cur, IntC<3>{}));
        step(word(nxt, IntC<0>{}));
    }
}

// NB: batches of 64 joints (J <= 64 NB); NG: groups of four steps (nsteps <= 4 NG).  A workgroup (one wave) takes `nt` consecutive frames.
// SRC 0 / 1: quaternions / ortho6d records (rotations/ortho6d.py:50-64 -> fk: the fused kernel of BASELINE config 4, here for long skeletons);
// QOUT: the quaternions ortho6d.to_quat would have returned, stored straight from the conversion's registers; PFO: per-frame offsets.
template <int PREC, int NB, int NG, int SRC, bool QOUT, bool PFO>
__global__ __launch_bounds__(PM_WAVE) void fk_wide_kernel(const FkWideArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool DYN = (PREC & PREC_DYN) != 0;
    // the next frame's records are requested before a frame's walk only where they are one quaternion a joint: six floats (or three more of
    // offsets) a joint in flight across the walk would double the kernel's registers
    constexpr bool AHEAD = SRC == 0 && !PFO;
    const int lane = threadIdx.x, J = a.J;
    const int64_t ngroups = (a.F + nt - 1) / nt;
    const int64_t grp = PM_ABLATED(a, 4) ? ((int64_t)blockIdx.x < ngroups ? (int64_t)blockIdx.x : -1) : xcd_tile(ngroups);
    if (grp < 0) return;
    const int64_t f0 = grp * nt;
    const int nf = (int)((a.F - f0) < nt ? (a.F - f0) : nt);
    const int nR = ((J + 2) * 9 + 3 + 3) & ~3;  // floats reserved for the rotation image (placed 0..3 floats into it)

    // a frame's global loads: rotation records (one per lane and batch), the root position, (PFO) the frame's offsets
    v4f q[SRC == 0 ? NB : 1];
    v2f x[SRC == 1 ? NB : 1][3];
    v3f_a4 po[PFO ? NB : 1];
    float gp = 0.0f;
    auto issue = [&](const int64_t f) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int j = b * PM_WAVE + lane, jc = j < J ? j : J - 1;
            if (b * PM_WAVE < J) {
                if constexpr (SRC == 0) q[b] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a.rot) + f * J + jc);
                else {
                    const v2f *p = reinterpret_cast<const v2f *>(a.rot) + (f * J + jc) * 3;  // 24-byte records are 8-byte aligned
                    x[b][0] = __builtin_nontemporal_load(p); x[b][1] = __builtin_nontemporal_load(p + 1); x[b][2] = __builtin_nontemporal_load(p + 2);
                }
                if constexpr (PFO) po[b] = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(a.offsets + (f * J + jc) * 3));
            }
        }
        gp = a.root_pos[f * 3 + (lane < 3 ? lane : 2)];
    };
    issue(f0);
    // once per workgroup: the step list (this quad's words) and -- shared offsets -- the table (one joint per lane and batch, like the rotations;
    // slots J and J + 1, what idle quads read and write, get zeros), all requested before the first is used
    uint32_t JW[NG + 1];  // lane (k, t): quad k's word of step 4 g + t
#pragma unroll
    for (int g = 0; g <= NG; ++g) JW[g] = a.jobs[(lane >> 2) * kFwStride + 4 * g + (lane & 3)];
    float o[PFO ? 1 : NB + 1][3];
    if constexpr (!PFO) {
#pragma unroll
        for (int b = 0; b <= NB; ++b) {  // (loads only: a select behind each load would make the table nine round trips instead of one)
            const int j = b * PM_WAVE + lane, jc = j < J ? j : J - 1;
            if (b * PM_WAVE < J + 2) { o[b][0] = a.offsets[3 * jc]; o[b][1] = a.offsets[3 * jc + 1]; o[b][2] = a.offsets[3 * jc + 2]; }
        }
    }
    // settle the list and the table here: left pending, the walk's first indexed read of JW would wait for EVERY vector-memory operation in
    // flight -- the next frame's quaternions among them
#pragma unroll
    for (int g = 0; g <= NG; ++g) asm volatile("" : "+v"(JW[g]));
    bool tbig = false;
    float tsum = 0.0f, tmx = 0.0f;
    if constexpr (!PFO) {
#pragma unroll
        for (int b = 0; b <= NB; ++b) {
            const int j = b * PM_WAVE + lane;
            const bool none = j == 0 || j >= J;  // offsets[0] is ignored (skeleton.py:49); idle slots: no translation
            if (b * PM_WAVE < J + 2) { o[b][0] = none ? 0.0f : o[b][0]; o[b][1] = none ? 0.0f : o[b][1]; o[b][2] = none ? 0.0f : o[b][2]; }
        }
#pragma unroll
        for (int b = 0; b <= NB; ++b) {
            const int j = b * PM_WAVE + lane;
            if (b * PM_WAVE < J && j < J) {
                const v4f cj = v4f{0.0f, o[b][0], o[b][1], o[b][2]};
                const float l1 = const_l1(cj);
                tbig = tbig || const_is_big(cj); tsum += l1; tmx = (l1 > tmx || l1 != l1) ? l1 : tmx;
            }
        }
    }
    // what the joint table says about the arithmetic a frame needs (PREC_DYN, see fk_tile): the same for every frame of the workgroup
    const bool table_big = __builtin_amdgcn_ballot_w64(tbig) != 0;
    const float bsum = wave_sum(tsum), bmax = (float)a.depth * wave_max(tmx);  // (NaN sticks in both)
    const float tbound_table = (bmax < bsum) ? bmax : bsum;

    for (int i = 0; i < nf; ++i) {
        const int64_t f = f0 + i;
        // the image: rotations [J + 2][9], positions [J + 2][3], each at its row's offset modulo four floats (the host checked the base pointers)
        const int64_t gr = f * J * 9, gq = f * J * 3;
        float *sRot = smem + (int)(gr & 3), *sPos = smem + nR + (int)(gq & 3);
        const float gpf = gp;  // this frame's root position (gp is refilled by the next frame's loads)

        // PREC_DYN: does this frame need the float64 rotations and the fixed-point chain?  (fk_tile's test, on one frame)
        bool big = false, poison = false;
        FxScale fx = {1.0f, 1.0f};
        if constexpr (DYN || (PREC & PREC_FX)) {
            bool mine = lane < 3 && !(fabsf(gpf) < kBigRoot);
            float tbound = tbound_table;
            if constexpr (PFO) {  // the frame's own offsets: their largest magnitude (NaN sticks), bound 3 depth max |t| like fk_tile
                float tmax = 0.0f;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int j = b * PM_WAVE + lane;
                    if (b * PM_WAVE < J && j > 0 && j < J) {
                        const float m = fmaxf(fmaxf(fabsf(po[b].x), fabsf(po[b].y)), fabsf(po[b].z));
                        const bool nan = po[b].x != po[b].x || po[b].y != po[b].y || po[b].z != po[b].z;
                        tmax = (nan || tmax != tmax) ? __builtin_nanf("") : ((m > tmax) ? m : tmax);
                    }
                }
                mine = mine || !(tmax < kBigOffset);
                tbound = 3.0f * (float)a.depth * wave_max(tmax);
            }
            big = table_big || __builtin_amdgcn_ballot_w64(mine) != 0 || !DYN;
            if (big) { big = fx_scale(tbound, lane < 3 ? fabsf(gpf) : 0.0f, fx); poison = !big; }
        }

        auto rest = [&](auto mode) __attribute__((always_inline)) {
            constexpr int M = decltype(mode)::value;
            constexpr bool FX = (M & PREC_FX) != 0;
            // the offsets into the position slots (the walk overwrites them with the positions), identity | zeros into the idle slots
#pragma unroll
            for (int b = 0; b <= NB; ++b) {
                const int j = b * PM_WAVE + lane;
                if (b * PM_WAVE < J + 2 && j < J + 2) {
                    if constexpr (PFO) {
                        const bool none = j == 0 || j >= J;
                        const int bb = b < NB ? b : NB - 1;  // (batch NB only ever holds idle slots)
                        sPos[3 * j] = none ? 0.0f : po[bb].x; sPos[3 * j + 1] = none ? 0.0f : po[bb].y; sPos[3 * j + 2] = none ? 0.0f : po[bb].z;
                    } else {
                        sPos[3 * j] = o[b][0]; sPos[3 * j + 1] = o[b][1]; sPos[3 * j + 2] = o[b][2];
                    }
                }
            }
            if (lane < 18) sRot[J * 9 + lane] = (lane == 0 || lane == 4 || lane == 8) ? 1.0f : 0.0f;  // idle quads: parent slot J (identity), own slot J + 1 (zeros)
            bool bad = false;  // FX only: a non-finite local rotation somewhere in the frame
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b * PM_WAVE < J) {  // wave-uniform
                    const int j = b * PM_WAVE + lane, jc = j < J ? j : J - 1;
                    float L[9];
                    float *slot = sRot + jc * 9;
                    if constexpr (SRC == 0) {
                        const float qi[4] = {q[b].x, q[b].y, q[b].z, q[b].w};
                        local_from_quat<M>(qi, L);
                        if (FX) bad = bad || !(fabsf(qi[0]) + fabsf(qi[1]) + fabsf(qi[2]) + fabsf(qi[3]) < 3e38f);  // NaN / Inf input
                        if (j < J && !PM_ABLATED(a, 8)) {  // (& 8, tuning build: without phase A's LDS writes)
#pragma unroll
                            for (int e = 0; e < 9; ++e) slot[e] = L[e];
                        }
                    } else {
                        const float xx[6] = {x[b][0].x, x[b][0].y, x[b][1].x, x[b][1].y, x[b][2].x, x[b][2].y};
                        float Q[4];
                        const bool ill = local_from_o6d<QOUT, M>(xx, a.eps, L, Q);
                        if (FX) bad = bad || !(fabsf(xx[0]) + fabsf(xx[1]) + fabsf(xx[2]) + fabsf(xx[3]) + fabsf(xx[4]) + fabsf(xx[5]) < 3e38f);
                        float *qslot = QOUT ? a.quat_out + (f * J + jc) * 4 : nullptr;
                        if (j < J) {
#pragma unroll
                            for (int e = 0; e < 9; ++e) slot[e] = L[e];
                            if constexpr (QOUT) { if (!ill) __builtin_nontemporal_store(v4f{Q[0], Q[1], Q[2], Q[3]}, reinterpret_cast<v4f *>(qslot)); }
                        }
                        // the rare float64 redo of a degenerate record, on its parked slot (in-order DS: after the plain values); its quaternion's only store
                        o6d_redo_ill<QOUT, false>(ill && j < J, xx, a.eps, slot, qslot);
                    }
                }
            }
            // the next frame's quaternions: in flight while this frame walks (the walk waits for LDS only)
            if (AHEAD && i + 1 < nf && !PM_ABLATED(a, 32)) issue(f + 1);
            const bool fixed = FX && __builtin_amdgcn_ballot_w64(bad) == 0;  // NaN / Inf rotations: the float walk propagates them
            if (lane < 3) sPos[lane] = (FX && fixed) ? __int_as_float((int)__builtin_rintf(gpf * fx.S)) : gpf;  // the root's position (its slot held offsets[0], which is ignored)
            wave_sync();
            if (!PM_ABLATED(a, 2)) {
                if (FX && fixed) fw_walk<true, NG>(sRot, sPos, JW, a.nsteps, lane, fx.S, false);
                else fw_walk<false, NG>(sRot, sPos, JW, a.nsteps, lane, 1.0f, poison);
            }
            wave_sync();
            if (FX && fixed) {  // fixed-point words -> fp32 in place; the root is the caller's value, bit for bit (skeleton.py:49)
                for (int e = lane; e < J * 3; e += PM_WAVE) sPos[e] = (float)__float_as_int(sPos[e]) * fx.invS;
                wave_sync();
                if (lane < 3) sPos[lane] = gpf;
                wave_sync();
            }
            if (!PM_ABLATED(a, 16)) {  // (& 16, tuning build: without the copy-out)
                fw_row_out(a.rotmats + gr, sRot, J * 9, lane);
                fw_row_out(a.pos + gq, sPos, J * 3, lane);
            }
            if (i + 1 < nf && (!AHEAD || PM_ABLATED(a, 32))) issue(f + 1);  // (& 32, tuning build: no prefetch across the walk)
            wave_sync();  // the next frame's parks come after this frame's copy-out reads (in-order DS; this keeps the compiler from mixing them)
        };
        if constexpr (DYN) {
            if (big) rest(IntC<PREC_F64 | PREC_FX>{});
            else rest(IntC<PREC & (PREC_RESID | PREC_F64)>{});
        } else {
            if ((PREC & PREC_FX) && !big) rest(IntC<PREC & ~PREC_FX>{});
            else rest(IntC<PREC>{});
        }
    }
}

template <int NB, int NG, int SRC, bool QOUT, bool PFO>
static int launch_fk_wide(const FkWideArgs &a, const int nt, const size_t lds, hipStream_t s) {
    constexpr int PREC = PREC_DYN | PREC_RESID;
    auto k = fk_wide_kernel<PREC, NB, NG, SRC, QOUT, PFO>;
    if (int e = allow_lds(k, lds)) return e;
    const int64_t ngroups = (a.F + nt - 1) / nt, grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("fk: %lld workgroups exceed the grid limit", (long long)grid); return PM_EUNSUPPORTED; }
    set_kernel_name("void pm::fk_wide_kernel<%d, %d, %d, %d, %s, %s>(pm::FkWideArgs, int)", PREC, NB, NG, SRC, tf(QOUT), tf(PFO));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
    return PM_AFTER_LAUNCH("fk launch");
}

// the variants other than the plain one (quaternion source, shared offsets) come with the long step list only
template <int NB>
static int launch_fk_wide_v(const FkWideArgs &a, const int nt, const size_t lds, const int src, const bool pfo, const bool short_list, hipStream_t s) {
    const bool qout = a.quat_out != nullptr;
    if (src == 0 && !pfo) return short_list ? launch_fk_wide<NB, 6, 0, false, false>(a, nt, lds, s) : launch_fk_wide<NB, 12, 0, false, false>(a, nt, lds, s);
    if (src == 0) return launch_fk_wide<NB, 12, 0, false, true>(a, nt, lds, s);
    if (!qout) return pfo ? launch_fk_wide<NB, 12, 1, false, true>(a, nt, lds, s) : launch_fk_wide<NB, 12, 1, false, false>(a, nt, lds, s);
    return pfo ? launch_fk_wide<NB, 12, 1, true, true>(a, nt, lds, s) : launch_fk_wide<NB, 12, 1, true, false>(a, nt, lds, s);
}

// fk on 16-byte aligned arrays: src_kind 0 quaternions / 1 ortho6d records (eps, quat_out or null), shared or per-frame offsets.  Returns false
// (nothing launched) when the tree needs more than kFwSteps steps, or more than max_quad_steps_per_joint_x10 / 10 quad-steps per joint (0: no
// such bound); true with rc set otherwise.
bool try_fk_wide(const int src_kind, const float *rot, const float *root_pos, const float *offsets, const bool offsets_per_frame, float *pos, float *rotmats,
                 float *quat_out, const float eps, const int64_t F, const int32_t J, const int32_t depth, const Parents &par, const int ablate,
                 const int max_quad_steps_per_joint_x10, hipStream_t s, int &rc) {
    FkWideArgs a;
    uint32_t list[(kFwSteps + 2) * 16];
    a.nsteps = fk_wide_plan(par, J, 16, kFwSteps, false, list);
    if (a.nsteps < 0) return false;
    // quad-steps the list spends per joint (1: every quad of every step has a joint; a deep, narrow tree idles most of them)
    if (max_quad_steps_per_joint_x10 > 0 && a.nsteps * 16 * 10 > max_quad_steps_per_joint_x10 * J) return false;
    const uint32_t idle = (uint32_t)(J + 1) | ((uint32_t)J << 16);
    for (int k = 0; k < 16; ++k)
        for (int st = 0; st < kFwStride; ++st) a.jobs[k * kFwStride + st] = st < a.nsteps ? list[st * 16 + k] : idle;
    a.rot = rot; a.root_pos = root_pos; a.offsets = offsets; a.pos = pos; a.rotmats = rotmats; a.quat_out = src_kind == 1 ? quat_out : nullptr;
    a.F = F; a.J = J; a.depth = depth; a.ablate = ablate; a.eps = eps; a.pad_ = 0;
    const size_t lds = ((size_t)(((J + 2) * 9 + 6) & ~3) + (size_t)(((J + 2) * 3 + 6) & ~3)) * sizeof(float);
    // frames per workgroup: the next frame's quaternions are requested before a frame's walk, so a workgroup wants a few -- while the launch
    // still has several workgroups per wave slot of the chip (PM_FKW_NT, PM_TUNING build only)
    int nt = F >= 131072 ? 4 : (F >= 32768 ? 2 : 1);
    nt = tune_env("PM_FKW_NT", nt);
    if (nt < 1) nt = 1;
    const bool short_list = a.nsteps <= 24;
    if (J <= 128) rc = launch_fk_wide_v<2>(a, nt, lds, src_kind, offsets_per_frame, short_list, s);
    else if (J <= 192) rc = launch_fk_wide_v<3>(a, nt, lds, src_kind, offsets_per_frame, short_list, s);
    else if (J <= 256) rc = launch_fk_wide_v<4>(a, nt, lds, src_kind, offsets_per_frame, short_list, s);
    else if (J <= 384) rc = launch_fk_wide_v<6>(a, nt, lds, src_kind, offsets_per_frame, short_list, s);
    else rc = launch_fk_wide_v<8>(a, nt, lds, src_kind, offsets_per_frame, short_list, s);
    return true;
}

}  // namespace pm

extern "C" int pm_fk_wide_plan_debug(const int32_t *parents, int32_t J, uint32_t *jobs) {
    PM_CHECK_ARGS(parents && jobs && J >= 1 && J <= PM_MAX_JOINTS, "fk_wide_plan: need parents, jobs and 1 <= J <= PM_MAX_JOINTS");
    pm::Parents par;
    if (int e = pm::pack_parents(parents, J, par)) return e < 0 ? e : -e;
    const int n = pm::fk_wide_plan(par, J, 16, pm::kFwSteps, false, jobs);
    return n < 0 ? PM_EUNSUPPORTED : n;
}
