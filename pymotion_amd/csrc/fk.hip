// fk.hip -- batched forward kinematics for gfx950 (reference: pymotion/ops/skeleton.py:16-61).
//
// One wave (= one workgroup) owns a tile of FPW consecutive frames (20 / 16 / 12 / 4 by skeleton size, see
// dispatch_fk; the text below uses 20) and a private LDS image of
// that tile's OUTPUTS, laid out exactly like HBM (`rotmats` tile then `pos` tile).  Two phases:
//
//  A. "local" phase, one lane per (frame, joint) element, 64 elements per pass: the lane loads its
//     quaternion straight from HBM (dwordx4, consecutive lanes = consecutive 16 B: perfectly
//     coalesced, all passes' loads issued before the first use), normalises it (skeleton.py:45),
//     expands it to the local rotation L (quat.py:276-317) and parks the 9 floats in the element's
//     own slot of the `rotmats` image.  All the transcendental / divide work happens here, fully
//     lane-parallel and off the dependency chain.
//
//  B. "row-parallel tree walk".  A world transform G_j = [R_j | p_j] obeys
//         G_j = G_parent(j) . [L_j | t_j]   =>   row r of G_j = (row r of R_parent) . [L_j | t_j] + [0 | p_parent[r]]
//     so the three rows of a frame's transforms never mix: THREE LANES own one frame (lane = 3 f + r),
//     60 of 64 lanes walk 20 skeletons at once, each lane carrying only its 4-float row of the
//     previous joint in registers.  Per joint a lane reads L_j (broadcast inside the lane triple,
//     fetched one joint ahead), does 12 FMAs and overwrites its row of slot j in place.  Joints are
//     visited in index order (parents[j] < j); when parents[j] == j-1 -- the common case in
//     DFS-ordered skeletons -- the parent row is already in registers, otherwise it is re-read from
//     the image.  That branch is wave-uniform; the skeleton constants {parent, offset} are staged once
//     per tile in LDS and read one joint ahead (a broadcast ds_read_b128), so nothing but DS traffic
//     shares lgkmcnt inside the walk and the per-joint critical path is 3 dependent FMAs.
//
// The finished image leaves with contiguous dwordx4 streaming stores.  Algorithmic HBM bytes per
// frame: 16 J + 12 read, 48 J written -- nothing is read or written twice, and the LDS
// footprint is just the output tile (48 J B per frame: 16.5 KiB for 16 frames of J = 22 -> 9 waves per CU).
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace pm {

enum FkSrc { SRC_QUAT = 0, SRC_O6D = 1 };
constexpr int kW4Groups = 10, kW4Steps = 4 * kW4Groups, kW4Stride = kW4Steps + 4;  // tree_walk_w4: steps its list holds (four to a register), words per slot

struct FkArgs {
    const float *src;       // [F,J,4] quats or [F,J,3,2] ortho6d
    const float *root_pos;  // [F,3]
    const float *offsets;   // [J,3] or [F,J,3]
    float *pos;             // [F,J,3]
    float *rotmats;         // [F,J,3,3]
    float *quat_out;        // [F,J,4] or nullptr (ortho6d source only)
    int64_t F;
    int32_t J;
    float eps;              // ortho6d Gram-Schmidt floor
    int32_t ablate;         // PM_TUNING build only (env PM_FK_ABLATE): 2 = no tree walk; always 0 in production
    int32_t pad;            // floats of padding per frame in each per-frame LDS region (0 or 4), set by dispatch_fk
    int32_t depth;          // edges on the longest root-to-leaf path: |p_j - root|_1 <= depth max_j |t_j|_1 (fixed-point scale, PREC_FX)
    int32_t wsteps;         // fk_pipe_kernel, four frames a wave: > 0 = walk the tree four JOINTS of a frame at a time over this many steps (tree_walk_w4)
    int32_t xchunk;         // tiles (fk_kernel) / tile groups (fk_pipe_kernel) per XCD chunk, 0 = one contiguous eighth per XCD (xcd_tile_chunked)
    int32_t pad2_;
    uint64_t fmap;          // tree_walk_q4, sixteen frames a wave: nibble q = the frame quad q walks (q4_frame_map: which eight frames share a half-wave)
    Parents parents;
    uint32_t wjobs[4 * kW4Stride];  // [slot][step]: joint | parent << 16 (a slot without a joint repeats the step's first one)
#ifdef PM_TUNING
    uint64_t *times;        // fk_pipe_kernel, env PM_FK_TIMES_PTR (a device address, tools/fk_xcd_time_probe.py): [workgroup][3] = XCC id, start, end (s_memrealtime)
#endif
};

// LDS bank conflicts of the walks: lanes of DIFFERENT frames touch the same joint slot in the same instruction, so the
// distance between frames in the image must not be a multiple of 8 floats (three-lane walk, 20 frames) resp. 16 floats
// (quad walk, 4 frames) -- which 9 J (rotations) and 3 J (positions, per-frame offsets) are exactly when J is a multiple
// of 8 resp. 16 (24-joint SMPL, 32, 64...: measured 46 % of peak at J = 32 against 60-65 % at J = 28 / 40 before,
// 62 % after).  Those skeletons get `pad` = 4 floats between frames (keeps 16-byte alignment: 9 J and 3 J are multiples
// of 4 then); the image is then no longer the HBM layout verbatim and phase A / copy-out address it per frame
// (image_slot / image_store).  Every other J keeps the linear image.  The choice is made in dispatch_fk.

// LDS floats per frame-joint: rot 9 + pos 3 (+ per-frame offsets 3) (+ quat_out 4)
template <int SRC, bool PFO, bool QOUT>
constexpr int fk_lds_floats() { return 12 + (PFO ? 3 : 0) + (QOUT ? 4 : 0); }

// quaternion -> local rotation of fk: normalise (skeleton.py:45, quat.py:423) then quat.py:276-317.
//
// PREC (bit mask) selects how much arithmetic the conversion gets.  The reference does all of this in float64
// (skeleton.py:44); what an all-fp32 evaluation loses is dominated by ONE term: q^ = q * rcp(|q| + eps) misses unit
// length by ~1.5 ulp, and the matrix is a quadratic form of q^, so (L - I) is off by twice that relative error
// (up to 1.3e-6 absolute).  Down the chain that error is multiplied by the bone offsets: harmless at metre scale,
// 2e-4 at centimetre scale (offsets ~30, |pos| ~400).
//   PREC_FAST    the plain fp32 evaluation (round 1)
//   PREC_RESID   fp32, but the matrix is scaled by 2 / |q^|^2 with |q^|^2 - 1 taken from an FMA chain (the residual
//                of the normalisation, accurate to a few 1e-8): + 9 instructions, L error 1.3e-6 -> 5e-7
//   PREC_F64     normalisation and quadratic form in float64 (products of fp32 inputs are exact there), one rounding
//                per matrix entry: L error 3e-8
//   PREC_FX      (walks) the translation chain p_j = p_parent + R_parent t_j runs in 32-bit FIXED POINT: integer adds do
//                not round, so the only roundings are one per joint of the fp32 dot product (an ulp of |t|, not of |p|)
//                and the final conversion.  The image holds the fixed-point words during the walk (a child that re-reads
//                its parent resumes the exact chain, no extra LDS) and is converted in place before the copy-out.
//   PREC_DYN     what the production library runs: every tile decides for itself (fk_tile_is_big) -- PREC_RESID for
//                human-scale data in metres, PREC_F64 | PREC_FX when bones or root positions are big enough for fp32
//                roundings of |p| to matter (centimetre mocap, far-away roots).
//   PREC_BIG_RESID (with PREC_DYN; shallow skeletons on the three-lane kernels, chosen in dispatch_fk): big tiles keep the
//                residual-scaled fp32 rotations and take only the fixed-point chain.  What the float64 rotations buy is their error
//                times the bone lengths summed down the chain; at depth <= 7 the fixed-point chain alone holds the 2-ulp bar
//                (measured on centimetre data, J = 22: RESID + FX 1.43 ulp against 0.96 with float64 rotations, 267 us against 283;
//                the 52-joint tree, depth 10, reads 2.7 ulp that way and keeps both).
// Measured at 2^20 x 22 / 2^18 x 52 (sustained, us) and, on offsets +-30 / root +-200, max |pos error| in ulps of the
// largest coordinate: FAST 267 / 175, 4.3 / 6.3 ulp; RESID 266 / 172, 2.2 / 2.9; F64 273 / 179, 2.0 / 3.0; a float64
// chain with its residuals in LDS 310-320 / 252, 0.9 / 1.4 -- but 3/4 of that cost is the LDS (occupancy), not the math.
// (enum PREC_* and local_from_quat: common.hpp)

// ---- phase B: row-parallel tree walk over one LDS tile ----------------------------------------------
// sRot slot (f, j) holds the local rotation L_j on entry and row-by-row the world rotation on exit;
// sPos receives the positions.  sConst[j] = {parent (int bits), t0, t1, t2}, entry J = clamp copy.
// Lane (f, r) owns row r of frame f; `gp` = root_pos[f][r].
// FX: positions in fixed point (see PREC_FX): sPos holds int32 words p * S while the walk runs.
// `poison` (wave-uniform; float walks only): a translation of the tile is NaN / Inf.  The reference multiplies homogeneous 4 x 4 matrices
// (skeleton.py:54-57), so row r of a joint's ROTATION carries the term p_parent[r] * 0 -- NaN as soon as the parent's position is not
// finite (a NaN root coordinate turns row r of every other joint's matrix into NaN).  The walks add that term when, and only when, a tile
// has such a translation: one scalar branch per joint otherwise.
template <bool PFO, bool FX>
__device__ __forceinline__ void tree_walk(float *sRot, float *sPos, const float *sOff, const float *sConst,
                                          const int J, const int pad, const int f, const int r, const float gp, const bool skip,
                                          const float S, const bool poison) {
    float *fL = sRot + f * (J * 9 + pad);  // this frame's slots (L before, G after)
    float *fRot = fL + r * 3;              // this lane's row inside a slot
    float *fPos = sPos + f * (J * 3 + pad) + r;
    const float *fOff = sOff + f * (J * 3 + pad);

    // Row r of joint j-1's transform, seeded so that joint 0 falls out of the same formula:
    // e_r . L = row r of L (exact: 1*x + 0*y + 0*z) and translation = root_pos[r] (offsets[0] ignored).
    float g0 = (r == 0) ? 1.0f : 0.0f, g1 = (r == 1) ? 1.0f : 0.0f, g2 = (r == 2) ? 1.0f : 0.0f;
    float gt = FX ? __int_as_float((int)__builtin_rintf(gp * S)) : gp;

    // One joint of the walk.  `L` and the joint's constants `c` were fetched a joint ahead.
    auto joint = [&](const int j, const float (&L)[9], const v4f c) {
        const int par = __builtin_amdgcn_readfirstlane(__float_as_int(c.x));
        float t0 = c.y, t1 = c.z, t2 = c.w;
        float p0 = g0, p1 = g1, p2 = g2, pt = gt;
        if (par != j - 1) {  // wave-uniform: not the previous joint -> its row is in the image
            p0 = fRot[par * 9]; p1 = fRot[par * 9 + 1]; p2 = fRot[par * 9 + 2];
            pt = fPos[par * 3];
        }
        if (PFO && j > 0) { t0 = fOff[3 * j]; t1 = fOff[3 * j + 1]; t2 = fOff[3 * j + 2]; }
        // FMAs spelled out: the same frame must round the same way in every instance of the kernel (tile sizes, tails)
        g0 = __builtin_fmaf(p2, L[6], __builtin_fmaf(p1, L[3], p0 * L[0]));
        g1 = __builtin_fmaf(p2, L[7], __builtin_fmaf(p1, L[4], p0 * L[1]));
        g2 = __builtin_fmaf(p2, L[8], __builtin_fmaf(p1, L[5], p0 * L[2]));
        const float dt = __builtin_fmaf(p2, t2, __builtin_fmaf(p1, t1, p0 * t0));
        if (FX) {
            gt = __int_as_float(__float_as_int(pt) + (int)__builtin_rintf(dt * S));
        } else {
            gt = dt + pt;
            if (poison && j > 0) poison_row(pt, g0, g1, g2);
        }
        // all three lanes of the frame have read slot j (in-order DS) -> overwrite in place
        fRot[j * 9] = g0; fRot[j * 9 + 1] = g1; fRot[j * 9 + 2] = g2;
        fPos[j * 3] = gt;
    };

    // Two joints per trip with ping-pong register sets, so the look-ahead costs no moves.  Slot j+1
    // still holds L_{j+1} while joint j is processed (it is only overwritten at step j+1); the slot
    // read past the last joint lies inside the LDS allocation (sPos follows sRot).
    const v4f *cst = reinterpret_cast<const v4f *>(sConst);
    float La[9], Lb[9];
    v4f ca, cb;
    lds_get<9>(fL, 0, La);
    ca = cst[0];
    for (int j = skip ? J : 0; j < J; j += 2) {
        lds_get<9>(fL, j + 1, Lb);
        cb = cst[j + 1];
        joint(j, La, ca);
        if (j + 1 >= J) break;
        lds_get<9>(fL, j + 2, La);
        ca = cst[j + 2 <= J ? j + 2 : J];
        joint(j + 1, Lb, cb);
    }
}

// ---- phase B, sixteen frames: the row walk with L shared across a quad (round 4) --------------------------------------------
// tree_walk's three lanes of a frame each read all nine entries of L_j (six LDS instructions per joint and wave), and the walks of the nine
// waves of a CU keep its LDS pipe about a third busy: prefetching the parent's row as well (three more reads per joint, every wait a counted
// one) made the kernel 8 % SLOWER (273 against 254 us) -- the walk is bound by LDS instructions, not by their latency.  With 16 frames per
// wave a frame can own a QUAD (lane 4 f + r, lane 3 idle): lane r reads only ROW r of L_j -- one third of the LDS traffic -- and the other
// two rows reach it through the DPP operand of the multiply-adds: g_c = sum_k p_k L[k][c] with L[k][c] a quad broadcast from lane k.  The
// operand that crosses lanes was loaded from LDS, not computed, so there is no VALU -> DPP wait state, and the dependency chain runs through
// the lane's own accumulators.
template <bool FX>
__device__ __forceinline__ void tree_walk_q4(float *sRot, float *sPos, const float *sConst, const int J, const int pad, const int f, const int r,
                                             const float gp, const bool skip, const float S, const bool poison) {
    float *fL = sRot + f * (J * 9 + pad);
    float *fRot = fL + r * 3;  // this lane's row inside a slot: L[r][0..2] before the step, G[r][0..2] after
    float *fPos = sPos + f * (J * 3 + pad) + r;
    float g0 = (r == 0) ? 1.0f : 0.0f, g1 = (r == 1) ? 1.0f : 0.0f, g2 = (r == 2) ? 1.0f : 0.0f;
    float gt = FX ? __int_as_float((int)__builtin_rintf(gp * S)) : gp;
    // out = p0 * bcast_0(l) + p1 * bcast_1(l) + p2 * bcast_2(l), l = this lane's element c of its row of L (lane k of the quad: L[k][c])
    auto dot_bcast = [](const float l, const float p0, const float p1, const float p2) __attribute__((always_inline)) {
        float acc;
        asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
            : "=&v"(acc)
            : "v"(l), "v"(p0), "v"(p1), "v"(p2));
        return acc;
    };
    auto joint = [&](const int j, const float (&Lr)[3], const v4f c) __attribute__((always_inline)) {
        const int par = __builtin_amdgcn_readfirstlane(__float_as_int(c.x));
        float p0 = g0, p1 = g1, p2 = g2, pt = gt;
        if (par != j - 1) {  // wave-uniform: not the previous joint -> its row is in the image
            p0 = fRot[par * 9]; p1 = fRot[par * 9 + 1]; p2 = fRot[par * 9 + 2];
            pt = fPos[par * 3];
        }
        g0 = dot_bcast(Lr[0], p0, p1, p2);
        g1 = dot_bcast(Lr[1], p0, p1, p2);
        g2 = dot_bcast(Lr[2], p0, p1, p2);
        const float dt = __builtin_fmaf(p2, c.w, __builtin_fmaf(p1, c.z, p0 * c.y));
        if (FX) gt = __int_as_float(__float_as_int(pt) + (int)__builtin_rintf(dt * S));
        else {
            gt = dt + pt;
            if (poison && j > 0) poison_row(pt, g0, g1, g2);
        }
        fRot[j * 9] = g0; fRot[j * 9 + 1] = g1; fRot[j * 9 + 2] = g2;
        fPos[j * 3] = gt;
    };
    const v4f *cst = reinterpret_cast<const v4f *>(sConst);
    float La[3], Lb[3];
    v4f ca, cb;
    La[0] = fRot[0]; La[1] = fRot[1]; La[2] = fRot[2];
    ca = cst[0];
    for (int j = skip ? J : 0; j < J; j += 2) {
        Lb[0] = fRot[(j + 1) * 9]; Lb[1] = fRot[(j + 1) * 9 + 1]; Lb[2] = fRot[(j + 1) * 9 + 2];
        cb = cst[j + 1];
        joint(j, La, ca);
        if (j + 1 >= J) break;
        La[0] = fRot[(j + 2) * 9]; La[1] = fRot[(j + 2) * 9 + 1]; La[2] = fRot[(j + 2) * 9 + 2];
        ca = cst[j + 2 <= J ? j + 2 : J];
        joint(j + 1, Lb, cb);
    }
}

// ---- phase B, wide form: TWELVE lanes per frame (row r x column c of [R | p]) -------------------------
// For big skeletons the LDS image (48 J B per frame) leaves room for few frames per CU, and with three
// lanes per frame a wave needs 8-20 frames to be worth its instructions.  Here a QUAD (4 consecutive
// lanes) owns row r of a frame and each lane ONE element of it (c < 3: R[r][c], c = 3: p[r]):
//     G[r][c] = sum_k Gp[r][k] * [L | t][k][c]  (+ Gp[r][3] for c = 3)
// The parent row lives in the quad's own registers (previous joint) and is broadcast by the DPP operand
// of the multiply itself, so a chain step is 4 VALU instructions per lane and never goes through LDS;
// a wave walks only FPW <= 5 frames (10 KiB of LDS at J = 52) and many more waves are resident.
// These kernels sit close to the VALU issue limit (SQ_ACTIVE_INST_VALU covers ~2/3 of the SIMD cycles), so
// the step is kept to ~10 instructions:
//   * phase A leaves L TRANSPOSED in the slot, so the three coefficients of a lane (column c of L, or
//     the offset t for the position lane) are contiguous: one pointer, immediate offsets;
//   * the parent indices ride in a VGPR across the lanes (lane i = joint base + i), one v_readlane per
//     step instead of an LDS read + readfirstlane behind an lgkmcnt wait.

// sum_k bcast_k(e) * a_k over the quad (k = 0..2), DPP folded into the multiplies.
__device__ __forceinline__ float quad_dot3(const float e, const float a0, const float a1, const float a2) {
    float acc;
    // s_nop 1: a VGPR written by VALU needs two wait states before a DPP read (the assembler does not
    // pad inside inline asm)
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(acc)
        : "v"(e), "v"(a0), "v"(a1), "v"(a2));
    return acc;
}

// (POISON is a template parameter here: the twelve-lane kernels are bound by instruction issue, and one scalar branch per step was 2-3 points
// at J = 52 -- same-box A/B, profiles/r05_poison_ab.txt)
template <bool PFO, bool FX, bool POISON = false>
__device__ __forceinline__ void tree_walk_quad(float *sRot, float *sPos, const float *sOff, const float *sConst,
                                               const int J, const int pad, const int f, const int r, const int c, const float seed_in,
                                               const int lane, const float S) {
    float *fL = sRot + f * (J * 9 + pad);
    // what this lane multiplies the parent row with at joint j: column c of L_j = row c of the transposed
    // slot, or, for the position lane, the offset t_j (constant table, or the per-frame offsets tile)
    const float *coef = (c < 3) ? (fL + 3 * c) : (PFO ? sOff + f * (J * 3 + pad) : sConst + 1);
    const int cstep = (c < 3) ? 9 : (PFO ? 3 : 4);
    // where this lane's element of joint j lives in the image (row-major G, positions)
    float *own0 = (c < 3) ? (fL + r * 3 + c) : (sPos + f * (J * 3 + pad) + r);
    const int ostep = (c < 3) ? 9 : 3;
    const float m3 = (c == 3) ? 1.0f : 0.0f;

    // FX (see PREC_FX): the position lane's element is a fixed-point word p * S (integer adds: no rounding of |p|); the
    // rotation lanes run the same instructions and keep the float.  A lane's `g` is what the next step's DPP operand
    // reads from lanes 0..2 of the quad (rotation elements) -- the position lane's bits are never multiplied.
    const float seed = (FX && c == 3) ? __int_as_float((int)__builtin_rintf(seed_in * S)) : seed_in;
    float g = seed;  // element (r, c) of joint j-1: joint 0 multiplies the seed row e_r | root_pos[r] (exact)
    float *own = own0;
    int par = -1;
    // One step, straight-line (no branch, so every LDS wait is a counted one):
    //   * `a` = this joint's coefficients, requested two steps ago; once used the same registers are
    //     refilled with joint j+2's (slots j+1, j+2 still hold L^T; the image and the table have two
    //     entries of slack past the last joint);
    //   * `pe` = the parent's element read from the image one step ago -- used when the parent is not
    //     joint j-1 (then it was finished, and written, before step j-1); otherwise the register chain;
    //   * first thing, the same read is issued for joint j+1 (`pen`).
    auto step = [&](const int j, const int parn, float (&a)[3], const float pe, float &pen, const bool last_may_be_dummy) {
        pen = own0[__umul24(parn, ostep)];
        const float e = (par == j - 1) ? g : pe;  // wave-uniform
        if (FX) {
            const float dot = quad_dot3(e, a[0], a[1], a[2]);
            const int gi = __float_as_int(e) + (int)__builtin_rintf(dot * S);
            g = (c == 3) ? __int_as_float(gi) : dot;
        } else {
            g = __builtin_fmaf(m3, e, quad_dot3(e, a[0], a[1], a[2]));  // + Gp[r][3] on the position lane
            if (POISON && j > 0) {  // + Gp[r][3] * 0 on the rotation lanes: the position lane's parent element, across the quad
                const float pp = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0xff, 0xf, 0xf, true));  // quad_perm:[3,3,3,3]
                g = (c < 3) ? __builtin_fmaf(pp, 0.0f, g) : g;
            }
        }
        if (!last_may_be_dummy || j < J) *own = g;
        own += ostep;
        a[0] = coef[2 * cstep]; a[1] = coef[2 * cstep + 1]; a[2] = coef[2 * cstep + 2];
        coef += cstep;
        par = parn;
    };
    float A[3] = {coef[0], coef[1], coef[2]}, B[3] = {coef[cstep], coef[cstep + 1], coef[cstep + 2]};
    if (c == 3) { A[0] = 0.0f; A[1] = 0.0f; A[2] = 0.0f; }  // root: translation = the seed itself (offsets[0] ignored)
    float peA = 0.0f, peB = 0.0f;
    for (int jb = 0; jb < J; jb += PM_WAVE) {
        // parents of joints jb+1 .. jb+64 across the lanes (the table repeats its last entry past J)
        const int i0 = jb + 1 + lane;
        const int pv = __float_as_int(sConst[4 * (i0 < J ? i0 : J)]);
        const int jend = (J - jb) < PM_WAVE ? (J - jb) : PM_WAVE;
        asm volatile("" ::"v"(pv));  // settle the window load here, not as an lgkmcnt(0) inside the loop
        for (int jj = 0; jj < jend; jj += 2) {  // pairs; for odd J the very last step is a dummy that stores nothing
            step(jb + jj, __builtin_amdgcn_readlane(pv, jj), A, peA, peB, false);
            step(jb + jj + 1, __builtin_amdgcn_readlane(pv, jj + 1), B, peB, peA, true);
        }
    }
}

// ---- phase B, four frames a wave, FOUR JOINTS of a frame at a time (round 5) ---------------------------------------------------------------
// tree_walk_quad visits a frame's J joints one after the other with twelve lanes, ~15 instructions a step for four frames: 3.75 a joint-frame,
// the biggest item of kernels that are bound by instruction issue (J = 52: ~5.2 a joint-frame at 64 % of the HBM spec).  But a humanoid is not a
// chain: SMPL-H's 52 joints are 10 levels deep, and list-scheduled four at a time (fk_wide_plan: a joint at the earliest one step after its
// parent, the longest path below first) they take 15 steps, not 52.  Here a QUAD owns one joint of a step -- sixteen quads: four frames x four
// slots -- and works like fk_wide_kernel's (fkwide.hip): lane r row r of [R | p], the joint's [L | t] shared across the quad through the DPP
// operand of the multiply-adds, the parent's row and the next step's [L | t] row read from the image, the step words in registers (lane t of a
// quad holds the word of step 4 g + t, a quad broadcast hands it out).  ~40 instructions a step for sixteen joint-frames: 2.5 a joint-frame
// with a full list, and SMPL-H's is 85 % full.  The image holds L row-major here (tree_walk_quad wants it transposed), the root takes no step
// (its slot holds L_0 = R_0; the caller parks its position), and a slot without a joint repeats the step's first joint (same reads, same writes).
// Same products in the same order as every other walk: the results are theirs to the bit.
template <bool PFO, bool FX>
__device__ __forceinline__ void tree_walk_w4(float *sRot, float *sPos, const float *sOff, const float *sConst, const int J, const int pad,
                                             const int lane, const uint32_t (&JW)[kW4Groups + 1], const int nsteps, const float S, const bool poison) {
    const int q = lane >> 2, f = q >> 2, r = (lane & 3) < 3 ? (lane & 3) : 2;  // (lane 3 of a quad shadows lane 2: it holds a quarter of the quad's words)
    float *fL = sRot + f * (J * 9 + pad) + r * 3, *fP = sPos + f * (J * 3 + pad) + r;
    const float *fT = PFO ? sOff + f * (J * 3 + pad) + r : sConst + 1 + r;  // t_j[r]: the per-frame offsets tile, or the joint table {parent, t0, t1, t2}
    constexpr unsigned TS = PFO ? 3u : 4u;
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
    auto word = [](const uint32_t v, auto t) __attribute__((always_inline)) {
        constexpr int T = decltype(t)::value;
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, T * 0x55, 0xf, 0xf, true);  // quad_perm:[T,T,T,T]
    };
    uint32_t w = word(JW[0], IntC<0>{});
    float *aj = fL + __umul24(w & 0xffffu, 9u), *pj = fP + __umul24(w & 0xffffu, 3u);
    float l0 = aj[0], l1 = aj[1], l2 = aj[2], tr = fT[__umul24(w & 0xffffu, TS)];
    auto step = [&](const uint32_t wn) __attribute__((always_inline)) {
        const unsigned par = w >> 16, jn = wn & 0xffffu;
        const float *ap = fL + __umul24(par, 9u);
        const float p0 = ap[0], p1 = ap[1], p2 = ap[2], pt = fP[__umul24(par, 3u)];
        float *an = fL + __umul24(jn, 9u), *pn = fP + __umul24(jn, 3u);  // the next step's [L | t] row: its slot is written by that step only
        const float n0 = an[0], n1 = an[1], n2 = an[2], nt = fT[__umul24(jn, TS)];
        float g0, g1, g2, dt;
        dot4(l0, l1, l2, tr, p0, p1, p2, g0, g1, g2, dt);
        float gt;
        if (FX) gt = __int_as_float(__float_as_int(pt) + (int)__builtin_rintf(dt * S));
        else {
            gt = dt + pt;
            if (poison) poison_row(pt, g0, g1, g2);
        }
        aj[0] = g0; aj[1] = g1; aj[2] = g2;
        *pj = gt;
        aj = an; pj = pn;
        l0 = n0; l1 = n1; l2 = n2; tr = nt;
        w = wn;
    };
    // (a finished joint's slot holds R, not L: exactly nsteps steps, no idle ones.  The group counter goes through readfirstlane: left to
    // itself the compiler kept it in a vector register and wrapped every indexed read of JW in a waterfall loop)
    const int ng = (nsteps + 3) >> 2;
#pragma nounroll
    for (int g = 0; g < ng; ++g) {
        const int gu = __builtin_amdgcn_readfirstlane(g), left = nsteps - 4 * gu;
        const uint32_t cur = JW[gu], nxt = JW[gu + 1];
        step(word(cur, IntC<1>{}));
        if (left > 1) {
            step(word(cur, IntC<2>{}));
            if (left > 2) {
                step(word(cur, IntC<3>{}));
                if (left > 3) step(word(nxt, IntC<0>{}));
            }
        }
    }
}

// {parent (int bits), t0, t1, t2} of joint j, clamped to the last joint; joint 0's "parent" is the seed row
template <bool PFO>
__device__ __forceinline__ v4f load_joint_const(const Parents &parents, const float *offsets, const int J, const int j) {
    const int jc = j < J ? j : J - 1;
    v4f c;
    c.x = __int_as_float(j == 0 ? -1 : parents.p[jc]);
    const bool none = PFO || j == 0;  // per-frame offsets come from their own tile; offsets[0] is ignored (skeleton.py:49)
    c.y = none ? 0.0f : offsets[3 * jc];
    c.z = none ? 0.0f : offsets[3 * jc + 1];
    c.w = none ? 0.0f : offsets[3 * jc + 2];
    return c;
}

// ---- PREC_DYN: which arithmetic a tile gets (kBigOffset / kBigRoot, FxScale, fx_scale: common.hpp) ------------
// fixed-point words of the position region -> fp32, in place (the region is then the output tile); n4 dwordx4
__device__ __forceinline__ void fx_to_float(float *sPos, const int n4, const float invS, const int lane) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    for (int i = lane; i < n4; i += PM_WAVE) {
        const v4i w = reinterpret_cast<const v4i *>(sPos)[i];
        reinterpret_cast<v4f *>(sPos)[i] = v4f{(float)w.x * invS, (float)w.y * invS, (float)w.z * invS, (float)w.w * invS};
    }
}

// where element e = (frame f, joint j) of a tile starts in a per-frame image region: `w` floats per joint,
// frames J * w + pad floats apart.  pad == 0: the region is linear (e * w), no division.
template <bool PAD>
__device__ __forceinline__ int image_slot(const int e, const int J, const float invJ, const int w, const int pad) {
    if (!PAD) return e * w;  // compile time: the linear image must not carry a branch between the records of a batch
    const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
    return e * w + f * pad;
}

// LDS image region (nf frames of `per_frame` floats, `pad` floats between frames) <-> its contiguous HBM tile.
template <bool VEC>
__device__ __forceinline__ void image_store(float *__restrict__ g, const float *lds, const int nf, const int per_frame,
                                            const int pad, const int lane) {
    if (pad == 0) { tile_store<VEC>(g, lds, nf * per_frame, lane); return; }
    if (VEC) {  // per_frame and pad are multiples of 4 floats whenever pad != 0
        const int V4 = per_frame >> 2, S4 = (per_frame + pad) >> 2, n4 = nf * V4;
        const float inv = 1.0f / (float)V4;
        for (int i = lane; i < n4; i += PM_WAVE) {
            const int f = (int)(((float)i + 0.5f) * inv);
            __builtin_nontemporal_store(reinterpret_cast<const v4f *>(lds)[i + f * (S4 - V4)], reinterpret_cast<v4f *>(g) + i);
        }
    } else {
        const float inv = 1.0f / (float)per_frame;
        for (int k = lane; k < nf * per_frame; k += PM_WAVE) g[k] = lds[k + (int)(((float)k + 0.5f) * inv) * pad];
    }
}
template <bool VEC>
__device__ __forceinline__ void image_load(const float *__restrict__ g, float *lds, const int nf, const int per_frame,
                                           const int pad, const int lane) {
    if (pad == 0) { tile_load<VEC>(g, lds, nf * per_frame, lane); return; }
    if (VEC) {
        const int V4 = per_frame >> 2, S4 = (per_frame + pad) >> 2, n4 = nf * V4;
        const float inv = 1.0f / (float)V4;
        for (int i = lane; i < n4; i += PM_WAVE) {
            const int f = (int)(((float)i + 0.5f) * inv);
            reinterpret_cast<v4f *>(lds)[i + f * (S4 - V4)] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(g) + i);
        }
    } else {
        const float inv = 1.0f / (float)per_frame;
        for (int k = lane; k < nf * per_frame; k += PM_WAVE) lds[k + (int)(((float)k + 0.5f) * inv) * pad] = g[k];
    }
}

// max |x| over a per-frame offsets tile in LDS (pads hold stale finite-or-not words of earlier tiles: skipped)
__device__ __forceinline__ float tile_abs_max(const float *sOff, const int nf, const int per_frame, const int pad, const int lane) {
    float m = 0.0f;
    for (int fr = 0; fr < nf; ++fr)
        for (int k = lane; k < per_frame; k += PM_WAVE) {
            const float v = fabsf(sOff[fr * (per_frame + pad) + k]);
            m = (v > m || v != v) ? v : m;  // NaN sticks
        }
    return m;
}

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT, bool PAD, int PREC>
__device__ __forceinline__ void fk_tile(const FkArgs &a, float *smem, const int64_t f0, const int nf, const int lane) {
    const int J = a.J;
    const int FJ = FPW * J;
    const int n = nf * J;  // (frame, joint) elements in this tile
    constexpr bool QUAD = FPW <= 5;  // few frames per wave (big J): 12 lanes per frame, see tree_walk_quad
    // a quad per frame, the joint's L shared through DPP (tree_walk_q4): sixteen frames fill the wave; with twelve / eight the last quads sit
    // the walk out (round 5: the walk is ~18 instructions per joint this way against ~42 for the three-lane one)
    constexpr bool Q4 = (FPW == 16 || FPW == 12 || FPW == 8) && !PFO;
    const bool q4 = Q4 && !PM_ABLATED(a, 8);  // PM_FK_ABLATE & 8 (tuning build): the three-lane walk
    constexpr bool DYN = (PREC & PREC_DYN) != 0;

    const int pad = PAD ? a.pad : 0;                   // floats between frames in the per-frame regions (see FkArgs::pad)
    const float invJ = 1.0f / (float)J;
    float *sRot = smem;                                // [FPW*(J*9+pad)]  FPW % 4 == 0 keeps every carve 16 B aligned
    float *sPos = sRot + FJ * 9 + FPW * pad;           // [FPW*(J*3+pad)]
    float *sOff = sPos + FJ * 3 + FPW * pad;           // [FPW*(J*3+pad)]  (PFO)
    float *sQo = sOff + (PFO ? FJ * 3 + FPW * pad : 0);  // [FPW*J*4]  (QOUT; lane-per-record access only: linear)
    float *sConst = sQo + (QOUT ? FJ * 4 : 0);         // [(J+4)*4]  per joint {parent (int bits), t0, t1, t2}

    // Every global load of the tile is issued up front, back to back, so the wave pays ONE memory
    // latency: root position (used first by phase B), skeleton constants, then the rotations.
    // Lanes >= 3*FPW shadow lanes 0.. (same frame, same row, same values, same addresses) and frames
    // past the end of a partial tile walk uninitialised slots of their own: phase B needs no masking.
    const int wl = q4 ? lane : lane % ((QUAD ? 12 : 3) * FPW);
    // (q4, sixteen frames: quad q walks frame fmap[q] -- see q4_frame_map; fewer: the quads past the tile shadow its last frame and sit the walk out)
    const int f = q4 ? (FPW == 16 ? (int)((a.fmap >> (4 * (lane >> 2))) & 15u) : ((lane >> 2) < FPW ? (lane >> 2) : FPW - 1)) : (QUAD ? wl / 12 : wl / 3);
    const int r = q4 ? ((lane & 3) < 3 ? (lane & 3) : 2) : (QUAD ? (wl - 12 * f) / 4 : wl - 3 * f);  // (q4: lane 3 of a quad shadows lane 2's loads and sits the walk out)
    const int c = wl & 3;  // QUAD only: column of [R | p]
    const float gp = (f < nf) ? a.root_pos[f0 * 3 + f * 3 + r] : 0.0f;

    // Skeleton constants -> LDS once per tile, {parent (int bits), t0, t1, t2} per joint.  (Scalar loads
    // inside the walk would share lgkmcnt with the DS traffic and, returning out of order, force full
    // lgkmcnt(0) drains on every joint.)  Entry J is a clamp copy for the walk's look-ahead.
    auto load_const = [&](const int j) { return load_joint_const<PFO>(a.parents, a.offsets, J, j); };
    const v4f c_first = load_const(lane <= J ? lane : J);  // joints 0..63 (all of them for J < 64)

    // input records of the first two batches, requested here, consumed by phase A (see `rest`)
    constexpr int B = (SRC == SRC_QUAT ? 4 : 2) * PM_WAVE;
    float qa[4][4], qb[4][4];   // SRC_QUAT: 4 quaternions per lane and batch
    v2f xa[2][3], xb[2][3];     // SRC_O6D: 2 records of three dwordx2 (24-byte records are 8-byte aligned: no LDS
                                // staging of the input, which at J = 52 is what buys a third resident wave per CU)
    const float *gsrc = a.src + f0 * J * (SRC == SRC_QUAT ? 4 : 6);
    // Loads and math are unconditional inside a batch (a record past the tile's end re-reads the last one): no
    // exec-mask branches between the records, so the compiler schedules and packs them together; only the LDS
    // write is guarded.
    auto load_q = [&](const int e0, float (&qi)[4][4]) {
        if (e0 >= n) return;  // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
            if (VEC) {
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gsrc) + ec);
                qi[u][0] = t.x; qi[u][1] = t.y; qi[u][2] = t.z; qi[u][3] = t.w;
            } else {
                qi[u][0] = gsrc[4 * ec]; qi[u][1] = gsrc[4 * ec + 1]; qi[u][2] = gsrc[4 * ec + 2]; qi[u][3] = gsrc[4 * ec + 3];
            }
        }
    };
    auto load_x = [&](const int e0, v2f (&x)[2][3]) {
        if (e0 >= n) return;  // wave-uniform
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
            if (VEC) {
                const v2f *p = reinterpret_cast<const v2f *>(gsrc) + 3 * ec;
                x[u][0] = __builtin_nontemporal_load(p);
                x[u][1] = __builtin_nontemporal_load(p + 1);
                x[u][2] = __builtin_nontemporal_load(p + 2);
            } else {
                const float *p = gsrc + 6 * ec;
                x[u][0] = v2f{p[0], p[1]}; x[u][1] = v2f{p[2], p[3]}; x[u][2] = v2f{p[4], p[5]};
            }
        }
    };
    // two batches (8 x 16 B per lane = 8 KiB per wave) in flight before the first use, then ping-pong: batch k+2
    // is requested before batch k is consumed
    if constexpr (SRC == SRC_QUAT) { load_q(0, qa); load_q(B, qb); } else { load_x(0, xa); load_x(B, xb); }
    // per-frame offsets: requested while the rotations are still in flight (one memory latency for both)
    if (PFO) image_load<VEC>(a.offsets + f0 * J * 3, sOff, nf, J * 3, pad, lane);
    if (lane <= J) reinterpret_cast<v4f *>(sConst)[lane] = c_first;
    bool tbig = lane < J && const_is_big(c_first);
    float tsum = lane < J ? const_l1(c_first) : 0.0f, tmx = tsum;
    for (int j = lane + PM_WAVE; j <= J; j += PM_WAVE) {
        const v4f cj = load_const(j);
        reinterpret_cast<v4f *>(sConst)[j] = cj;
        if (j < J) { const float l1 = const_l1(cj); tbig = tbig || const_is_big(cj); tsum += l1; tmx = (l1 > tmx || l1 != l1) ? l1 : tmx; }
    }

    // ---- PREC_DYN: does this tile need the float64 rotations and the fixed-point chain? ------------------------
    bool big = false, poison = false;
    FxScale fx = {1.0f, 1.0f};
    if constexpr (DYN || (PREC & PREC_FX)) {
        bool mine = tbig || !(fabsf(gp) < kBigRoot);
        float tmax = 0.0f;
        if (PFO) {
            wave_sync();
            tmax = tile_abs_max(sOff, nf, J * 3, pad, lane);
            mine = mine || !(tmax < kBigOffset);
        }
        big = __builtin_amdgcn_ballot_w64(mine) != 0 || !DYN;
        if (big) {  // wave-uniform; a non-finite bound (NaN / Inf inputs) keeps the plain path, which propagates them
            const float bsum = wave_sum(tsum), bmax = (float)a.depth * wave_max(tmx);  // (NaN sticks in both)
            const float tbound = PFO ? 3.0f * (float)a.depth * wave_max(tmax) : ((bmax < bsum) ? bmax : bsum);
            big = fx_scale(tbound, fabsf(gp), fx);
            poison = !big;  // a translation of the tile is NaN / Inf (or absurd): the float walk carries the reference's p_parent[r] * 0 (see tree_walk)
        }
    }

    // ---- phase A, phase B and the copy-out for one arithmetic level M ------------------------------------------
    auto rest = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
        constexpr bool FX = (M & PREC_FX) != 0;
        bool bad = false;  // FX only: a non-finite local rotation somewhere in the tile
        if constexpr (SRC == SRC_QUAT) {
            auto do_batch = [&](const int e0, const float (&qi)[4][4]) {
                if (e0 >= n) return;  // wave-uniform
                float L[4][9];
#pragma unroll
                for (int u = 0; u < 4; ++u) local_from_quat<M>(qi[u], L[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * PM_WAVE + lane;
                    if (FX) bad = bad || !(fabsf(qi[u][0]) + fabsf(qi[u][1]) + fabsf(qi[u][2]) + fabsf(qi[u][3]) < 3e38f);  // NaN / Inf input
                    if (e < n) put_local<QUAD>(sRot + image_slot<PAD>(e, J, invJ, 9, pad), L[u]);
                }
            };
            for (int e0 = 0; e0 < n; e0 += 2 * B) {
                do_batch(e0, qa);
                load_q(e0 + 2 * B, qa);
                do_batch(e0 + B, qb);
                load_q(e0 + 3 * B, qb);
            }
        } else {
            // rotations/ortho6d.py:50-64 : 6D -> matrix -> quaternion (itself normalised), then fk's own
            // normalise and to_matrix: the chain ortho6d.to_quat -> fk of the reference (see the shortcut below).
            auto do_batch = [&](const int e0, const v2f (&x)[2][3]) {
                if (e0 >= n) return;
                float L[2][9], Q[2][4];
                bool ill[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float xx[6] = {x[u][0].x, x[u][0].y, x[u][1].x, x[u][1].y, x[u][2].x, x[u][2].y};
                    ill[u] = local_from_o6d<QOUT, M>(xx, a.eps, L[u], Q[u]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = e0 + u * PM_WAVE + lane;
                    if (FX) bad = bad || !(fabsf(x[u][0].x) + fabsf(x[u][0].y) + fabsf(x[u][1].x) + fabsf(x[u][1].y) + fabsf(x[u][2].x) + fabsf(x[u][2].y) < 3e38f);
                    if (e < n) {
                        put_local<QUAD>(sRot + image_slot<PAD>(e, J, invJ, 9, pad), L[u]);
                        if (QOUT) lds_put<4>(sQo, e, Q[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {  // the rare float64 redo, on the parked slots (in-order DS: after the plain values)
                    const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
                    const float xx[6] = {x[u][0].x, x[u][0].y, x[u][1].x, x[u][1].y, x[u][2].x, x[u][2].y};
                    o6d_redo_ill<QOUT, QUAD>(ill[u] && e < n, xx, a.eps, sRot + image_slot<PAD>(ec, J, invJ, 9, pad), sQo + 4 * ec);
                }
            };
            for (int e0 = 0; e0 < n; e0 += 2 * B) {
                do_batch(e0, xa);
                load_x(e0 + 2 * B, xa);
                do_batch(e0 + B, xb);
                load_x(e0 + 3 * B, xb);
            }
        }
        // ---- phase B ---------------------------------------------------------------------------------------
        wave_sync();
        const bool fixed = FX && __builtin_amdgcn_ballot_w64(bad) == 0;  // NaN / Inf rotations: the float walk propagates them
        if constexpr (QUAD) {
            const float seed = (c == 3) ? gp : ((c == r) ? 1.0f : 0.0f);
            if (!PM_ABLATED(a, 2)) {
                if (FX && fixed) tree_walk_quad<PFO, FX>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, fx.S);
                else if (poison) tree_walk_quad<PFO, false, true>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, 1.0f);
                else tree_walk_quad<PFO, false>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, 1.0f);
            }
        } else if (Q4 && q4) {
            if ((lane & 3) < 3 && (lane >> 2) < FPW) {  // (the DPP operands come from lanes 0..2 of the quad only)
                if (FX && fixed) tree_walk_q4<FX>(sRot, sPos, sConst, J, pad, f, r, gp, PM_ABLATED(a, 2), fx.S, false);
                else tree_walk_q4<false>(sRot, sPos, sConst, J, pad, f, r, gp, PM_ABLATED(a, 2), 1.0f, poison);
            }
        } else {
            if (FX && fixed) tree_walk<PFO, FX>(sRot, sPos, sOff, sConst, J, pad, f, r, gp, PM_ABLATED(a, 2), fx.S, false);
            else tree_walk<PFO, false>(sRot, sPos, sOff, sConst, J, pad, f, r, gp, PM_ABLATED(a, 2), 1.0f, poison);
        }
        wave_sync();
        if (FX && fixed) {  // fixed-point words -> fp32 in place; the root is the caller's value, bit for bit (skeleton.py:49)
            fx_to_float(sPos, (FPW * (J * 3 + pad)) >> 2, fx.invS, lane);
            if (!QUAD || c == 3) sPos[f * (J * 3 + pad) + r] = gp;
            wave_sync();
        }
        image_store<VEC>(a.rotmats + f0 * J * 9, sRot, nf, J * 9, pad, lane);
        image_store<VEC>(a.pos + f0 * J * 3, sPos, nf, J * 3, pad, lane);
        if (QOUT) tile_store<VEC>(a.quat_out + f0 * J * 4, sQo, n * 4, lane);
    };
    if constexpr (DYN) {
        if (big) rest(IntC<((PREC & PREC_BIG_RESID) ? PREC_RESID : PREC_F64) | PREC_FX>{});
        else rest(IntC<PREC & (PREC_RESID | PREC_F64)>{});
    } else {
        if ((PREC & PREC_FX) && !big) rest(IntC<PREC & ~PREC_FX>{});  // static FX (tuning): non-finite bound -> float walk
        else rest(IntC<PREC>{});
    }
}

// One tile (FPW frames) per single-wave workgroup; XCD-aware tile order (common.hpp).
template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT, bool PAD, int PREC>
__global__ __launch_bounds__(PM_WAVE) void fk_kernel(const FkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = PM_ABLATED(a, 4) ? ((int64_t)blockIdx.x < ntiles ? (int64_t)blockIdx.x : -1) : xcd_tile_chunked(ntiles, a.xchunk);  // tuning: linear tile order
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    fk_tile<FPW, VEC, PFO, SRC, QOUT, PAD, PREC>(a, smem, f0, nf, threadIdx.x);
}

// ---- pipelined form for mid-size skeletons (tree_walk_quad shape: FPW = 4, J <= 64, shared offsets) ---
// With ~50 joints the serial walk is a third of a tile's life and a wave that walks has nothing in
// flight; 15 resident waves then cannot keep HBM busy (measured: 53 % of peak for the fused ortho6d
// config against 71 % with the walk ablated).  Here a workgroup owns `nt` consecutive tiles and keeps
// the memory system fed from inside the walk:
//     loads(i+1) are issued BEFORE walk(i) into registers (<= 4 records per lane),
//     after the walk:  math(i+1) from those registers -> copy-out(i) -> park L(i+1) in LDS -> loads(i+2)
// so loads overlap the walk and the stores overlap the next tile's math and walk.  The vmcnt wait in
// front of math(i+1) only ever covers loads that are a whole walk old (the stores of tile i are issued
// after it).  The skeleton table is staged once per workgroup.
// (Second launch bound: at most 128 VGPRs for the four-records-per-lane form, i.e. four waves per SIMD.  Its tiles are small
// -- 5 to 10 KiB of LDS -- so registers, not LDS, bound residency, and the two arithmetic levels of PREC_DYN in one kernel
// had pushed the per-frame-offsets variant to 166 VGPRs: 2^20 x 22 with per-frame offsets 416 us, 369 with the bound.  The ortho6d
// source is left alone: bounded, the variant with the quaternion output spills 40-100 registers (268 -> 409 us), and the one
// without (134 VGPRs once the float64 redo of degenerate records moved behind the parking) gains nothing from a fourth wave.)
// FIXED (J a multiple of 4, 16-byte aligned linear image): the copy-out is a fixed number of unconditional dwordx4 stores, see copy_out.
template <int FPW, int EPL, bool VEC, int SRC, bool QOUT, bool PAD, bool PFO, int PREC, bool FIXED = false>
__global__ __launch_bounds__(PM_WAVE, (EPL <= 4 && SRC == SRC_QUAT) ? ((PFO && EPL == 4) ? 3 : 4) : ((EPL <= 4 && QOUT && !PFO) ? 3 : 1)) void fk_pipe_kernel(const FkArgs a, const int nt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool DYN = (PREC & PREC_DYN) != 0;
    constexpr bool QUAD = FPW <= 5;  // records per lane: FPW * J <= 64 * EPL
    const int lane = threadIdx.x;
    const int J = a.J;
    const int FJ = FPW * J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t ngroups_ = (ntiles + nt - 1) / nt;
    const int64_t group = PM_ABLATED(a, 4) ? ((int64_t)blockIdx.x < ngroups_ ? (int64_t)blockIdx.x : -1) : xcd_tile_chunked(ngroups_, a.xchunk);  // tuning: linear group order
    if (group < 0) return;
    int64_t t0 = group * nt;
    int cnt = (int)((ntiles - t0) < nt ? (ntiles - t0) : nt);
#ifdef PM_TUNING
    const uint64_t wg_start = a.times ? __builtin_amdgcn_s_memrealtime() : 0;
    // PM_FK_ABLATE & 64 (an experiment, two tiles a workgroup): the first `stagger` workgroups of every XCD take one, two and three tiles in turn
    // (triples over the same six tiles) instead of two each -- do 3584 workgroups that load, walk and store in step cost the launch its fixed 18 us?
    if (PM_ABLATED(a, 64) && nt == 2) {
        const int64_t per_xcd = ((ntiles + 1) / 2 + PM_NXCD - 1) / PM_NXCD, i = blockIdx.x / PM_NXCD, k = i / 3;
        const int64_t base = (group - i) * 2;  // the XCD's first tile
        if (i < 447 && 3 * k + 2 < per_xcd && base + 6 * k + 6 <= ntiles) {
            const int m = (int)(i - 3 * k);
            t0 = base + 6 * k + (m == 0 ? 0 : (m == 1 ? 1 : 3));
            cnt = m + 1;
        }
    }
#endif

    const int pad = PAD ? a.pad : 0;             // see FkArgs::pad
    const float invJ = 1.0f / (float)J;
    float *sRot = smem;                          // [FPW*(J*9+pad)]
    float *sPos = sRot + FJ * 9 + FPW * pad;     // [FPW*(J*3+pad)]
    float *sOff = sPos + FJ * 3 + FPW * pad;     // [FPW*(J*3+pad)]  (PFO: per-frame offsets)
    float *sConst = sOff + (PFO ? FJ * 3 + FPW * pad : 0);  // [(J+4)*4]
    // four joints of a frame at a time (tree_walk_w4) when the host's step list says so: this quad's step words, for the life of the workgroup
    const bool wide = QUAD && a.wsteps > 0;
    uint32_t JW[kW4Groups + 1];
#pragma unroll
    for (int g = 0; g <= kW4Groups; ++g) JW[g] = 0u;
    if constexpr (QUAD) {
        if (wide) {
#pragma unroll
            for (int g = 0; g <= kW4Groups; ++g) JW[g] = a.wjobs[((lane >> 2) & 3) * kW4Stride + 4 * g + (lane & 3)];
#pragma unroll
            for (int g = 0; g <= kW4Groups; ++g) asm volatile("" : "+v"(JW[g]));  // settled here: pending, the walk's indexed read would wait for every load in flight
        }
    }
    // (QOUT: the quaternions are one 16-byte record per lane with consecutive lanes on consecutive records -- a contiguous stream
    // as they stand -- and leave straight from the conversion's registers.  Round 2 parked them in a fourth LDS region and copied
    // that out: 16 J B more image per frame and 16 more live registers across the walk, 57.9 % against 61.8 % without the output.)
    // the joint table, and what it says about the arithmetic the tiles need (PREC_DYN, see fk_tile)
    bool tbig_l = false;
    float tsum_l = 0.0f, tmx_l = 0.0f;
    for (int j = lane; j <= J; j += PM_WAVE) {
        const v4f cj = load_joint_const<PFO>(a.parents, a.offsets, J, j);
        reinterpret_cast<v4f *>(sConst)[j] = cj;
        if (j < J) { const float l1 = const_l1(cj); tbig_l = tbig_l || const_is_big(cj); tsum_l += l1; tmx_l = (l1 > tmx_l || l1 != l1) ? l1 : tmx_l; }
    }
    bool tbig = false;   // wave-uniform
    float tsum = 0.0f;   // bound of |p_j - root| from the joint table (shared offsets), see fx_scale
    if constexpr (!PFO && (DYN || (PREC & PREC_FX))) {
        tbig = __builtin_amdgcn_ballot_w64(tbig_l) != 0;
        const float bsum = wave_sum(tsum_l), bmax = (float)a.depth * wave_max(tmx_l);  // (NaN sticks in both)
        tsum = (bmax < bsum) ? bmax : bsum;
    }

    constexpr bool Q4 = (FPW == 16 || FPW == 12 || FPW == 8) && !PFO;  // a quad per frame, L shared through DPP (tree_walk_q4; see fk_tile)
    const int wl = lane % ((QUAD ? 12 : 3) * FPW);
    const int f = Q4 ? (FPW == 16 ? (int)((a.fmap >> (4 * (lane >> 2))) & 15u) : ((lane >> 2) < FPW ? (lane >> 2) : FPW - 1)) : (QUAD ? wl / 12 : wl / 3);
    const int r = Q4 ? ((lane & 3) < 3 ? (lane & 3) : 2) : (QUAD ? (wl - 12 * f) / 4 : wl - 3 * f);
    const int c = wl & 3;  // QUAD only: column of [R | p]

    v4f in4[EPL];       // SRC_QUAT: one quaternion per record
    v2f in2[EPL][3];    // SRC_O6D: 24-byte records as three dwordx2 (see fk_tile)
    v3f_a4 inO[EPL];    // PFO: the record's offset (12-byte records, consecutive lanes on consecutive records)
    float gp = 0.0f;
    auto issue = [&](const int64_t f0, const int nf) {
        const int n = nf * J;
        gp = (f < nf) ? a.root_pos[(f0 + f) * 3 + r] : 0.0f;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = u * PM_WAVE + lane;
            if constexpr (PFO) {
                inO[u] = v3f_a4{0.0f, 0.0f, 0.0f};
                if (e < n) inO[u] = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(a.offsets + (f0 * J + e) * 3));
            }
            if constexpr (SRC == SRC_QUAT) {
                const float *g = a.src + f0 * J * 4;
                in4[u] = v4f{1.0f, 0.0f, 0.0f, 0.0f};
                if (e < n) {
                    if (VEC) in4[u] = PM_ABLATED(a, 512) ? reinterpret_cast<const v4f *>(g)[e] : __builtin_nontemporal_load(reinterpret_cast<const v4f *>(g) + e);  // & 512 (tuning build): plain loads
                    else in4[u] = v4f{g[4 * e], g[4 * e + 1], g[4 * e + 2], g[4 * e + 3]};
                }
            } else {
                const float *g = a.src + f0 * J * 6;
                in2[u][0] = v2f{1.0f, 0.0f}; in2[u][1] = v2f{0.0f, 1.0f}; in2[u][2] = v2f{0.0f, 0.0f};
                if (e < n) {
                    if (VEC) {
                        const v2f *p = reinterpret_cast<const v2f *>(g) + 3 * e;
                        in2[u][0] = __builtin_nontemporal_load(p);
                        in2[u][1] = __builtin_nontemporal_load(p + 1);
                        in2[u][2] = __builtin_nontemporal_load(p + 2);
                    } else {
                        const float *p = g + 6 * e;
                        in2[u][0] = v2f{p[0], p[1]}; in2[u][1] = v2f{p[2], p[3]}; in2[u][2] = v2f{p[4], p[5]};
                    }
                }
            }
        }
    };
    auto copy_out = [&](const int64_t f0, const int nfr) {
        // A FIXED number of unconditional dwordx4 stores (lanes past the tile's end repeat its last vector): behind store loops of
        // unknown length every later wait on a load -- the next tile's records are in flight by then -- is a vmcnt(0), i.e. a wait for
        // these stores to be acknowledged; counted, they drain under the next tile's math and walk.
        if constexpr (FIXED) {
            constexpr int NR = (EPL * 9 + 3) / 4, NP = (EPL * 3 + 3) / 4;
            const int n4r = (nfr * J * 9) >> 2, n4p = (nfr * J * 3) >> 2;
            v4f *gr = reinterpret_cast<v4f *>(a.rotmats + f0 * J * 9), *gp4 = reinterpret_cast<v4f *>(a.pos + f0 * J * 3);
            const v4f *lr = reinterpret_cast<const v4f *>(sRot), *lp = reinterpret_cast<const v4f *>(sPos);
            int ln = lane;
            asm volatile("" : "+v"(ln));
            if (PM_ABLATED(a, 256)) {  // PM_FK_ABLATE & 256 (tuning build): plain instead of nontemporal stores
#pragma unroll
                for (int u = 0; u < NR; ++u) { const int i = u * PM_WAVE + ln, ic = i < n4r ? i : n4r - 1; gr[ic] = lr[ic]; }
#pragma unroll
                for (int u = 0; u < NP; ++u) { const int i = u * PM_WAVE + ln, ic = i < n4p ? i : n4p - 1; gp4[ic] = lp[ic]; }
                return;
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) { const int i = u * PM_WAVE + ln, ic = i < n4r ? i : n4r - 1; __builtin_nontemporal_store(lr[ic], gr + ic); }
#pragma unroll
            for (int u = 0; u < NP; ++u) { const int i = u * PM_WAVE + ln, ic = i < n4p ? i : n4p - 1; __builtin_nontemporal_store(lp[ic], gp4 + ic); }
            return;
        }
        image_store<VEC>(a.rotmats + f0 * J * 9, sRot, nfr, J * 9, pad, lane);
        image_store<VEC>(a.pos + f0 * J * 3, sPos, nfr, J * 3, pad, lane);
    };

    int64_t f0 = t0 * FPW, f0_prev = 0;
    int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW), nf_prev = 0;
    issue(f0, nf);
    for (int i = 0; i < cnt; ++i) {
        const int n = nf * J;
        const float gp_i = gp;
        // PREC_DYN: the arithmetic of THIS tile (see fk_tile): float64 rotations + fixed-point chain when the bones or
        // this tile's root positions are big
        bool big = false, poison = false;
        FxScale fx = {1.0f, 1.0f};
        if constexpr (DYN || (PREC & PREC_FX)) {
            bool mine = !(fabsf(gp_i) < kBigRoot);
            float tmax = 0.0f;
            if constexpr (PFO) {
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const float m = fmaxf(fmaxf(fabsf(inO[u].x), fabsf(inO[u].y)), fabsf(inO[u].z));
                    const bool nan = inO[u].x != inO[u].x || inO[u].y != inO[u].y || inO[u].z != inO[u].z;
                    tmax = nan ? __builtin_nanf("") : ((m > tmax) ? m : tmax);
                }
                mine = mine || !(tmax < kBigOffset);
            }
            big = tbig || __builtin_amdgcn_ballot_w64(mine) != 0 || !DYN;
            if (big) {
                const float bound_t = PFO ? 3.0f * (float)a.depth * wave_max(tmax) : tsum;
                big = fx_scale(bound_t, fabsf(gp_i), fx);
                poison = !big;  // (see fk_tile)
            }
        }
        // math of tile i, in registers (phase A of fk_tile)
        float L[EPL][9];
        float *gq = QOUT ? a.quat_out + f0 * J * 4 : nullptr;  // this tile's quaternion records
        bool bad = false, ill[EPL];
#pragma unroll
        for (int u = 0; u < EPL; ++u) ill[u] = false;
        auto math = [&](auto mode) {
            constexpr int M = decltype(mode)::value;
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                if constexpr (SRC == SRC_QUAT) {
                    const float qi[4] = {in4[u].x, in4[u].y, in4[u].z, in4[u].w};
                    local_from_quat<M>(qi, L[u]);
                } else {
                    const float xx[6] = {in2[u][0].x, in2[u][0].y, in2[u][1].x, in2[u][1].y, in2[u][2].x, in2[u][2].y};
                    float Q[4];
                    ill[u] = local_from_o6d<QOUT, M>(xx, a.eps, L[u], Q);
                    if constexpr (QOUT) {
                        const int e = u * PM_WAVE + lane;
                        if (e < n && !ill[u]) {  // (an ill record's quaternion is stored once, by its float64 redo: no second store to an address)
                            if (VEC) __builtin_nontemporal_store(v4f{Q[0], Q[1], Q[2], Q[3]}, reinterpret_cast<v4f *>(gq) + e);
                            else { gq[4 * e] = Q[0]; gq[4 * e + 1] = Q[1]; gq[4 * e + 2] = Q[2]; gq[4 * e + 3] = Q[3]; }
                        }
                    }
                }
                if (M & PREC_FX) {  // NaN / Inf input record
                    if constexpr (SRC == SRC_QUAT) bad = bad || !(fabsf(in4[u].x) + fabsf(in4[u].y) + fabsf(in4[u].z) + fabsf(in4[u].w) < 3e38f);
                    else bad = bad || !(fabsf(in2[u][0].x) + fabsf(in2[u][0].y) + fabsf(in2[u][1].x) + fabsf(in2[u][1].y) + fabsf(in2[u][2].x) + fabsf(in2[u][2].y) < 3e38f);
                }
            }
        };
        if constexpr (DYN) {
            if (big) math(IntC<PREC_F64 | PREC_FX>{});
            else math(IntC<PREC & (PREC_RESID | PREC_F64)>{});
        } else {
            math(IntC<PREC>{});
        }
        const bool fixed = big && __builtin_amdgcn_ballot_w64(bad) == 0;  // NaN / Inf rotations: the float walk propagates them
        if (i > 0) copy_out(f0_prev, nf_prev);  // the image of tile i-1 leaves ...
#pragma unroll
        for (int u = 0; u < EPL; ++u) {        // ... and tile i's local rotations take its place (in-order DS)
            const int e = u * PM_WAVE + lane;
            if (e < n) {
                if (QUAD && wide) put_local<false>(sRot + image_slot<PAD>(e, J, invJ, 9, pad), L[u]);  // (tree_walk_w4 reads rows of L)
                else put_local<QUAD>(sRot + image_slot<PAD>(e, J, invJ, 9, pad), L[u]);
                if constexpr (PFO) {
                    float *o = sOff + image_slot<PAD>(e, J, invJ, 3, pad);
                    o[0] = inO[u].x; o[1] = inO[u].y; o[2] = inO[u].z;
                }
            }
        }
        if constexpr (SRC == SRC_O6D) {  // the rare float64 redo of degenerate records, on the parked slots, before their inputs are overwritten
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const int e = u * PM_WAVE + lane, ec = e < n ? e : (n > 0 ? n - 1 : 0);
                const float xx[6] = {in2[u][0].x, in2[u][0].y, in2[u][1].x, in2[u][1].y, in2[u][2].x, in2[u][2].y};
                if (QUAD && wide) o6d_redo_ill<QOUT, false>(ill[u] && e < n, xx, a.eps, sRot + image_slot<PAD>(ec, J, invJ, 9, pad), QOUT ? gq + 4 * ec : nullptr);
                else o6d_redo_ill<QOUT, QUAD>(ill[u] && e < n, xx, a.eps, sRot + image_slot<PAD>(ec, J, invJ, 9, pad), QOUT ? gq + 4 * ec : nullptr);
            }
        }
        f0_prev = f0; nf_prev = nf;
        if (i + 1 < cnt) {
            f0 += FPW;
            nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
            issue(f0, nf);                      // in flight during the walk below
        }
        wave_sync();
        constexpr bool CAN_FX = DYN || (PREC & PREC_FX);
        if constexpr (QUAD) {
            const float seed = (c == 3) ? gp_i : ((c == r) ? 1.0f : 0.0f);
            if (wide) {  // (wave-uniform) four joints of a frame at a time
                // the roots' positions (their rotation slots hold L_0 = R_0 as parked): the twelve-lane mapping's position lanes have them
                if (c == 3 && lane < 12 * FPW) sPos[f * (J * 3 + pad) + r] = (CAN_FX && fixed) ? __int_as_float((int)__builtin_rintf(gp_i * fx.S)) : gp_i;
                wave_sync();
                if (!PM_ABLATED(a, 2)) {
                    if (CAN_FX && fixed) tree_walk_w4<PFO, CAN_FX>(sRot, sPos, sOff, sConst, J, pad, lane, JW, a.wsteps, fx.S, false);
                    else tree_walk_w4<PFO, false>(sRot, sPos, sOff, sConst, J, pad, lane, JW, a.wsteps, 1.0f, poison);
                }
            } else if (!PM_ABLATED(a, 2)) {
                if (CAN_FX && fixed) tree_walk_quad<PFO, CAN_FX>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, fx.S);
                else if (poison) tree_walk_quad<PFO, false, true>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, 1.0f);
                else tree_walk_quad<PFO, false>(sRot, sPos, sOff, sConst, J, pad, f, r, c, seed, lane, 1.0f);
            }
        } else if constexpr (Q4) {
            if ((lane & 3) < 3 && (lane >> 2) < FPW) {  // (the DPP operands come from lanes 0..2 of a quad)
                if (CAN_FX && fixed) tree_walk_q4<CAN_FX>(sRot, sPos, sConst, J, pad, f, r, gp_i, PM_ABLATED(a, 2), fx.S, false);
                else tree_walk_q4<false>(sRot, sPos, sConst, J, pad, f, r, gp_i, PM_ABLATED(a, 2), 1.0f, poison);
            }
        } else {
            if (CAN_FX && fixed) tree_walk<PFO, CAN_FX>(sRot, sPos, sOff, sConst, J, pad, f, r, gp_i, PM_ABLATED(a, 2), fx.S, false);
            else tree_walk<PFO, false>(sRot, sPos, sOff, sConst, J, pad, f, r, gp_i, PM_ABLATED(a, 2), 1.0f, poison);
        }
        wave_sync();
        if (CAN_FX && fixed) {  // fixed-point words -> fp32 in place; the root is the caller's value, bit for bit
            fx_to_float(sPos, (FPW * (J * 3 + pad)) >> 2, fx.invS, lane);
            if (!QUAD || c == 3) sPos[f * (J * 3 + pad) + r] = gp_i;
            wave_sync();
        }
    }
    copy_out(f0_prev, nf_prev);
#ifdef PM_TUNING
    if (a.times) {  // which XCD ran this workgroup, from when to when (100 MHz counter): are the eight XCDs done at the same time?
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t wg_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { a.times[3 * (uint64_t)blockIdx.x] = xcc & 15u; a.times[3 * (uint64_t)blockIdx.x + 1] = wg_start; a.times[3 * (uint64_t)blockIdx.x + 2] = wg_end; }
    }
#endif
}


// ---- long skeletons (beyond 92 joints): the three-lane row walk over a STREAMED image -------------------------------------
// The tile kernels keep a whole frame's 48 J B of output in LDS while its tree is walked: beyond ~90 joints that is 4-12 KB per frame, a CU
// holds a few dozen frames and the twelve-lanes-per-frame walk they have to use costs 3 walk instructions per joint and frame against 0.7
// for the three-lane walk (J = 128: 46 % of the HBM spec, 129: 35 %, 250: 24 %; the reference's loop, skeleton.py:51-58, has no such
// cliff).  Here a wave owns FPW frames for ALL their joints but only a CHUNK of 32 (24: see CARRY below) joints of them is in LDS at a time:
//   * 32 records of a frame are 512 B of quaternions in, 1152 B of rotation matrices and 384 B of positions out -- whole 128-byte
//     lines of every array when J is a multiple of 32, and 9- / 3-line pieces with at most two partial lines otherwise (round 3's
//     lane-per-frame attempt left 96- and 288-byte pieces, which the chip writes at 1.7-2.7 TB/s);
//   * the walk is tree_walk's: lane (f, r) carries row r of the previous joint's [R | p] in registers and reads the joint's local
//     rotation from the chunk image.  A parent that is not the previous joint comes from the image if it lies in the same chunk, else
//     from one of kFsSlots register sets the host coloured the cross-chunk branch points onto (fk_stream_plan; four floats a set);
//   * the next chunk's quaternions are requested before this chunk's walk (registers), converted and parked after its copy-out; full
//     chunks leave with a FIXED number of unconditional stores so that the wait for those quaternions is a counted one (see copy_out
//     of fk_pipe_kernel, and `run` below: no path to that wait may hold a load or a store loop of unknown length);
//   * addresses are a wave-uniform 64-bit base (scalar unit) plus 24-bit products per lane: the 64-bit multiply per vector that
//     (f0 + fe) * J * 9 costs on the vector unit was 5 points at J = 128 (54.8 -> 59.9 % on one box);
//   * the joint table is in LDS one chunk at a time (the whole table cost a wave per CU from ~130 joints on);
//   * big-magnitude tiles (PREC_DYN) take float64 local rotations and the fixed-point translation chain like every fk kernel; the
//     words stay in the image and the slots, and are converted on their way out.  A NaN / Inf that turns up in a later chunk (the tile
//     kernels look at the whole tile before they choose) poisons the word: INT_MIN travels down the chain and leaves as NaN.
constexpr int kFsSlots = 8;
enum : int { FS_CHAIN = 0xff, FS_ROOT = 0xfd, FS_LDS = 0x80, FS_NONE = 0xff };
struct FkStreamArgs {
    const float *rot, *root_pos, *offsets;
    float *pos, *rotmats;
    int64_t F;
    int32_t J, depth, ablate;
    int32_t chs;  // joints per chunk: 32 without carry, 24 with
    int32_t rs, ps;  // carry variant: floats between frames in the rotation / position image
    int32_t code[PM_MAX_JOINTS];  // joint j: load | save << 8; load = slot, FS_LDS | index inside the chunk, FS_CHAIN, FS_ROOT; save = slot or FS_NONE
};

// Host plan: which joints' rows must survive in registers (a child in a LATER chunk that does not follow directly), coloured onto
// kFsSlots sets by live range.  Returns false if the skeleton needs more.
static bool fk_stream_plan(const Parents &par, const int J, const int chs, int32_t *code) {
    int last_use[PM_MAX_JOINTS], slot_of[PM_MAX_JOINTS], busy_until[kFsSlots];
    for (int j = 0; j < J; ++j) { last_use[j] = -1; slot_of[j] = -1; }
    for (int j = 1; j < J; ++j) {
        const int p = par.p[j];
        if (p == j - 1 || p / chs == j / chs) continue;
        if (last_use[p] < j) last_use[p] = j;
    }
    for (int k = 0; k < kFsSlots; ++k) busy_until[k] = -1;
    for (int j = 0; j < J; ++j) {
        const int p = (j == 0) ? -1 : par.p[j];
        int load, save = FS_NONE;
        if (j == 0) load = FS_ROOT;
        else if (p == j - 1) load = FS_CHAIN;
        else if (p / chs == j / chs) load = FS_LDS | (p % chs);
        else load = slot_of[p];
        if (last_use[j] >= 0) {
            int k = 0;
            while (k < kFsSlots && busy_until[k] > j) ++k;
            if (k == kFsSlots) return false;
            busy_until[k] = last_use[j];
            slot_of[j] = k;
            save = k;
        }
        code[j] = load | (save << 8);
    }
    return true;
}

struct FsSaves { float g[kFsSlots][4]; };
template <int K>
__device__ __forceinline__ void fs_slot_load(const int ld, const FsSaves &sv, float &p0, float &p1, float &p2, float &pt) {
    if constexpr (K < kFsSlots) {
        int code = ld;
        asm volatile("" : "+s"(code));  // an opaque copy per test (an indexed array would live in scratch memory)
        if (code == K) { p0 = sv.g[K][0]; p1 = sv.g[K][1]; p2 = sv.g[K][2]; pt = sv.g[K][3]; }
        fs_slot_load<K + 1>(ld, sv, p0, p1, p2, pt);
    }
}
template <int K>
__device__ __forceinline__ void fs_slot_save(const int st, FsSaves &sv, const float g0, const float g1, const float g2, const float gt) {
    if constexpr (K < kFsSlots) {
        int code = st;
        asm volatile("" : "+s"(code));
        if (code == K) { sv.g[K][0] = g0; sv.g[K][1] = g1; sv.g[K][2] = g2; sv.g[K][3] = gt; }
        fs_slot_save<K + 1>(st, sv, g0, g1, g2, gt);
    }
}

// Whole 128-byte lines for ANY joint count (CARRY): a frame's segment of chunk c starts at float (f J + CHS c) 9 of `rotmats` and (f J + CHS c) 3
// of `pos` -- on a line only when J is a multiple of 32.  Written as it stands, every segment ends in two partial lines that the next chunk
// completes ~10 us later, by which time the line has left the L2 (measured with whole lines faked by padding the output rows to 32 joints:
// J = 250 40.8 -> 51.4 %, 252 42.7 -> 53.8 %).  So the image row of a frame is laid out from the LINE its segment starts in: [0, cr) holds the cr < 32
// floats the chunk before left behind, [cr, cr + len) this chunk's segment; out go the 16-byte vectors up to the last line boundary inside the row,
// the floats past it move to the row's front (two ds_read_b128 / ds_write_b128 a lane) and leave with chunk c + 1.  Only a frame's very first and
// last vector (chunk 0 / the last chunk; 16 bytes shared with the neighbouring frame) are written float by float.  Rows are CH * {9, 3} + 32 floats
// (+ what fs_pick_stride adds against bank conflicts); with 24-joint chunks that is six waves a CU, which the walk needs (it is one dependent
// chain per frame: at J = 128, 6 / 5 / 4 waves a CU measured 61.3 / 54.0 / 44.9 %, the walk switched off 64-65 % throughout).
template <int FPW, bool CARRY, int CH>
__global__ __launch_bounds__(PM_WAVE) void fk_stream_kernel(const FkStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int XR = CARRY ? 32 : 0;  // room in front of a frame's segment for the floats carried over from the chunk before (see above)
    // floats between frames in the chunk images: never a multiple of 8 (see the note on LDS bank conflicts above); with a carry the host picks them
    // per joint count (fs_pick_stride: the frames' segments start a.rs + 9 J floats apart modulo the banks)
    const int RS = CARRY ? a.rs : CH * 9 + 4, PS = CARRY ? a.ps : CH * 3 + 4;
    constexpr int EPL = (FPW * CH + PM_WAVE - 1) / PM_WAVE;
    const int lane = threadIdx.x, J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile(ntiles);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    float *sRot = smem;               // [FPW][RS]
    float *sPos = sRot + FPW * RS;    // [FPW][PS]
    float *sRoot = sPos + FPW * PS;   // [FPW][3] the frames' root positions, bit for bit (chunk 0's copy-out of a fixed-point tile)
    float *sConst = sRoot + 4 * FPW;  // [CH + 2] {-, t0, t1, t2} of the chunk being walked (the whole table would cost a wave per CU from ~130 joints on)
    const int CHS = a.chs;  // joints per chunk (<= CH, a multiple of 4)
    const int NC = (J + CHS - 1) / CHS;
    const int Jo = PM_ABLATED(a, 64) ? ((J + 31) & ~31) : J, Ji = PM_ABLATED(a, 128) ? ((J + 31) & ~31) : J;  // PM_FK_ABLATE & 64 / & 128 (tuning build, wrong results): output / input frames 32 joints apart
    // floats the segment of (chunk c, frame fe) sits past a 128-byte line of `rotmats` / `pos` (only the low bits matter)
    const int f0l = (int)(f0 & 31), jr = (Jo * 9) & 31, jp = (Jo * 3) & 31;
    const int br = (int)((reinterpret_cast<uintptr_t>(a.rotmats) >> 2) & 31), bp = (int)((reinterpret_cast<uintptr_t>(a.pos) >> 2) & 31);
    auto carry_r = [&](const int c, const int fe) __attribute__((always_inline)) { return CARRY ? ((br + __mul24(f0l + fe, jr) + c * CHS * 9) & 31) : 0; };
    auto carry_p = [&](const int c, const int fe) __attribute__((always_inline)) { return CARRY ? ((bp + __mul24(f0l + fe, jp) + c * CHS * 3) & 31) : 0; };

    // what the joint table says about the arithmetic this tile needs (PREC_DYN, see fk_tile)
    bool tbig_l = false;
    float tsum_l = 0.0f, tmx_l = 0.0f;
    for (int j = lane; j < J; j += PM_WAVE) {
        const int jc = j < J ? j : J - 1;
        const bool none = j == 0;  // offsets[0] is ignored (skeleton.py:49)
        const v4f cj = v4f{0.0f, none ? 0.0f : a.offsets[3 * jc], none ? 0.0f : a.offsets[3 * jc + 1], none ? 0.0f : a.offsets[3 * jc + 2]};
        if (j < J) { const float l1 = const_l1(cj); tbig_l = tbig_l || const_is_big(cj); tsum_l += l1; tmx_l = (l1 > tmx_l || l1 != l1) ? l1 : tmx_l; }
    }
    constexpr bool Q4 = FPW == 16;  // a quad per frame, L shared through DPP (see tree_walk_q4): a third of the walk's LDS reads
    const int wl = lane % (3 * FPW);
    const int f = Q4 ? (lane >> 2) : wl / 3, r = Q4 ? ((lane & 3) < 3 ? (lane & 3) : 2) : wl - 3 * f;  // (Q4: lane 3 of a quad shadows lane 2 and sits the walk out)
    const int64_t fg = f0 + (f < nf ? f : nf - 1);  // frames past a partial tile repeat its last one (their stores are predicated)
    const float gp = a.root_pos[fg * 3 + r];
    if (lane < 3 * FPW || Q4) sRoot[f * 3 + r] = gp;
    bool big = false, tpoison = false;
    FxScale fx = {1.0f, 1.0f};
    {
        const bool tbig = __builtin_amdgcn_ballot_w64(tbig_l) != 0;
        const float bsum = wave_sum(tsum_l), bmax = (float)a.depth * wave_max(tmx_l);  // (NaN sticks in both)
        big = tbig || __builtin_amdgcn_ballot_w64(!(fabsf(gp) < kBigRoot)) != 0;
        if (big) { big = fx_scale((bmax < bsum) ? bmax : bsum, fabsf(gp), fx); tpoison = !big; }  // false for a non-finite bound: the float walk propagates NaN / Inf (tpoison: see tree_walk's `poison`)
    }

    v4f in4[EPL];
    v4f cn_next = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int cv_next = 0;  // the chunk's per-joint codes across the lanes (lane i: joint 32 c + i): one v_readlane per step -- an s_load inside the
                      // walk shares lgkmcnt with its LDS traffic and drains it on every joint (measured: 10.7 us per chunk with the s_load)
    auto issue = [&](const int c) __attribute__((always_inline)) {  // chunk c's quaternions -> registers; record e = (frame e / nj, joint e % nj of the chunk)
        const int nj = (J - c * CHS) < CHS ? (J - c * CHS) : CHS;
        const float inv = 1.0f / (float)nj;
        const v4f *src = reinterpret_cast<const v4f *>(a.rot) + (f0 * Ji + c * CHS);  // (wave-uniform: the 64-bit product stays on the scalar unit)
        {
            const int jc = c * CHS + (lane & 31);
            cv_next = a.code[jc < J ? jc : J - 1];
            const int jt = c * CHS + lane, jo = jt < J ? jt : J - 1;  // the chunk's rows of the joint table (lanes 0 .. CH + 1; one slot of slack for the walk's look-ahead)
            const bool none = jt == 0;  // offsets[0] is ignored (skeleton.py:49)
            const float o0 = a.offsets[3 * jo], o1 = a.offsets[3 * jo + 1], o2 = a.offsets[3 * jo + 2];  // (unconditional: a load under a branch would end the counted waits)
            cn_next = v4f{0.0f, none ? 0.0f : o0, none ? 0.0f : o1, none ? 0.0f : o2};
        }
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = u * PM_WAVE + lane;
            int fe = (int)(((float)e + 0.5f) * inv);
            const int jl = e - fe * nj;
            fe = fe < nf ? fe : nf - 1;
            in4[u] = __builtin_nontemporal_load(src + (__mul24(fe, Ji) + jl));
        }
    };
    auto park = [&](const int c, auto mode) __attribute__((always_inline)) {  // phase A for chunk c: quaternion -> local rotation -> its slot of the image
        constexpr int M = decltype(mode)::value;
        const int nj = (J - c * CHS) < CHS ? (J - c * CHS) : CHS;
        const float inv = 1.0f / (float)nj;
        if (lane < CH + 2) reinterpret_cast<v4f *>(sConst)[lane] = cn_next;
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
            const int e = u * PM_WAVE + lane;
            const int fe = (int)(((float)e + 0.5f) * inv), jl = e - fe * nj;
            const float qi[4] = {in4[u].x, in4[u].y, in4[u].z, in4[u].w};
            float L[9];
            local_from_quat<M>(qi, L);
            if (fe < FPW && !PM_ABLATED(a, 32)) lds_put<9>(sRot + __mul24(fe, RS) + carry_r(c, fe) + jl * 9, 0, L);  // (& 32: without phase A's LDS writes)
            if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two conversions in flight, not EPL
        }
    };

    // ---- the walk over one chunk (tree_walk's step; FX: positions as fixed-point words) ----
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, gt = 0.0f;
    FsSaves sv;
#pragma unroll
    for (int k = 0; k < kFsSlots; ++k) { sv.g[k][0] = 0.0f; sv.g[k][1] = 0.0f; sv.g[k][2] = 0.0f; sv.g[k][3] = 0.0f; }
    auto walk = [&](const int c, const int cv, auto fxmode) __attribute__((always_inline)) {
        constexpr bool FX = decltype(fxmode)::value != 0;
        const int nj = (J - c * CHS) < CHS ? (J - c * CHS) : CHS;
        float *fL = sRot + __mul24(f, RS) + carry_r(c, f), *fRot = fL + r * 3, *fPos = sPos + __mul24(f, PS) + carry_p(c, f) + r;
        const v4f *cst = reinterpret_cast<const v4f *>(sConst);
        auto dot_bcast = [](const float l, const float p0, const float p1, const float p2) __attribute__((always_inline)) {  // p0 L[0][c] + p1 L[1][c] + p2 L[2][c], L[k][c] from lane k of the quad
            float acc;
            asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
                : "=&v"(acc)
                : "v"(l), "v"(p0), "v"(p1), "v"(p2));
            return acc;
        };
        constexpr int NL = Q4 ? 3 : 9;  // floats of L_j a lane reads: its own row (Q4) or all of it
        auto getL = [&](const int jl, float (&L)[NL]) __attribute__((always_inline)) {
            if constexpr (Q4) { L[0] = fRot[jl * 9]; L[1] = fRot[jl * 9 + 1]; L[2] = fRot[jl * 9 + 2]; }
            else lds_get<9>(fL, jl, L);
        };
        auto joint = [&](const int jl, const float (&L)[NL], const v4f cj) __attribute__((always_inline)) {
            const int code = __builtin_amdgcn_readlane(cv, jl);  // wave-uniform
            const int ld = code & 0xff, st = (code >> 8) & 0xff;
            float p0 = g0, p1 = g1, p2 = g2, pt = gt;
            if (ld == FS_ROOT) {  // e_r . L = row r of L (exact), translation = the root position (offsets[0] ignored)
                p0 = (r == 0) ? 1.0f : 0.0f; p1 = (r == 1) ? 1.0f : 0.0f; p2 = (r == 2) ? 1.0f : 0.0f;
                pt = FX ? __int_as_float((int)__builtin_rintf(gp * fx.S)) : gp;
            } else if (ld != FS_CHAIN) {
                if (ld & FS_LDS) {
                    const int pl = ld & 0x7f;
                    p0 = fRot[pl * 9]; p1 = fRot[pl * 9 + 1]; p2 = fRot[pl * 9 + 2];
                    pt = fPos[pl * 3];
                } else {
                    fs_slot_load<0>(ld, sv, p0, p1, p2, pt);
                }
            }
            if constexpr (Q4) {
                g0 = dot_bcast(L[0], p0, p1, p2); g1 = dot_bcast(L[1], p0, p1, p2); g2 = dot_bcast(L[2], p0, p1, p2);
            } else {
                g0 = __builtin_fmaf(p2, L[6], __builtin_fmaf(p1, L[3], p0 * L[0]));
                g1 = __builtin_fmaf(p2, L[7], __builtin_fmaf(p1, L[4], p0 * L[1]));
                g2 = __builtin_fmaf(p2, L[8], __builtin_fmaf(p1, L[5], p0 * L[2]));
            }
            const float dt = __builtin_fmaf(p2, cj.w, __builtin_fmaf(p1, cj.z, p0 * cj.y));
            if (FX) {
                const int pw = __float_as_int(pt);
                const bool poison = pw == (int)0x80000000 || !(fabsf(dt) < 3e38f);  // a NaN / Inf met on the way: it leaves as NaN, and so does everything below
                gt = __int_as_float(poison ? (int)0x80000000 : pw + (int)__builtin_rintf(dt * fx.S));
            } else {
                gt = dt + pt;
                if (tpoison && ld != FS_ROOT) poison_row(pt, g0, g1, g2);
            }
            fRot[jl * 9] = g0; fRot[jl * 9 + 1] = g1; fRot[jl * 9 + 2] = g2;
            fPos[jl * 3] = gt;
            if (st != FS_NONE) fs_slot_save<0>(st, sv, g0, g1, g2, gt);
        };
        if (Q4 && (lane & 3) == 3) return;  // (the DPP operands come from lanes 0..2 of a quad)
        float La[NL], Lb[NL];
        v4f ca, cb;
        getL(0, La);
        ca = cst[0];
        for (int jl = PM_ABLATED(a, 2) ? nj : 0; jl < nj; jl += 2) {
            getL(jl + 1, Lb);  // (slot jl + 1 still holds L; one slot of slack past the chunk lies inside the frame's padding + the next frame)
            cb = cst[jl + 1];
            joint(jl, La, ca);
            if (jl + 1 >= nj) break;
            getL(jl + 2, La);
            ca = cst[jl + 2];
            joint(jl + 1, Lb, cb);
        }
    };

    // ---- copy-out of chunk c: per frame nj * 9 floats of rotation matrices and nj * 3 of positions, contiguous in HBM ----
    // One 16-byte vector of a frame's (shifted) image: `lo` .. `hi` = the floats of it that belong to this segment.
    auto out_vec = [&](float *g, const float *l, const int lo, const int hi, const bool fx_words, const int root_q0, const int fe) __attribute__((always_inline)) {
        // fixed-point words are converted here (the image keeps them for the walk); root_q0: where in this vector the frame's root position
        // starts (chunk 0 of `pos` only, else out of reach): the root is the caller's value, bit for bit (skeleton.py:49)
        const v4f raw = *reinterpret_cast<const v4f *>(l);
        float v[4] = {raw.x, raw.y, raw.z, raw.w};
        if (fx_words) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = __float_as_int(v[q]);
                v[q] = (w == (int)0x80000000) ? __builtin_nanf("") : (float)w * fx.invS;
                if (root_q0 < 100) {  // (chunk 0 of `pos` only; from LDS: a global load here would change the count of the waits behind it)
                    const int ri = q - root_q0;
                    if (ri >= 0 && ri < 3) v[q] = sRoot[fe * 3 + ri];
                }
            }
        }
        if (lo == 0 && hi == 4) {
            *reinterpret_cast<v4f *>(g) = v4f{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q >= lo && q < hi) g[q] = v[q];
        }
    };
    // segment geometry of (chunk c, frame fe): cr = its carry, len = floats of the segment.  Float i of the image row is float i of the 128-byte
    // line the segment starts in; the row holds [0, cr) = what chunk c - 1 left behind, [cr, cr + len) = this chunk.  Out go the full vectors
    // kf .. ke - 1: from the frame's first float (chunk 0) or the line boundary on, up to the last line boundary inside the row (or the
    // frame's end, last chunk); the floats past it move to the row's front for chunk c + 1.
    auto copy_region = [&](float *gbase, float *lbase, const int stride, auto per_joint_c, auto rootc_c, const int c, const int nj, const bool fx_words, auto full_c, auto carry_of)
        __attribute__((always_inline)) {
        constexpr int per_joint = decltype(per_joint_c)::value;
        constexpr bool full_chunk = decltype(full_c)::value != 0;  // every chunk but the last is one
        constexpr bool rootc = decltype(rootc_c)::value != 0;  // chunk 0 of `pos` on a fixed-point tile: the root leaves as the caller's bits
        const int len = nj * per_joint;
        const bool firstc = c == 0, lastc = c == NC - 1;
        float *gtile = gbase + (f0 * Jo + c * CHS) * per_joint;  // (wave-uniform) frame fe's segment starts fe * gstep floats further on
        const int gstep = Jo * per_joint;
        auto first_vec = [&](const int cr) __attribute__((always_inline)) { return firstc ? ((cr + 3) >> 2) : 0; };
        auto end_vec = [&](const int cr) __attribute__((always_inline)) { return (lastc || !CARRY) ? ((cr + len) >> 2) : (((cr + len) >> 5) << 3); };
        // (1) the partial first vector of a frame (chunk 0) and its partial last one (last chunk), float by float: they share their 16 bytes with the neighbouring frame
        if (CARRY && (firstc || lastc) && lane < 2 * nf && !PM_ABLATED(a, 16)) {  // PM_FK_ABLATE & 16 (tuning build): without the partial vectors
            const int fe = lane >> 1, last = lane & 1;
            const int cr = carry_of(c, fe), end = cr + len;
            float *g = gtile + (__mul24(fe, gstep) - cr);
            if (!last && firstc && (cr & 3) != 0) { const int k = cr >> 2; out_vec(g + 4 * k, lbase + __mul24(fe, stride) + 4 * k, cr & 3, 4, fx_words, rootc ? cr - 4 * k : 100, fe); }
            if (last && lastc && (end & 3) != 0) { const int k = end >> 2; out_vec(g + 4 * k, lbase + __mul24(fe, stride) + 4 * k, 0, end & 3, fx_words, 100, fe); }
        }
        // (2) the full vectors: unconditional dwordx4, for a full chunk a FIXED number of them (slots past a frame's full vectors, or past the
        // tile's end, repeat a vector that is stored anyway) so that the wait for the next chunk's quaternions stays a counted one.
        // With a carry, FOUR LANES A FRAME (round 5): lane (fe, q) takes the frame's vectors kf + q, kf + q + 4, ... -- one row base, one clamp and
        // two shifts per vector, where round 4's form (kept below for segments that start on lines: a wave instruction covers one frame's
        // kilobyte) pays a division by the vectors per frame, the frame's carry and both row bases per vector.  What that buys, measured in
        // ONE process on the same arrays (the streamed walk's timing depends on where the allocator put the arrays: boxes and runs differ by
        // 5-8 points on it while the tile kernels repeat to a point, so separate runs say nothing -- tools/fk_long_ab.py,
        // profiles/r05_fk_long_ab.txt): J = 100 / 112 / 144 / 200 / 400 +0.9 / +2.1 / +1.0 / +1.8 / +0.3 points, 129 / 130 / 250 / 300 / 511 the
        // same; without a carry 0.5-1.1 points BEHIND.  The copy-out's index arithmetic was not what holds the carry variant at 48-59 %.
        constexpr int PV = (CH * per_joint + XR) / 4;  // vectors of a full segment and its carry: an upper bound per frame
        if constexpr (CARRY) {
            constexpr int LPF = PM_WAVE / FPW;             // lanes per frame in the copy-out
            {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int fe0 = ln / LPF, q = ln - fe0 * LPF;
                const int fe = fe0 < nf ? fe0 : nf - 1;
                const int cr = carry_of(c, fe), kf = first_vec(cr), ke = end_vec(cr);
                const float *lrow = lbase + __mul24(fe, stride);
                float *grow = gtile + (__mul24(fe, gstep) - cr);
                if constexpr (full_chunk) {
                    constexpr int NT = (PV + LPF - 1) / LPF;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int k0 = kf + q + LPF * t, k = k0 < ke ? k0 : ke - 1;
                        out_vec(grow + 4 * k, lrow + 4 * k, 0, 4, fx_words, rootc ? cr - 4 * k : 100, fe);
                    }
                } else {
                    const int nt = ((len + XR + 3) >> 2) + LPF - 1;  // (a short last chunk: as many trips as its longest frame can need)
                    for (int t = 0; LPF * t < nt; ++t) {
                        const int k = kf + q + LPF * t;
                        if (k < ke && fe0 < nf) out_vec(grow + 4 * k, lrow + 4 * k, 0, 4, fx_words, rootc ? cr - 4 * k : 100, fe);
                    }
                }
            }
        } else {
            if constexpr (full_chunk) {
                constexpr int NS = (FPW * PV + PM_WAVE - 1) / PM_WAVE;
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const float ipv = 1.0f / (float)PV;
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    const int i = u * PM_WAVE + ln;
                    int fe = (int)(((float)i + 0.5f) * ipv);
                    int kk = i - fe * PV;
                    if (fe >= nf) { fe = nf - 1; kk = 0; }
                    const int cr = carry_of(c, fe), kf = first_vec(cr), nfull = end_vec(cr) - kf;
                    const int k = kf + (kk < nfull ? kk : nfull - 1);
                    out_vec(gtile + (__mul24(fe, gstep) - cr + 4 * k), lbase + (__mul24(fe, stride) + 4 * k), 0, 4, fx_words, rootc ? cr - 4 * k : 100, fe);
                }
            } else {
                const int pv = (len + XR + 3) >> 2;  // (a short last chunk: as many slots per frame as it can have vectors)
                const float ipv = 1.0f / (float)pv;
                for (int i = lane; i < nf * pv; i += PM_WAVE) {
                    const int fe = (int)(((float)i + 0.5f) * ipv), kk = i - fe * pv;
                    const int cr = carry_of(c, fe), kf = first_vec(cr), nfull = end_vec(cr) - kf;
                    if (kk < nfull) out_vec(gtile + (__mul24(fe, gstep) - cr + 4 * (kf + kk)), lbase + (__mul24(fe, stride) + 4 * (kf + kk)), 0, 4, fx_words, rootc ? cr - 4 * (kf + kk) : 100, fe);
                }
            }
        }
        // (3) what lies past the last line boundary (less than 32 floats a frame) moves to the front of the row: chunk c + 1 lands behind it
        if (CARRY && !lastc) {
            wave_sync();  // (the copy-out's reads of these rows come first: other lanes read what this lane overwrites)
            for (int i = lane; i < FPW * 8; i += PM_WAVE) {
                const int fe = i >> 3, q = i & 7;
                const int end = carry_of(c, fe) + len, we = (end >> 5) << 5;
                float *row = lbase + __mul24(fe, stride);
                if (we + 4 * q < end) {  // only the vectors that hold carried floats (nothing is read past the row's end)
                    const v4f v = *reinterpret_cast<const v4f *>(row + we + 4 * q);
                    *reinterpret_cast<v4f *>(row + 4 * q) = v;
                }
            }
            wave_sync();  // (... and the next chunk's local rotations land behind the carried floats)
        }
    };
    auto copy_out = [&](const int c, const bool fixed_words, auto full_c) __attribute__((always_inline)) {
        const int nj = (J - c * CHS) < CHS ? (J - c * CHS) : CHS;
        copy_region(a.rotmats, sRot, RS, IntC<9>{}, IntC<0>{}, c, nj, false, full_c, carry_r);
        if (fixed_words && c == 0) copy_region(a.pos, sPos, PS, IntC<3>{}, IntC<1>{}, c, nj, true, full_c, carry_p);
        else copy_region(a.pos, sPos, PS, IntC<3>{}, IntC<0>{}, c, nj, fixed_words, full_c, carry_p);
    };

    // Chunks with a successor are full ones, and their copy-out is straight-line code with a fixed number of stores: the wait for chunk c + 1's
    // quaternions (issued before the walk of chunk c) is then a counted one and the stores drain under the next walk.  A store loop of unknown length
    // on ANY path to that wait turns it into vmcnt(7 - u) -- all stores acknowledged -- which is why the last chunk has its own copy of the code.
    auto run = [&](auto mode) __attribute__((always_inline)) {
        constexpr int M = decltype(mode)::value;
        constexpr bool FX = (M & PREC_FX) != 0;
        issue(0);
        park(0, mode);
        int c = 0;
        for (; c + 1 < NC; ++c) {
            const int cv = cv_next;
            asm volatile("" ::"v"(cv));  // settle the code load here, not inside the walk
            issue(c + 1);  // in flight during the walk below
            wave_sync();
            walk(c, cv, IntC<FX ? 1 : 0>{});
            wave_sync();
            copy_out(c, FX, IntC<1>{});
            wave_sync();        // (other lanes' copy-out reads of the slots this lane is about to overwrite come first)
            park(c + 1, mode);
        }
        const int cv = cv_next;
        wave_sync();
        walk(c, cv, IntC<FX ? 1 : 0>{});
        wave_sync();
        if (J - c * CHS == CHS) copy_out(c, FX, IntC<1>{});
        else copy_out(c, FX, IntC<0>{});
    };
    if (big) run(IntC<PREC_F64 | PREC_FX>{});
    else run(IntC<PREC_RESID>{});
}

// Carry variant: the segment of frame f starts f (stride + step) + const floats into the banks, step = floats per frame in HBM.  Of the strides that
// keep rows 16-byte aligned, the first one under which the walk's accesses (lane -> frame f, row r; word f (stride + step) + mul r) collide least in
// a 32-lane group of ds_read_b32 / ds_write_b32 (32 banks): measured before, with one stride for all, J = 252 (eight frames on one bank) 37.5 %.
template <int FPW>
static int fs_pick_stride(const int min_floats, const int step, const int mul) {
    int best = min_floats, best_cost = 1 << 30;
    for (int extra = 0; extra < 32; extra += 4) {
        const int sb = (min_floats + extra + step) & 31;
        int cost = 0;
        for (int g = 0; g < 2; ++g) {
            int cnt[32] = {0}, seen_n = 0, seen[32];
            for (int lane = 32 * g; lane < 32 * g + 32; ++lane) {
                const int wl = lane % (3 * FPW);
                const int f = FPW == 16 ? (lane >> 2) : wl / 3, r = FPW == 16 ? ((lane & 3) < 3 ? (lane & 3) : 2) : wl - 3 * f;
                const int word = f * 4096 + mul * r;  // (distinct addresses: lanes on one address are one access)
                bool dup = false;
                for (int k = 0; k < seen_n; ++k) dup = dup || seen[k] == word;
                if (dup) continue;
                seen[seen_n++] = word;
                const int b = (f * sb + mul * r) & 31;
                if (++cnt[b] > cost) cost = cnt[b];
            }
        }
        if (cost < best_cost) { best_cost = cost; best = min_floats + extra; }  // (a tie-break on the copy-out's banks -- rows half the banks apart -- measured the same to 0.1 point)
    }
    return best;
}

template <int FPW, bool CARRY, int CH>
static int launch_fk_stream(FkStreamArgs &a, hipStream_t s) {
    int RS = CH * 9 + 4, PS = CH * 3 + 4;
    if (CARRY) {
        a.rs = RS = tune_env("PM_FKS_RS", fs_pick_stride<FPW>(CH * 9 + 32, a.J * 9, 3));  // PM_TUNING build only
        a.ps = PS = tune_env("PM_FKS_PS", fs_pick_stride<FPW>(CH * 3 + 32, a.J * 3, 1));
    }
    const size_t lds = ((size_t)FPW * (RS + PS + 4) + 4 * ((size_t)CH + 2) + 16) * sizeof(float) + (size_t)tune_env("PM_FKS_LDSX", 0);  // (PM_TUNING build only: bytes of LDS on top, fewer waves a CU)
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("fk: grid too large"); return PM_EUNSUPPORTED; }
    auto k = fk_stream_kernel<FPW, CARRY, CH>;
    if (int e = allow_lds(k, lds)) return e;
    set_kernel_name("void pm::fk_stream_kernel<%d, %s, %d>(pm::FkStreamArgs)", FPW, CARRY ? "true" : "false", CH);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH("fk launch");
}

// quaternion source, shared offsets, 16-byte aligned arrays; false: not eligible (the caller falls back to the tile kernels)
static bool try_fk_stream(const FkArgs &fa, hipStream_t s, int &rc) {
    FkStreamArgs a;
    // segments that start and end on 128-byte lines of both outputs (32-joint chunks of a skeleton whose joint count is a multiple of 32) need no carry
    const bool lines = fa.J % 32 == 0 && ((reinterpret_cast<uintptr_t>(fa.pos) | reinterpret_cast<uintptr_t>(fa.rotmats)) & 127) == 0;
    // chunks of 24 joints leave the carry variant six waves a CU (32: five); measured on one box, 24 / 32: J = 100 56.2 / 50.5 %, 120 58.4 / 51.4,
    // 144 58.7 / 52.0, 200 58.9 / 53.7, 250 50.3 / 51.2, 300 56.8 / 55.1, 400 61.6 / 57.1, 511 59.4 / 57.3 (28: between the two, ahead only where 28 divides J)
    const int chs_auto = lines ? 32 : 24;
    a.chs = tune_env("PM_FKS_CHS", chs_auto);  // PM_TUNING build only: 24 or 32
    if (a.chs != 24 && a.chs != 32) a.chs = chs_auto;
    if (!fk_stream_plan(fa.parents, fa.J, a.chs, a.code)) return false;
    a.rot = fa.src; a.root_pos = fa.root_pos; a.offsets = fa.offsets; a.pos = fa.pos; a.rotmats = fa.rotmats;
    a.F = fa.F; a.J = fa.J; a.depth = fa.depth; a.ablate = fa.ablate;
    // (twenty frames a wave -- five waves a CU -- were ahead beyond 384 joints while the whole joint table sat in LDS; with the chunk's rows only,
    // sixteen are: J = 512 62.1 / 57.4 %, 400 54.0 / 52.1; the kernel is instantiated for sixteen)
    const bool carry = tune_env("PM_FKS_CARRY", 0) != 0 || !lines || a.chs != 32;
    if (!carry) rc = launch_fk_stream<16, false, 32>(a, s);
    else if (a.chs == 24) rc = launch_fk_stream<16, true, 24>(a, s);
    else rc = launch_fk_stream<16, true, 32>(a, s);
    return true;
}

// Arithmetic of the production library (see local_from_quat); the PM_TUNING build can override it per call (PM_FK_PREC)
// on the main variants to measure what each step costs.
#ifndef PM_FK_PREC_DEFAULT
#define PM_FK_PREC_DEFAULT (PREC_DYN | PREC_RESID)
#endif
constexpr int kBigResidMaxDepth = 7;
constexpr int kFkEightFramesMaxJ = 39, kFkEightFramesMaxJO6d = 39, kFkEightFramesMaxJO6dQ = 31;  // see dispatch_fk
// fk_stream_kernel: every skeleton beyond 128 joints; below, multiples of 32 from 64 on (whole lines, no carry) and multiples of 4 from 96 on (chain-like,
// 2^19 frames, stream / pipelined tile kernel on one box: J = 64 64.6 / 59.6 %, 96 68.0 / 55.9, 100 56.2 / 52.1, 104 57.2 / 53.1, 112 58.7 / 46.4,
// 120 58.4 / 48.1, 128 69.2 / 47.3; 80 54.1 / 57.9, 97 53.9 / 56.4, 127 50.2 / 47.4)
constexpr int kFkStreamMinJ = 96, kFkStreamMinLinesJ = 64;
constexpr int kFkWideMinJ = 100;  // fk_wide_kernel (fkwide.hip) beyond; up to here the four-frame pipelined tiles with tree_walk_w4 are the better shape on the same trees
static bool fk_stream_wanted(const int J) { return J > 128 || (J >= kFkStreamMinJ && J % 4 == 0) || (J >= kFkStreamMinLinesJ && J % 32 == 0); }

// ---- tree_walk_q4, sixteen frames a wave: which eight frames share a half-wave (round 6) -------------------------------------------------
// Every DS instruction of that walk is a ds_read_b32 / ds_write_b32 whose 64 lanes touch the SAME joint slot of sixteen frames: lane (q, r) row r of
// frame q's slot, i.e. float  frame * S + 9 j + 3 r + k  of the rotation image (S = 9 J + pad floats between frames) and  frame * S' + 3 j + r  of the
// position image (S' = 3 J + pad).  The LDS serves a b32 instruction as two groups of 32 lanes over 32 banks of one dword, one cycle a group plus
// one for every further distinct address on the busiest bank: with frames 0..7 in the first half-wave and S = 198 (J = 22) frame f + 1's row 0 sits on
// the bank of frame f's row 2 -- 6 f + 6 -- and every row read and write of the walk takes twice its cycles (rocprofv3, round 5: SQ_LDS_BANK_CONFLICT
// 22.7 M cycles against 23.7 M SQ_ACTIVE_INST_LDS on the headline kernel).  Which frames share a half-wave is free: quad q may walk any frame of the
// tile.  With the EVEN frames in one half and the odd ones in the other the strides double (12 and 4 at J = 22) and the 24 addresses of a group fall on
// 24 different banks.  The host picks, per joint count, the split with the fewest extra cycles from a few families (brute force over all 6435 splits
// finds nothing better for any J <= 29 but J = 9 / 23, whose optimum is the fifth candidate); J = 11, 15, 17, 21, 25, 29 have no conflict-free split.
static uint64_t q4_frame_map(const int J, const int pad) {
    static const uint8_t kSplits[5][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},   // blocks (rounds 4-5)
        {0, 2, 4, 6, 8, 10, 12, 14, 1, 3, 5, 7, 9, 11, 13, 15},   // even | odd
        {0, 1, 4, 5, 8, 9, 12, 13, 2, 3, 6, 7, 10, 11, 14, 15},   // pairs
        {0, 3, 4, 7, 8, 11, 12, 15, 1, 2, 5, 6, 9, 10, 13, 14},
        {0, 2, 4, 5, 7, 9, 12, 14, 1, 3, 6, 8, 10, 11, 13, 15}};
    const int Sr = 9 * J + pad, Sp = 3 * J + pad;
    auto extra = [](const uint8_t *fr, const int S, const int rstep) {  // extra LDS cycles of one b32 instruction over one 32-lane group
        int addr[32][4], n[32];
        for (int b = 0; b < 32; ++b) n[b] = 0;
        int worst = 1;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 3; ++r) {
                const int a = fr[i] * S + r * rstep, b = a & 31;
                bool seen = false;
                for (int k = 0; k < n[b]; ++k) seen = seen || addr[b][k] == a;
                if (!seen && n[b] < 4) { addr[b][n[b]++] = a; if (n[b] > worst) worst = n[b]; }
            }
        return worst - 1;
    };
    int best = 0, best_cost = 1 << 30;
    for (int c = 0; c < 5; ++c) {
        const int cost = 3 * (extra(kSplits[c], Sr, 3) + extra(kSplits[c] + 8, Sr, 3)) + extra(kSplits[c], Sp, 1) + extra(kSplits[c] + 8, Sp, 1);
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    best = tune_env("PM_FK_FMAP", best);  // PM_TUNING build only: 0...4
    uint64_t m = 0;
    for (int q = 0; q < 16; ++q) m |= (uint64_t)kSplits[best < 0 || best > 4 ? 0 : best][q] << (4 * q);
    return m;
}

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT, bool PAD, int PREC>
static int launch_fk_pp(const FkArgs &a, hipStream_t s) {
    const size_t lds = ((size_t)FPW * (a.J * fk_lds_floats<SRC, PFO, QOUT>() + a.pad * (PFO ? 3 : 2)) + 4 * (a.J + 4)) * sizeof(float);
    auto k = fk_kernel<FPW, VEC, PFO, SRC, QOUT, PAD, PREC>;
    if (int e = allow_lds(k, lds)) return e;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) {
        set_error("fk: %lld tiles exceed the grid limit", (long long)grid);
        return PM_EUNSUPPORTED;
    }
    set_kernel_name("void pm::fk_kernel<%d, %s, %s, %d, %s, %s, %d>(pm::FkArgs)", FPW, tf(VEC), tf(PFO), SRC, tf(QOUT), tf(PAD), PREC);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH("fk launch");
}

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT, bool PAD>
static int launch_fk_p(const FkArgs &a, hipStream_t s) {
#ifdef PM_TUNING
    if constexpr (VEC && !PFO && !QOUT) {
        switch (tune_env("PM_FK_PREC", PM_FK_PREC_DEFAULT)) {
            case 0: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 0>(a, s);
            case 1: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 1>(a, s);
            case 2: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 2>(a, s);
            case 5: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 5>(a, s);
            case 6: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 6>(a, s);
            case 17: break;  // production's own choice, below
            case 49: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 49>(a, s);
            case 117: return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, 17>(a, s);  // 17 without the shallow-skeleton shortcut
            default: set_error("PM_FK_PREC must be 0, 1, 2, 5, 6, 17, 49 or 117"); return PM_EINVAL;
        }
    }
#endif
    // shallow skeletons (the 22-joint body: depth 7): big-magnitude tiles keep the fp32 rotations, see PREC_BIG_RESID
    if constexpr (SRC == SRC_QUAT && !PFO && FPW > 5) {
        if (a.depth <= kBigResidMaxDepth) return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, PM_FK_PREC_DEFAULT | PREC_BIG_RESID>(a, s);
    }
    return launch_fk_pp<FPW, VEC, PFO, SRC, QOUT, PAD, PM_FK_PREC_DEFAULT>(a, s);
}

template <int FPW, int EPL, bool VEC, int SRC, bool QOUT, bool PAD, bool PFO, int PREC>
static int launch_fk_pipe_pp(const FkArgs &a, const int nt, hipStream_t s) {
    const size_t lds = ((size_t)FPW * (a.J * (12 + (PFO ? 3 : 0)) + (PFO ? 3 : 2) * a.pad) + 4 * (a.J + 4)) * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW, ngroups = (ntiles + nt - 1) / nt;
    const int64_t grid = ((ngroups + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("fk: grid too large"); return PM_EUNSUPPORTED; }
    if constexpr (VEC && !PAD && !QOUT) {
        if (a.J % 4 == 0 && tune_env("PM_FK_FIXED_STORES", 1)) {  // every tile, full or partial, is whole dwordx4: fixed-count copy-out
            auto kf = fk_pipe_kernel<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, PREC, true>;
            if (int e = allow_lds(kf, lds)) return e;
            set_kernel_name("void pm::fk_pipe_kernel<%d, %d, %s, %d, %s, %s, %s, %d, true>(pm::FkArgs, int)", FPW, EPL, tf(VEC), SRC, tf(QOUT), tf(PAD), tf(PFO), PREC);
            hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
            return PM_AFTER_LAUNCH("fk launch");
        }
    }
    auto k = fk_pipe_kernel<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, PREC>;
    if (int e = allow_lds(k, lds)) return e;
    set_kernel_name("void pm::fk_pipe_kernel<%d, %d, %s, %d, %s, %s, %s, %d>(pm::FkArgs, int)", FPW, EPL, tf(VEC), SRC, tf(QOUT), tf(PAD), tf(PFO), PREC);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, nt);
    return PM_AFTER_LAUNCH("fk launch");
}

template <int FPW, int EPL, bool VEC, int SRC, bool QOUT, bool PAD, bool PFO>
static int launch_fk_pipe_p(const FkArgs &a, const int nt, hipStream_t s) {
#ifdef PM_TUNING
    if constexpr (VEC && !PFO && !QOUT && EPL == 4) {
        switch (tune_env("PM_FK_PREC", PM_FK_PREC_DEFAULT)) {
            case 0: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 0>(a, nt, s);
            case 1: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 1>(a, nt, s);
            case 2: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 2>(a, nt, s);
            case 5: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 5>(a, nt, s);
            case 6: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 6>(a, nt, s);
            case 17: case 49: case 117: return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, 17>(a, nt, s);  // (the shallow-skeleton shortcut is the three-lane kernels')
            default: set_error("PM_FK_PREC must be 0, 1, 2, 5, 6, 17, 49 or 117"); return PM_EINVAL;
        }
    }
#endif
    return launch_fk_pipe_pp<FPW, EPL, VEC, SRC, QOUT, PAD, PFO, PM_FK_PREC_DEFAULT>(a, nt, s);
}

template <int FPW, int EPL, bool VEC, int SRC, bool QOUT>
static int launch_fk_pipe(const FkArgs &a, const bool pfo, const int nt, hipStream_t s) {
    if (pfo) return a.pad ? launch_fk_pipe_p<FPW, EPL, VEC, SRC, QOUT, true, true>(a, nt, s) : launch_fk_pipe_p<FPW, EPL, VEC, SRC, QOUT, false, true>(a, nt, s);
    return a.pad ? launch_fk_pipe_p<FPW, EPL, VEC, SRC, QOUT, true, false>(a, nt, s) : launch_fk_pipe_p<FPW, EPL, VEC, SRC, QOUT, false, false>(a, nt, s);
}

template <int FPW, int EPL, int SRC>
static int dispatch_fk_pipe(const FkArgs &a, bool vec, bool pfo, const int nt, hipStream_t s) {
    const bool qout = a.quat_out != nullptr;
    if constexpr (SRC == SRC_O6D) {
        if (qout) return vec ? launch_fk_pipe<FPW, EPL, true, SRC, true>(a, pfo, nt, s) : launch_fk_pipe<FPW, EPL, false, SRC, true>(a, pfo, nt, s);
    }
    return vec ? launch_fk_pipe<FPW, EPL, true, SRC, false>(a, pfo, nt, s) : launch_fk_pipe<FPW, EPL, false, SRC, false>(a, pfo, nt, s);
}

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT>
static int launch_fk(const FkArgs &a, hipStream_t s) {
    return a.pad ? launch_fk_p<FPW, VEC, PFO, SRC, QOUT, true>(a, s) : launch_fk_p<FPW, VEC, PFO, SRC, QOUT, false>(a, s);
}

#ifdef PM_TUNING
constexpr bool kAllFkShapes = true;   // PM_FK_FPW can force any shape on any variant
#else
constexpr bool kAllFkShapes = false;  // only the (shape, variant) pairs dispatch_fk can pick are compiled
#endif

template <int FPW, int SRC>
static int dispatch_fk2(const FkArgs &a, bool vec, bool pfo, hipStream_t s) {
    const bool qout = a.quat_out != nullptr;
#define PM_FK_CASE(V, P, Q) \
    if (vec == V && pfo == P && qout == Q) return launch_fk<FPW, V, P, SRC, Q>(a, s);
    constexpr bool WITH_PFO = FPW <= 5 || kAllFkShapes;  // per-frame offsets always take the twelve-lane shape (dispatch_fk)
    PM_FK_CASE(true, false, false)
    PM_FK_CASE(false, false, false)
    if constexpr (WITH_PFO) {
        PM_FK_CASE(true, true, false)
        PM_FK_CASE(false, true, false)
    }
    if constexpr (SRC == SRC_O6D) {
        PM_FK_CASE(true, false, true)
        PM_FK_CASE(false, false, true)
        if constexpr (WITH_PFO) {
            PM_FK_CASE(true, true, true)
            PM_FK_CASE(false, true, true)
        }
    }
#undef PM_FK_CASE
    set_error("fk: no kernel variant");
    return PM_EUNSUPPORTED;
}

// Frames per wave (FPW, a multiple of 4 so that every tile base stays 16-byte aligned for any J).
// The LDS image (48 J B per frame) bounds residency, and with too few waves per CU nothing hides the walk's
// latency; fewer frames per wave leave walk lanes idle.  Four shapes cover the range (measured at 2^19 frames, % of
// 8 TB/s on the 64 J + 12 B of a frame; at 2^18 x 52: FPW 20/16/12/8 with three lanes per frame 376/311/307/253 us,
// FPW 4 with twelve lanes per frame 178 us; FPW 2 / 5: 227 / 176 us):
//   FPW 20, 3 lanes per frame   while 10 tiles fit a CU's LDS (J <= 16;  J = 8 / 12: 64 / 73 % against 60 / 70 % for FPW 16),
//   FPW 16                      while 7 fit (J <= 29;  J = 20 / 22 / 23 / 24 / 26 / 28: 70 / 72 / 68 / 65 / 70 / 66 % against
//                               68 / 66 / 65 / 60 / 56 / 51 % for FPW 20; 2^20 x 22: 262.8 us against 270.5 us),
//   FPW 12                      while 8 fit and the frames do not alias (J <= 34, J != 32;  J = 30 / 31 / 33 / 34:
//                               65 / 62 / 64 / 64 % against 62 / 61 / 60 / 61 % for the quad shape; J = 36: 58.5 against 60.7 %),
//   FPW 4, 12 lanes per frame (tree_walk_quad, pipelined tiles) beyond that.
// Frame padding against LDS bank aliasing (FkArgs::pad) depends on the shape: the frames of the three-lane walk
// alias when 9 J is a multiple of 8 floats (J % 8 == 0), the 4 frames of the quad walk sit 8 banks apart then and only
// alias when it is a multiple of 16 (J % 16 == 0; J = 40: 261 us without, 272 us with the padding).
template <int SRC>
static int dispatch_fk(const FkArgs &a_in, bool vec, bool pfo, hipStream_t s) {
    FkArgs a = a_in;
    const int extras = (pfo ? 3 : 0) + (a.quat_out ? 4 : 0), nreg = pfo ? 3 : 2;
    auto frame_bytes = [&](const int pad) { return ((size_t)a.J * (12 + extras) + (size_t)pad * nreg) * sizeof(float); };
    const size_t fixed = 4 * ((size_t)a.J + 4) * sizeof(float) + 256;
    const int pad3 = (a.J % 8 == 0) ? 4 : 0, pad12 = (a.J % 16 == 0) ? 4 : 0;
    auto tiles = [&](const int fpw) { return kMaxLds / ((size_t)fpw * frame_bytes(pad3) + fixed); };
    // (round 5) EIGHT frames per wave with a quad per frame (tree_walk_q4: half the wave walks, ~18 instructions a joint against ~15 per joint
    // for the four frames of the twelve-lane walk) where the 16- / 12-frame tiles no longer fit and the walk is still short; one box, % of the
    // HBM spec, eight-frame tile / what ran before: quaternion source J = 32 67.7 / 61.2, 36 66.5 / 61.0 (40: 61.5 / 64.4, 48: 57 / 63);
    // ortho6d source J = 16 69.0 / 65.0, 22 67.3 / 57.3, 28 70.0 / 64.1, 32 68.8 / 64.0, 34 65.2 / 54.2, 36 65.5 / 64.5 (40: 60.3 / 63.5),
    // with the quaternion output J = 16 67.7 / 58.4, 22 65.8 / 59.8, 28 65.8 / 64.4, 31 64.0 / 59.9 (32: 64.6 / 65.8)
    // (profiles/r05_fk_q4_shapes.txt).
    int pick = 4;
    if constexpr (SRC == SRC_QUAT) {
        // (round 5: twenty frames walk three lanes per frame; sixteen with a quad per frame read J = 8 / 12 / 14 / 15 / 16 70.5 / 76.4 / 76.0 /
        // 74.4 / 69.3 % against 67.5 / 74.8 / 73.7 / 70.9 / 68.2 on one box -- the twenty-frame tile is kept below eight joints only)
        if (tiles(20) >= 10 && a.J < 8) pick = 20;
        else if (tiles(16) >= 7) pick = 16;
        else if (tiles(12) >= 8 && a.J % 8 != 0) pick = 12;
        else if (a.J <= kFkEightFramesMaxJ) pick = 8;
    } else {
        if (a.J < 8 && tiles(20) >= 6) pick = 20;
        else if (a.J < 16) pick = 16;  // (J = 8 / 12 / 14 / 15, sixteen / twenty frames: 68.4 / 72.4 / 73.4 / 72.0 against 65.8 / 71.8 / 68.4 / 67.0 %; with the quaternion output 67.4 / 72.6 / 72.6 / 70.4 against 65.0 / 70.7 / 64.9 / 65.5)
        else if (a.J <= (a.quat_out ? kFkEightFramesMaxJO6dQ : kFkEightFramesMaxJO6d)) pick = 8;
    }
    // Per-frame offsets (a bigger image) do better on the twelve-lane shape at any joint count (2^20 x 22: 422 us three-lane, 434 with
    // FPW 16, 358 us twelve-lane).
    if (pfo) pick = 4;
    pick = tune_env("PM_FK_FPW", pick);  // PM_TUNING build only: 20, 16, 12, 8 or 4
    // (round 5) four frames a wave and four JOINTS of a frame at a time (fk_pipe_kernel with tree_walk_w4) when the tree is wide enough for that
    // to take fewer instructions than one joint at a time: ~40 a step for sixteen joint-frames against ~15 for four (twelve lanes a frame) or ~18 for
    // eight (a quad a frame), i.e. step lists of at most 0.32 J steps -- SMPL-H's 52 joints: 15.  One box, humanoids with hands / random trees, % of
    // the HBM spec, this walk / what ran before (profiles/r05_fk_w4_sweep.txt): J = 40 66.9 / 63.4 and 72.2 / 63.4, 48 70.2 / 63.1 and 74.6 / 66.1, 52
    // (SMPL-H) 72.8 / 64.3 and 74.0 / 64.7, 64 67.8 / 54.1 and 69.2 / 58.8, 80 68.9 / 56.4 and 70.1 / 56.9, 92 70.9 / 57.6 and 71.5 / 57.6, 100 66.7 /
    // 57.5 and 68.8 / 62.7; random trees of 24...36 joints 64-70 / 56-62 (the sixteen- / eight-frame tiles keep the humanoids there: equal);
    // chains lose 3-6 points and never qualify.  From 101 joints on fk_wide_kernel is the better shape (112: 62.9 / 58.7, 128: 73.5 / 62.6).
    // PM_FK_W4 (PM_TUNING build only): 0 never, 1 whenever the list holds the tree (and the four-frame kernel is what runs).
    a.wsteps = 0;
    memset(a.wjobs, 0, sizeof(a.wjobs));  // (the kernarg copy takes the whole struct: no uninitialised words in it)
    if (const int w4 = tune_env("PM_FK_W4", -1); w4 != 0 && a.J <= 128 && (w4 == 1 || a.J >= (pfo ? 8 : (SRC == SRC_QUAT ? 24 : 30)))) {  // (per-frame offsets take the four-frame kernel at any joint count: the 22-joint body 61.2 -> 64.4 %)
        uint32_t list[(kW4Steps + 2) * 4];
        const int n = fk_wide_plan(a.parents, a.J, 4, kW4Steps, true, list);
        if (n > 0 && (w4 == 1 || 100 * n <= 32 * a.J)) {
            a.wsteps = n;
            for (int k = 0; k < 4; ++k)
                for (int st = 0; st < kW4Stride; ++st) a.wjobs[k * kW4Stride + st] = st < n ? list[st * 4 + k] : 0u;
            if (w4 != 1) pick = 4;
        }
    }
    // (the ortho6d source with per-frame offsets and quaternions: eight records a lane AND nineteen floats a joint in LDS from 93 joints on)
    const bool all_three = SRC == SRC_O6D && pfo && a.quat_out != nullptr;
    const int wide_min = all_three ? 92 : kFkWideMinJ;
    const bool w4_first = a.wsteps > 0 && a.J <= wide_min;  // (beyond: the wave-per-frame walk first; tree_walk_w4 still serves what it declines)
#ifdef PM_TUNING
    if constexpr (SRC == SRC_QUAT) {  // the three-lane tile, several tiles per workgroup with the next tile's records prefetched into registers
        const int pnt = tune_env("PM_FK_PIPE3", 0);
        if (pnt > 0 && !pfo) {
            a.pad = pad3;
            if (pick == 16) a.fmap = q4_frame_map(a.J, a.pad);
            if (pick == 20 && a.J <= 22) return dispatch_fk_pipe<20, 7, SRC>(a, vec, pfo, pnt, s);
            if (pick == 16 && a.J <= 24) return dispatch_fk_pipe<16, 6, SRC>(a, vec, pfo, pnt, s);
            if (pick == 12 && a.J <= 26) return dispatch_fk_pipe<12, 5, SRC>(a, vec, pfo, pnt, s);
            if (pick == 16 && a.J <= 52) return dispatch_fk_pipe<16, 13, SRC>(a, vec, pfo, pnt, s);  // the whole 40 KB tile, three waves a CU, software-pipelined
            if (pick == 12 && a.J <= 53) return dispatch_fk_pipe<12, 10, SRC>(a, vec, pfo, pnt, s);
            if (pick == 8 && a.J <= 40) return dispatch_fk_pipe<8, 5, SRC>(a, vec, pfo, pnt, s);
            if (pick == 8 && a.J <= 56) return dispatch_fk_pipe<8, 7, SRC>(a, vec, pfo, pnt, s);
            if (pick == 8 && a.J <= 64) return dispatch_fk_pipe<8, 8, SRC>(a, vec, pfo, pnt, s);
        }
    }
#endif
    if (pick != 20 && pick != 16 && pick != 12 && pick != 8 && pick != 4) { set_error("PM_FK_FPW must be 20, 16, 12, 8 or 4"); return PM_EINVAL; }
    if ((size_t)pick * frame_bytes(pad3) + fixed > kMaxLds) pick = 4;
    a.pad = (pick == 4) ? pad12 : pad3;
    if (pick == 16) a.fmap = q4_frame_map(a.J, a.pad);
    const size_t per_frame = frame_bytes(a.pad);
    // wide trees from 93 joints on (fkwide.hip: a wave per frame, its lanes over the joints of a host-made step list), FIRST when the list keeps
    // the quads busy -- at most 2.5 quad-steps per joint: random trees (parents[j] uniform in [0, j)) 59-70 % of the HBM spec at J = 96...128 where
    // the pipelined tile kernel reads 46-56 %, 63-73 % at J = 129...512 where the streamed walk declines them and the four-frame tiles read 9-34 %
    // (profiles/r05_fk_wide_sweep.txt); deep, narrow trees go to the streamed walk below.  PM_FK_WIDE (PM_TUNING build only): 0 never, 1 from any
    // joint count and any list.
    const int wd = tune_env("PM_FK_WIDE", -1);
    const bool wide_plain = SRC == SRC_QUAT && !pfo && vec && a.quat_out == nullptr;
    // long skeletons: the streamed three-lane walk (fk_stream_kernel) where the topology's cross-chunk branch points fit its register
    // slots; PM_FK_STREAM (PM_TUNING build only): 0 never, 1 from any joint count
    const int st = tune_env("PM_FK_STREAM", -1);
    // (and only with the joint-frames to fill the chip: sixteen frames to a wave that walks all J joints -- chain-like J = 64 / 128 / 256 at
    // 2^10 frames: 16 / 28 / 54 us against 9 / 15 / 26 us for the four-frame tiles; the crossovers sit at F J = 0.7-1.0 M)
    const bool stream_ok = SRC == SRC_QUAT && !pfo && vec && a.quat_out == nullptr && st != 0 &&
                           (st == 1 || (fk_stream_wanted(a.J) && lane_per_frame_pays(a.F, a.J, kFkStreamMinJointFrames)));
    // (whole-line rows of 96 / 128 joints: the streamed walk first -- a humanoid with hands reads 59.6 / 62.8 % there against 55.8 / 61.3 %)
    const bool stream_first = a.J % 32 == 0 && a.J <= 128 && wd != 1;
    // the other sources at 101...128 joints: where the four-frame image is padded (multiples of sixteen joints) and for the ortho6d source with
    // per-frame offsets and quaternions (eight records a lane AND nineteen floats a joint in LDS: 47-51 % on the pipelined tiles).  Random
    // trees / a body with hands, 2^18 frames, wide / pipelined tiles: per-frame offsets J = 112 62.9 / 61.2, 128 66.1 / 51.6 (104: 63.6 / 71.1,
    // 120: 59.6 / 63.6); ortho6d 112 66.0 / 62.9, 128 66.7 / 62.4 (104: 64.3 / 68.8); with quaternions 112 66.4 / 61.4, 128 65.5 / 61.1; all
    // three 96 65.0 / 46.8, 100 60.2 / 38-47, 104 66.4 / 48.4, 112 69.5 / 49.3, 120 60.5 / 51.3, 128 66.1 / 50.8 (profiles/r05_fk_wide_variants_mid.txt)
    const bool wide_mid = !wide_plain && vec && a.J > wide_min && a.J <= 128 && (a.J % 16 == 0 || all_three);
    auto wide = [&](const int bound, int &rc) {
        return try_fk_wide(SRC == SRC_QUAT ? 0 : 1, a.src, a.root_pos, a.offsets, pfo, a.pos, a.rotmats, a.quat_out, a.eps, a.F, a.J, a.depth, a.parents,
                           a.ablate, bound, s, rc);
    };
    if (!w4_first) {
        int rc = PM_OK;
        if constexpr (SRC == SRC_QUAT)
            if (stream_ok && stream_first && try_fk_stream(a, s, rc)) return rc;
        if ((wide_plain || wide_mid) && wd != 0 && (wd == 1 || a.J > wide_min) && wide(wd == 1 ? 0 : 25, rc)) return rc;
        if constexpr (SRC == SRC_QUAT)
            if (stream_ok && !stream_first && try_fk_stream(a, s, rc)) return rc;
        // whatever is left beyond 128 joints -- trees the streamed walk declined, the ortho6d source, per-frame offsets: the wide walk if its
        // step list holds the tree at all (the four-frame tiles that are left read 9-34 % there)
        if (vec && wd != 0 && (a.J > 128 || (wd == 1 && !wide_plain)) && wide(0, rc)) return rc;
    }
    if (pick == 4 && a.J <= 128) {
        // mid-size skeletons: registers-first phase A and tiles pipelined inside a workgroup (fk_pipe_kernel; 4 records
        // per lane up to 64 joints, 8 up to 128).  Measured at 2^18 x 52: fused ortho6d 224 us (fk_kernel) -> 188 / 183 /
        // 187 / 198 us with 1 / 2 / 4 / 8 tiles per workgroup; the same structure on the 20-frame tile of J = 22 is
        // slower (293 vs 270 us).
        // (beyond 64 joints -- eight records per lane, two waves per SIMD -- four tiles: J = 66 / 96 / 128 at 2^19 frames 589 / 760 / 1209 us
        // with two, 555 / 725 / 1165 us with four)
        int nt = ((a.F + 3) / 4 >= 16384) ? (a.J > 64 ? 4 : 2) : 1;
        nt = tune_env("PM_FK_NT", nt);  // PM_TUNING build only: tiles per workgroup, 0 = fk_kernel
        // (two records per lane cover 4 x 32 joints: J = 28 / 30 / 32 at 2^19 frames 205 / 215 / 230 us with four, 195 / 203 / 209 us with two)
        if (nt > 0 && a.J <= 32 && tune_env("PM_FK_EPL2", 1)) return dispatch_fk_pipe<4, 2, SRC>(a, vec, pfo, nt, s);
        if (nt > 0 && a.J <= 64) return dispatch_fk_pipe<4, 4, SRC>(a, vec, pfo, nt, s);
        // (six records per lane cover 4 x 96 joints in 142-150 VGPRs -- three waves per SIMD -- where eight take 214-222: chain-like
        // skeletons at 2^19 frames, eight / six records: J = 65 549 / 457 us, 72 592 / 507, 80 634 / 585, 88 - / 667, 96 718 / 732)
        if (nt > 0 && a.J <= 92 && tune_env("PM_FK_EPL6", 1)) return dispatch_fk_pipe<4, 6, SRC>(a, vec, pfo, nt, s);
        if (nt > 0) return dispatch_fk_pipe<4, 8, SRC>(a, vec, pfo, nt, s);
    }
    // (The pipelined multi-tile structure on the 20-frame three-lane tile, measured again in round 2 with the current kernels:
    // 257 us against 270 us on one box, 295-310 us against 271 us on the next -- fewer, longer workgroups cannot absorb
    // per-XCD bandwidth differences the way one-tile workgroups under hardware dispatch do.  Not used.)
    switch (pick) {
        case 20: return dispatch_fk2<20, SRC>(a, vec, pfo, s);
        case 8: return dispatch_fk2<8, SRC>(a, vec, pfo, s);
        case 16: return dispatch_fk2<16, SRC>(a, vec, pfo, s);
        case 12: if constexpr (SRC == SRC_QUAT || kAllFkShapes) return dispatch_fk2<12, SRC>(a, vec, pfo, s); else break;
        case 4: if (4 * per_frame + fixed <= kMaxLds) return dispatch_fk2<4, SRC>(a, vec, pfo, s);
    }
    if constexpr (SRC == SRC_O6D) {
        if (a.quat_out) {
            // 511 / 512 joints with per-frame offsets AND the quaternion output on a tree the wide walk's step list does not hold: four frames
            // of 19 floats a joint are 164 KB.  The reference's own two steps instead (rotations/ortho6d.py:50-64, then ops/skeleton.py:13-61):
            // the quaternions go to the caller's array and fk reads them back
            if (int e = pm_o6d_to_quat_f32(a.src, a.F * a.J, a.eps, a.quat_out, s)) return e;
            FkArgs b = a_in;
            b.src = a.quat_out; b.quat_out = nullptr;
            return dispatch_fk<SRC_QUAT>(b, vec, pfo, s);
        }
    }
    set_error("fk: J=%d does not fit the LDS tile", a.J);
    return PM_EUNSUPPORTED;
}

static int fk_common(int src_kind, const float *src, const float *root_pos, const float *offsets,
                     int offsets_per_frame, const int32_t *parents, int64_t F, int32_t J, float eps,
                     float *pos, float *rotmats, float *quat_out, pm_stream_t stream) {
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "fk: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(src && root_pos && offsets && parents && pos && rotmats, "fk: null pointer");
    FkArgs a;
    a.src = src; a.root_pos = root_pos; a.offsets = offsets; a.pos = pos; a.rotmats = rotmats;
    a.quat_out = quat_out; a.F = F; a.J = J; a.eps = eps; a.pad = 0;  // set by dispatch_fk, per walk shape
    a.wsteps = 0;
    a.fmap = 0xfedcba9876543210ull;
    a.xchunk = tune_env("PM_FK_XCHUNK", kXcdChunk); a.pad2_ = 0;  // tiles (fk_kernel) / tile groups (fk_pipe_kernel) per XCD chunk, see xcd_tile_chunked; PM_TUNING build only: 0 = rounds 1-5's contiguous eighths
#ifdef PM_TUNING
    a.times = nullptr;
    if (const char *e = getenv("PM_FK_TIMES_PTR")) a.times = reinterpret_cast<uint64_t *>(strtoull(e, nullptr, 0));
#endif
    a.ablate = tune_env("PM_FK_ABLATE", 0);
    if (int e = pack_parents(parents, J, a.parents)) return e;
    {
        int dep[PM_MAX_JOINTS];
        a.depth = 0;
        for (int32_t j = 0; j < J; ++j) {
            dep[j] = (j == 0) ? 0 : dep[a.parents.p[j]] + 1;
            if (dep[j] > a.depth) a.depth = dep[j];
        }
    }
    const bool vec = aligned16(src) && aligned16(pos) && aligned16(rotmats) &&
                     (!offsets_per_frame || aligned16(offsets)) && (!quat_out || aligned16(quat_out));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (src_kind == SRC_QUAT) return dispatch_fk<SRC_QUAT>(a, vec, offsets_per_frame != 0, s);
    return dispatch_fk<SRC_O6D>(a, vec, offsets_per_frame != 0, s);
}

}  // namespace pm

extern "C" int pm_fk_f32(const float *rot, const float *root_pos, const float *offsets, int offsets_per_frame,
                         const int32_t *parents, int64_t F, int32_t J, float *pos, float *rotmats,
                         pm_stream_t stream) {
    return pm::fk_common(pm::SRC_QUAT, rot, root_pos, offsets, offsets_per_frame, parents, F, J, 0.0f, pos,
                         rotmats, nullptr, stream);
}

extern "C" int pm_fk_from_ortho6d_f32(const float *o6d, const float *root_pos, const float *offsets,
                                      int offsets_per_frame, const int32_t *parents, int64_t F, int32_t J,
                                      float eps, float *pos, float *rotmats, float *quat_out,
                                      pm_stream_t stream) {
    return pm::fk_common(pm::SRC_O6D, o6d, root_pos, offsets, offsets_per_frame, parents, F, J, eps, pos,
                         rotmats, quat_out, stream);
}
