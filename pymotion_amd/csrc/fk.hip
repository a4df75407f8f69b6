// fk.hip -- batched forward kinematics for gfx950 (reference: pymotion/ops/skeleton.py:16-61).
//
// Mapping ("row-parallel tree walk").  A world transform G_j = [R_j | p_j] obeys
//     G_j = G_parent(j) . [L_j | t_j]      =>      row r of G_j = (row r of R_parent) . [L_j | t_j] + [0 | p_parent[r]]
// i.e. the three rows of a frame's transforms never mix.  THREE LANES own one frame (lane = 3*f + r,
// r = row), so a 64-lane wave walks FPW = 20 skeletons at once (60 lanes), each lane carrying just the
// 4 floats of its row of the previous joint's transform in registers.  Joints are visited in index
// order (parents[j] < j); when parents[j] == j-1 -- the common case in DFS-ordered skeletons -- the
// parent row is already in registers, otherwise it is re-read from the LDS output tile.  The branch is
// wave-uniform (topology is shared by all frames) and `parents` arrives in the kernarg segment, so it
// costs scalar instructions only.
//
// Data movement per wave-tile (J joints, FPW frames): rot tile (FPW*J*16 B, contiguous in HBM) is
// loaded with dwordx4 per lane into LDS; each step reads one quaternion per frame from LDS
// (ds_read_b128, broadcast inside the lane triple), normalises it, builds L_j and emits 3+1 floats per
// lane into the LDS images of `rotmats` / `pos`, which are laid out EXACTLY like the HBM outputs.  The
// finished images leave with contiguous dwordx4 streaming stores.  Algorithmic HBM bytes per frame:
// 16J + 12 read, 48J written -- nothing is read or written twice.
#include "common.hpp"

namespace pm {

enum FkSrc { SRC_QUAT = 0, SRC_O6D = 1 };

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
    Parents parents;
};

template <int SRC>
constexpr int src_width() { return SRC == SRC_QUAT ? 4 : 6; }

// LDS floats per frame-joint: pos 3 + rot 9 + source + (per-frame offsets 3) + (quat_out 4)
template <int SRC, bool PFO, bool QOUT>
constexpr int fk_lds_floats() { return 12 + src_width<SRC>() + (PFO ? 3 : 0) + (QOUT ? 4 : 0); }

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT>
__global__ __launch_bounds__(PM_WAVE) void fk_kernel(const FkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SW = src_width<SRC>();
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile(ntiles);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int FJ = FPW * J;

    float *sRot = smem;                 // [FPW*J*9]   16B-aligned: FPW*J*9*4 with FPW % 4 == 0
    float *sPos = sRot + FJ * 9;        // [FPW*J*3]
    float *sSrc = sPos + FJ * 3;        // [FPW*J*SW]
    float *sOff = sSrc + FJ * SW;       // [FPW*J*3]   (PFO)
    float *sQo = sOff + (PFO ? FJ * 3 : 0);  // [FPW*J*4] (QOUT)

    tile_load<VEC>(a.src + f0 * J * SW, sSrc, nf * J * SW, lane);
    if (PFO) tile_load<VEC>(a.offsets + f0 * J * 3, sOff, nf * J * 3, lane);

    const int f = lane / 3;
    const int r = lane - 3 * f;
    const bool act = f < nf;  // lanes 3*FPW.. and frames past F idle
    const int fc = act ? f : 0;
    const float gp = act ? a.root_pos[f0 * 3 + lane] : 0.0f;
    wave_sync();

    // row r of the transform of joint j-1 (registers) -- seeded so that joint 0 falls out of the
    // same formula: e_r . L = row r of L (exact: 1*x + 0*y + 0*z), translation = root_pos[r].
    float g0 = (r == 0) ? 1.0f : 0.0f, g1 = (r == 1) ? 1.0f : 0.0f, g2 = (r == 2) ? 1.0f : 0.0f, gt = gp;
    const float *fSrc = sSrc + fc * J * SW;
    float *fRot = sRot + fc * J * 9 + r * 3;
    float *fPos = sPos + fc * J * 3 + r;

    // Skeleton constants for joint j are fetched one iteration ahead with scalar loads (kernarg
    // `parents`, global `offsets`), so neither sits on the per-joint dependency chain.
    int par_n = -1;  // joint 0: "parent" = the seed above
    float t0n = 0.0f, t1n = 0.0f, t2n = 0.0f;
    for (int j = 0; j < J; ++j) {
        const int par = par_n;
        float t0 = t0n, t1 = t1n, t2 = t2n;
        {
            const int jn = (j + 1 < J) ? j + 1 : j;
            par_n = a.parents.p[jn];
            if (!PFO) { t0n = a.offsets[3 * jn]; t1n = a.offsets[3 * jn + 1]; t2n = a.offsets[3 * jn + 2]; }
        }
        float q[4], L[9];
        if constexpr (SRC == SRC_QUAT) {
            float qi[4];
            lds_get<4>(fSrc, j, qi);
            qnormalize(qi, 1e-8f, q);  // skeleton.py:45 normalises inside fk
        } else {
            // rotations/ortho6d.py:50-64 : 6D -> matrix -> quaternion (itself normalised), then fk's
            // own normalise, exactly the chain ortho6d.to_quat -> fk of the reference.
            float x[6], m[9], qi[4];
            lds_get<6>(fSrc, j, x);
            o6d2m(x, a.eps, m);
            m2q(m, qi);
            if (QOUT && act && r == 0) lds_put<4>(sQo, fc * J + j, qi);
            qnormalize(qi, 1e-8f, q);
        }
        q2m(q, L);

        float p0 = g0, p1 = g1, p2 = g2, pt = gt;
        if (par != j - 1) {  // wave-uniform: not the previous joint -> its row is in the LDS image
            p0 = fRot[par * 9]; p1 = fRot[par * 9 + 1]; p2 = fRot[par * 9 + 2];
            pt = fPos[par * 3];
        }
        if (PFO && j > 0) {
            const float *o = sOff + (fc * J + j) * 3;
            t0 = o[0]; t1 = o[1]; t2 = o[2];
        }
        g0 = p0 * L[0] + p1 * L[3] + p2 * L[6];
        g1 = p0 * L[1] + p1 * L[4] + p2 * L[7];
        g2 = p0 * L[2] + p1 * L[5] + p2 * L[8];
        gt = p0 * t0 + p1 * t1 + p2 * t2 + pt;
        if (act) {
            fRot[j * 9] = g0; fRot[j * 9 + 1] = g1; fRot[j * 9 + 2] = g2;
            fPos[j * 3] = gt;
        }
    }
    wave_sync();
    tile_store<VEC>(a.rotmats + f0 * J * 9, sRot, nf * J * 9, lane);
    tile_store<VEC>(a.pos + f0 * J * 3, sPos, nf * J * 3, lane);
    if (QOUT) tile_store<VEC>(a.quat_out + f0 * J * 4, sQo, nf * J * 4, lane);
}

template <int FPW, bool VEC, bool PFO, int SRC, bool QOUT>
static int launch_fk(const FkArgs &a, hipStream_t s) {
    const size_t lds = (size_t)FPW * a.J * fk_lds_floats<SRC, PFO, QOUT>() * sizeof(float);
    auto k = fk_kernel<FPW, VEC, PFO, SRC, QOUT>;
    if (int e = allow_lds(k, lds)) return e;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) {
        set_error("fk: %lld tiles exceed the grid limit", (long long)grid);
        return PM_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    return check_hip(hipGetLastError(), "fk launch");
}

template <int FPW, int SRC>
static int dispatch_fk2(const FkArgs &a, bool vec, bool pfo, hipStream_t s) {
    const bool qout = a.quat_out != nullptr;
#define PM_FK_CASE(V, P, Q) \
    if (vec == V && pfo == P && qout == Q) return launch_fk<FPW, V, P, SRC, Q>(a, s);
    PM_FK_CASE(true, false, false)
    PM_FK_CASE(true, true, false)
    PM_FK_CASE(false, false, false)
    PM_FK_CASE(false, true, false)
    if constexpr (SRC == SRC_O6D) {
        PM_FK_CASE(true, false, true)
        PM_FK_CASE(true, true, true)
        PM_FK_CASE(false, false, true)
        PM_FK_CASE(false, true, true)
    }
#undef PM_FK_CASE
    set_error("fk: no kernel variant");
    return PM_EUNSUPPORTED;
}

// Frames per wave: 20 (60 lanes busy; 20*J*{3,9,4,6} floats are all multiples of 4, which keeps every
// tile base 16-byte aligned for any J) while at least two tiles fit a CU's LDS, else 8, 4.
template <int SRC>
static int dispatch_fk(const FkArgs &a, bool vec, bool pfo, hipStream_t s) {
    const size_t per_frame =
        (size_t)a.J * (12 + src_width<SRC>() + (pfo ? 3 : 0) + (a.quat_out ? 4 : 0)) * sizeof(float);
    if (20 * per_frame <= kMaxLds / 2) return dispatch_fk2<20, SRC>(a, vec, pfo, s);
    if (8 * per_frame <= kMaxLds / 2) return dispatch_fk2<8, SRC>(a, vec, pfo, s);
    if (4 * per_frame <= kMaxLds) return dispatch_fk2<4, SRC>(a, vec, pfo, s);
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
    a.quat_out = quat_out; a.F = F; a.J = J; a.eps = eps;
    if (int e = pack_parents(parents, J, a.parents)) return e;
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
