// dq.hip -- root-centred dual-quaternion encode / decode and from_global_rotations for gfx950.
//   to_root_dual_quat    pymotion/ops/skeleton.py:207-244   (sequential along each parent chain)
//   from_root_dual_quat  pymotion/ops/skeleton.py:173-204   (every joint needs only its parent's INPUT)
//   from_global_rotations pymotion/ops/skeleton.py:64-93    (same gather shape)
// Same wave-private tiling as fk.hip: HBM <-> LDS with contiguous dwordx4, AoS access from LDS.
#include "common.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------------
// to_root_dual_quat: a quaternion payload does not split by rows, so ONE LANE walks ONE frame
// (FPW frames per wave, lanes >= FPW idle).  State per lane: the previous joint's root-space (q, t)
// in registers; other parents are re-read from the LDS image of the output.  LDS per frame-joint:
// 16 B staged input quaternion + 32 B output dual quaternion; the root-space (q,t) of a joint is
// recovered from that image when needed as a parent (q = dq[0:4]; t kept in a 12 B side image).
// ---------------------------------------------------------------------------------------------------
struct ToRootArgs {
    const float *rot;       // [F,J,4]
    const float *root_pos;  // [F,3]
    const float *offsets;   // [J,3]
    float *dq;              // [F,J,8]
    int64_t F;
    int32_t J;
    Parents parents;
};

template <int FPW, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void to_root_dq_kernel(const ToRootArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile(ntiles);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int FJ = FPW * J;
    float *sDq = smem;            // [FPW*J*8]  image of the output tile
    float *sQ = sDq + FJ * 8;     // [FPW*J*4]  staged input
    float *sT = sQ + FJ * 4;      // [FPW*J*3]  root-space translations (parents' lookup)

    tile_load<VEC>(a.rot + f0 * J * 4, sQ, nf * J * 4, lane);
    const bool act = lane < nf;
    const int fc = act ? lane : 0;
    float rp[3] = {0.0f, 0.0f, 0.0f};
    if (act) {
        rp[0] = a.root_pos[(f0 + lane) * 3];
        rp[1] = a.root_pos[(f0 + lane) * 3 + 1];
        rp[2] = a.root_pos[(f0 + lane) * 3 + 2];
    }
    wave_sync();

    float gq[4] = {1.0f, 0.0f, 0.0f, 0.0f}, gt[3] = {0.0f, 0.0f, 0.0f};  // joint j-1, root space
    for (int j = 0; j < J; ++j) {
        float q[4], t[3];
        lds_get<4>(sQ, fc * J + j, q);
        const int par = (j == 0) ? 0 : a.parents.p[j];
        if (j == 0) {
            t[0] = rp[0]; t[1] = rp[1]; t[2] = rp[2];  // skeleton.py:232
        } else {
            t[0] = a.offsets[3 * j]; t[1] = a.offsets[3 * j + 1]; t[2] = a.offsets[3 * j + 2];
            if (par != 0) {  // skeleton.py:236-241 ; joints hanging off the root stay local
                float pq[4], pt[3];
                if (par == j - 1) {
                    pq[0] = gq[0]; pq[1] = gq[1]; pq[2] = gq[2]; pq[3] = gq[3];
                    pt[0] = gt[0]; pt[1] = gt[1]; pt[2] = gt[2];
                } else {
                    lds_get<4>(sDq, (fc * J + par) * 2, pq);  // real part of the parent's dq
                    lds_get<3>(sT, fc * J + par, pt);
                }
                float tv[3], qq[4];
                qmulvec(pq, t, tv);
                t[0] = tv[0] + pt[0]; t[1] = tv[1] + pt[1]; t[2] = tv[2] + pt[2];
                qmul(pq, q, qq);
                q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
            }
        }
        float d[8];
        rt2dq(q, t, d);
        if (act) {
            lds_put<8>(sDq, fc * J + j, d);
            lds_put<3>(sT, fc * J + j, t);
        }
        gq[0] = q[0]; gq[1] = q[1]; gq[2] = q[2]; gq[3] = q[3];
        gt[0] = t[0]; gt[1] = t[1]; gt[2] = t[2];
    }
    wave_sync();
    tile_store<VEC>(a.dq + f0 * J * 8, sDq, nf * J * 8, lane);
}

template <int FPW>
static int launch_to_root(const ToRootArgs &a, bool vec, hipStream_t s) {
    const size_t lds = (size_t)FPW * a.J * 15 * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("to_root_dq: grid too large"); return PM_EUNSUPPORTED; }
    if (vec) {
        auto k = to_root_dq_kernel<FPW, true>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    } else {
        auto k = to_root_dq_kernel<FPW, false>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    }
    return check_hip(hipGetLastError(), "to_root_dq launch");
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
constexpr int gather_lds_w() { return MODE == 0 ? 8 + 3 + 4 : 4 + 4; }

template <int MODE, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void gather_parent_kernel(const GatherArgs a, const int fpw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IW = gather_in_w<MODE>();
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + fpw - 1) / fpw;
    const int64_t tile = xcd_tile(ntiles);
    if (tile < 0) return;
    const int64_t f0 = tile * fpw;
    const int nf = (int)((a.F - f0) < fpw ? (a.F - f0) : fpw);
    const int FJ = fpw * J;
    float *sIn = smem;             // [fpw*J*IW]
    float *sO0 = sIn + FJ * IW;    // [fpw*J*3] or [fpw*J*4]
    float *sO1 = sO0 + FJ * (MODE == 0 ? 3 : 4);

    tile_load<VEC>(a.in + f0 * J * IW, sIn, nf * J * IW, lane);
    wave_sync();
    const int n = nf * J;
    for (int e = lane; e < n; e += PM_WAVE) {
        const int f = e / J, j = e - f * J;
        const int par = (j == 0) ? 0 : a.parents.p[j];
        if constexpr (MODE == 0) {
            float d[8], q[4], t[3];
            lds_get<8>(sIn, e, d);
            dq2rt(d, q, t);  // dual_quat.py:75-83
            if (j != 0 && par != 0) {  // skeleton.py:194-203, parent still in root space
                float pd[8], pq[4], pt[3];
                lds_get<8>(sIn, f * J + par, pd);
                dq2rt(pd, pq, pt);
                const float inv[4] = {pq[0], -pq[1], -pq[2], -pq[3]};
                const float dv[3] = {t[0] - pt[0], t[1] - pt[1], t[2] - pt[2]};
                float qq[4];
                qmulvec(inv, dv, t);
                qmul(inv, q, qq);
                q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
            }
            lds_put<3>(sO0, e, t);
            lds_put<4>(sO1, e, q);
        } else {
            float g[4], o[4];
            lds_get<4>(sIn, e, g);
            if (j == 0) {
                o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = g[3];
            } else {  // skeleton.py:85-91 : conj(global_parent) (x) global_child
                float pg[4];
                lds_get<4>(sIn, f * J + par, pg);
                const float inv[4] = {pg[0], -pg[1], -pg[2], -pg[3]};
                qmul(inv, g, o);
            }
            lds_put<4>(sO0, e, o);
        }
    }
    wave_sync();
    if constexpr (MODE == 0) {
        tile_store<VEC>(a.out0 + f0 * J * 3, sO0, n * 3, lane);
        tile_store<VEC>(a.out1 + f0 * J * 4, sO1, n * 4, lane);
    } else {
        tile_store<VEC>(a.out0 + f0 * J * 4, sO0, n * 4, lane);
    }
}

template <int MODE>
static int launch_gather(const GatherArgs &a, bool vec, hipStream_t s) {
    // frames per wave: multiple of 4 (16-byte tile bases), ~256 (frame,joint) items per wave
    const size_t per_frame = (size_t)a.J * gather_lds_w<MODE>() * sizeof(float);
    int fpw = (int)((256 + a.J - 1) / a.J);
    fpw = (fpw + 3) & ~3;
    while (fpw > 4 && fpw * per_frame > kMaxLds / 4) fpw -= 4;
    if (fpw * per_frame > kMaxLds) { set_error("gather: J too large for LDS"); return PM_EUNSUPPORTED; }
    const size_t lds = fpw * per_frame;
    const int64_t ntiles = (a.F + fpw - 1) / fpw;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("gather: grid too large"); return PM_EUNSUPPORTED; }
    if (vec) {
        auto k = gather_parent_kernel<MODE, true>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, fpw);
    } else {
        auto k = gather_parent_kernel<MODE, false>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a, fpw);
    }
    return check_hip(hipGetLastError(), "gather launch");
}

}  // namespace pm

extern "C" int pm_to_root_dq_f32(const float *rot, const float *root_pos, const int32_t *parents,
                                 const float *offsets, int64_t F, int32_t J, float *dq, pm_stream_t stream) {
    using namespace pm;
    PM_CHECK_ARGS(F >= 0 && J >= 1 && J <= PM_MAX_JOINTS, "to_root_dq: need F >= 0 and 1 <= J <= PM_MAX_JOINTS");
    if (F == 0) return PM_OK;
    PM_CHECK_ARGS(rot && root_pos && parents && offsets && dq, "to_root_dq: null pointer");
    ToRootArgs a;
    a.rot = rot; a.root_pos = root_pos; a.offsets = offsets; a.dq = dq; a.F = F; a.J = J;
    if (int e = pack_parents(parents, J, a.parents)) return e;
    const bool vec = aligned16(rot) && aligned16(dq);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t per_frame = (size_t)J * 15 * sizeof(float);
    if (32 * per_frame <= kMaxLds / 4) return launch_to_root<32>(a, vec, s);
    if (16 * per_frame <= kMaxLds / 2) return launch_to_root<16>(a, vec, s);
    if (4 * per_frame <= kMaxLds) return launch_to_root<4>(a, vec, s);
    set_error("to_root_dq: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
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
