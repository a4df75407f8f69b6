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
//     rot_j = from_to(offsets[c0], inv(G_pre) (P_c0 - P_j))                                   (:136-141)
//     for each further child gc:  G_j = G_pre (x) rot_j
//         rot_j = rot_j (x) from_to_axis(offsets[gc], inv(G_j)(P_gc - P_j), inv(G_j) normalize(P_c0 - P_j))   (:147-168)
//     G_j = G_pre (x) rot_j/(|rot_j| + 1e-8)  (fk normalises its inputs, skeleton.py:45; from_to's axis is only
//     unit up to its own eps);  joints without children keep the identity (:126-130).
// HBM traffic: 12 J B/frame in, 16 J out, both as coalesced one-record-per-lane streams.
#include <stdlib.h>

#include "common.hpp"

namespace pm {

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
    Topo16 topo;
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

template <int FPW, bool VEC>
__global__ __launch_bounds__(PM_WAVE) void from_root_positions_kernel(const IkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int J = a.J;
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t tile = xcd_tile(ntiles);
    if (tile < 0) return;
    const int64_t f0 = tile * FPW;
    const int nf = (int)((a.F - f0) < FPW ? (a.F - f0) : FPW);
    const int n = nf * J;
    const int FS = ik_frame_stride(J);
    float *sS = smem;                         // [FPW * FS]  slot (f, j): position, then world quaternion
    float *sOff = sS + FPW * FS;              // [J * 3]
    int *sTopo = reinterpret_cast<int *>(sOff + 3 * J);  // [J] parent | [J+1] cstart | [J] clist
    const float invJ = 1.0f / (float)J;

    // positions: one 12-byte record per lane, consecutive lanes on consecutive records (coalesced dwordx3),
    // four loads per lane in flight, re-packed into the 16-byte slots
    {
        const float *g = a.pos + f0 * J * 3;
        for (int e0 = 0; e0 < n; e0 += 4 * PM_WAVE) {
            v3f_a4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * PM_WAVE + lane, ec = e < n ? e : n - 1;
                p[u] = __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(g + 3 * ec));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * PM_WAVE + lane;
                const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
                const int j = e - f * J;
                if (e < n) { float *sl = sS + f * FS + 4 * j; sl[0] = p[u].x; sl[1] = p[u].y; sl[2] = p[u].z; }
            }
        }
    }
    for (int j = lane; j < J; j += PM_WAVE) {  // rest directions, normalised once (from_to would do it per frame: quat.py:541)
        const float o[3] = {a.offsets[3 * j], a.offsets[3 * j + 1], a.offsets[3 * j + 2]};
        float u[3];
        vnormalize(o, 1e-8f, u);
        sOff[3 * j] = u[0]; sOff[3 * j + 1] = u[1]; sOff[3 * j + 2] = u[2];
    }
    for (int j = lane; j <= J; j += PM_WAVE) {
        if (j < J) { sTopo[j] = a.topo.parent[j]; sTopo[2 * J + 1 + j] = a.topo.clist[j]; }
        sTopo[J + j] = a.topo.cstart[j];
    }
    wave_sync();

    // ---- the walk: one lane per frame ------------------------------------------------------------------------
    const int f = lane % FPW;  // lanes >= FPW shadow lanes 0.. ; frames past a partial tile use their own slots
    float *fS = sS + f * FS;
    float g[4] = {1.0f, 0.0f, 0.0f, 0.0f};  // world quaternion of the previous joint
    for (int j = 0; j < J; ++j) {
        const int par = sTopo[j];
        float gpre[4] = {g[0], g[1], g[2], g[3]};
        if (j == 0) { gpre[0] = 1.0f; gpre[1] = 0.0f; gpre[2] = 0.0f; gpre[3] = 0.0f; }
        else if (par != j - 1) lds_get<4>(fS, par, gpre);  // wave-uniform: a finished joint's slot holds its G
        const int cs = sTopo[J + j], ce = sTopo[J + j + 1];  // wave-uniform
        float rot[4] = {1.0f, 0.0f, 0.0f, 0.0f};
        if (ce > cs) {
            const int c0 = sTopo[2 * J + 1 + cs];
            float pj[4], pc[4];
            lds_get<4>(fS, j, pj);
            lds_get<4>(fS, c0, pc);  // children come later: their slots still hold positions
            const float d[3] = {pc[0] - pj[0], pc[1] - pj[1], pc[2] - pj[2]};
            const float inv[4] = {gpre[0], -gpre[1], -gpre[2], -gpre[3]};
            float pred[3];
            qmulvec(inv, d, pred);
            const float rest[3] = {sOff[3 * c0], sOff[3 * c0 + 1], sOff[3 * c0 + 2]};  // already unit
            float predn[3];
            vnormalize(pred, 1e-8f, predn);
            from_to_unit(rest, predn, rot);
            for (int k = cs + 1; k < ce; ++k) {  // roll correction from every further child
                const int gc = sTopo[2 * J + 1 + k];
                float gj[4], rn[4], pg[4];
                qnormalize(rot, 1e-8f, rn);  // the reference's fk normalises local rotations (skeleton.py:45)
                qmul(gpre, rn, gj);
                const float ginv[4] = {gj[0], -gj[1], -gj[2], -gj[3]};
                lds_get<4>(fS, gc, pg);
                const float dg[3] = {pg[0] - pj[0], pg[1] - pj[1], pg[2] - pj[2]};
                float predg[3], dn[3], axis[3], roll[4], r2[4];
                qmulvec(ginv, dg, predg);
                vnormalize(d, 1e-8f, dn);
                qmulvec(ginv, dn, axis);
                const float restg[3] = {sOff[3 * gc], sOff[3 * gc + 1], sOff[3 * gc + 2]};  // already unit
                float predgn[3];
                vnormalize(predg, 1e-8f, predgn);
                from_to_axis_unit(restg, predgn, axis, roll);
                qmul(rot, roll, r2);
                rot[0] = r2[0]; rot[1] = r2[1]; rot[2] = r2[2]; rot[3] = r2[3];
            }
        }
        if (ce > cs) {  // (a childless joint keeps the identity, nobody reads its G: nothing to do)
            float rn[4];
            qnormalize(rot, 1e-8f, rn);
            qmul(gpre, rn, g);
            lds_put<4>(fS, j, g);  // P_j is dead from here on
        }
    }
    wave_sync();

    // ---- local rotations back out of the world quaternions, lane per (frame, joint), straight to HBM -----------
    float *gout = a.out + f0 * J * 4;
    for_each_slot<2>(n, lane, [&](const int e, const bool valid) {
        const int fr = (int)(((float)e + 0.5f) * invJ);
        const int j = e - fr * J;
        const float *fq = sS + fr * FS;
        float gj[4], gp[4], o[4];
        lds_get<4>(fq, j, gj);
        lds_get<4>(fq, sTopo[j], gp);
        const float inv[4] = {gp[0], -gp[1], -gp[2], -gp[3]};
        qmul(inv, gj, o);
        const bool leaf = sTopo[J + j + 1] == sTopo[J + j];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = leaf ? (k == 0 ? 1.0f : 0.0f) : ((j == 0) ? gj[k] : o[k]);
        if (valid) {
            if (VEC) __builtin_nontemporal_store(v4f{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4f *>(gout) + e);
            else { gout[4 * e] = o[0]; gout[4 * e + 1] = o[1]; gout[4 * e + 2] = o[2]; gout[4 * e + 3] = o[3]; }
        }
    });
}

template <int FPW>
static int launch_ik(const IkArgs &a, bool vec, hipStream_t s) {
    const size_t lds = ((size_t)FPW * ik_frame_stride(a.J) + 3 * a.J + 3 * a.J + 2) * sizeof(float);
    const int64_t ntiles = (a.F + FPW - 1) / FPW;
    const int64_t grid = ((ntiles + PM_NXCD - 1) / PM_NXCD) * PM_NXCD;
    if (grid > 0x7fffffffLL) { set_error("from_root_positions: grid too large"); return PM_EUNSUPPORTED; }
    if (vec) {
        auto k = from_root_positions_kernel<FPW, true>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    } else {
        auto k = from_root_positions_kernel<FPW, false>;
        if (int e = allow_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(PM_WAVE), lds, s, a);
    }
    return check_hip(hipGetLastError(), "from_root_positions launch");
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
    const size_t per_frame = (size_t)ik_frame_stride(J) * sizeof(float), fixed = (size_t)(6 * J + 2) * sizeof(float) + 256;
    {
        const int v = tune_env("PM_IK_FPW", 0);  // PM_TUNING build only
        if (v == 64 && 64 * per_frame + fixed <= kMaxLds) return launch_ik<64>(a, vec, s);
        if (v == 32 && 32 * per_frame + fixed <= kMaxLds) return launch_ik<32>(a, vec, s);
        if (v == 16) return launch_ik<16>(a, vec, s);
        if (v == 8) return launch_ik<8>(a, vec, s);
    }
    if (4 * (64 * per_frame + fixed) <= kMaxLds) return launch_ik<64>(a, vec, s);  // every lane busy, >= 4 waves per CU
    if (4 * (32 * per_frame + fixed) <= kMaxLds) return launch_ik<32>(a, vec, s);
    if (2 * (16 * per_frame + fixed) <= kMaxLds) return launch_ik<16>(a, vec, s);
    if (4 * per_frame + fixed <= kMaxLds) return launch_ik<4>(a, vec, s);
    set_error("from_root_positions: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
}
