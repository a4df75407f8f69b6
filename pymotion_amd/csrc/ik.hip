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
// HBM traffic: 12 J B/frame in, 16 J out.  Positions are staged in LDS (coalesced), results leave coalesced.
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
    const int FS = 4 * J + 4;                 // padded frame stride of the quaternion images (see dq.hip)
    float *sOut = smem;                       // [FPW * FS]   local rotations (output image)
    float *sG = sOut + FPW * FS;              // [FPW * FS]   world quaternions of finished joints
    float *sP = sG + FPW * FS;                // [FPW * J * 3] positions tile (linear)
    float *sOff = sP + FPW * J * 3;           // [J * 3]
    int *sTopo = reinterpret_cast<int *>(sOff + 3 * J);  // [J] parent | [J+1] cstart | [J] clist

    tile_load<VEC>(a.pos + f0 * J * 3, sP, n * 3, lane);
    for (int i = lane; i < 3 * J; i += PM_WAVE) sOff[i] = a.offsets[i];
    for (int j = lane; j <= J; j += PM_WAVE) {
        if (j < J) { sTopo[j] = a.topo.parent[j]; sTopo[2 * J + 1 + j] = a.topo.clist[j]; }
        sTopo[J + j] = a.topo.cstart[j];
    }
    wave_sync();

    const int f = lane % FPW;  // lanes >= FPW shadow lanes 0.. ; frames past a partial tile use their own slots
    const float *fP = sP + f * J * 3;
    float *fOut = sOut + f * FS, *fG = sG + f * FS;
    for (int j = 0; j < J; ++j) {
        float gpre[4] = {1.0f, 0.0f, 0.0f, 0.0f};
        if (j > 0) lds_get<4>(fG, sTopo[j], gpre);
        const int cs = sTopo[J + j], ce = sTopo[J + j + 1];  // wave-uniform
        float rot[4] = {1.0f, 0.0f, 0.0f, 0.0f};
        if (ce > cs) {
            const int c0 = sTopo[2 * J + 1 + cs];
            const float pj[3] = {fP[3 * j], fP[3 * j + 1], fP[3 * j + 2]};
            const float d[3] = {fP[3 * c0] - pj[0], fP[3 * c0 + 1] - pj[1], fP[3 * c0 + 2] - pj[2]};
            const float inv[4] = {gpre[0], -gpre[1], -gpre[2], -gpre[3]};
            float pred[3];
            qmulvec(inv, d, pred);
            const float rest[3] = {sOff[3 * c0], sOff[3 * c0 + 1], sOff[3 * c0 + 2]};
            from_to(rest, pred, true, rot);
            for (int k = cs + 1; k < ce; ++k) {  // roll correction from every further child
                const int gc = sTopo[2 * J + 1 + k];
                float gj[4], rn[4];
                qnormalize(rot, 1e-8f, rn);  // the reference's fk normalises local rotations (skeleton.py:45)
                qmul(gpre, rn, gj);
                const float ginv[4] = {gj[0], -gj[1], -gj[2], -gj[3]};
                const float dg[3] = {fP[3 * gc] - pj[0], fP[3 * gc + 1] - pj[1], fP[3 * gc + 2] - pj[2]};
                float predg[3], dn[3], axis[3], roll[4], r2[4];
                qmulvec(ginv, dg, predg);
                vnormalize(d, 1e-8f, dn);
                qmulvec(ginv, dn, axis);
                const float restg[3] = {sOff[3 * gc], sOff[3 * gc + 1], sOff[3 * gc + 2]};
                from_to_axis(restg, predg, axis, true, roll);
                qmul(rot, roll, r2);
                rot[0] = r2[0]; rot[1] = r2[1]; rot[2] = r2[2]; rot[3] = r2[3];
            }
        }
        float g[4], rn[4];
        qnormalize(rot, 1e-8f, rn);
        qmul(gpre, rn, g);
        lds_put<4>(fOut, j, rot);
        lds_put<4>(fG, j, g);
    }
    wave_sync();
    float *gout = a.out + f0 * J * 4;
    const float invJ = 1.0f / (float)J;
    for (int i = lane; i < n; i += PM_WAVE) {  // dwordx4 i = (frame i / J, joint i % J) of the padded image
        const int fr = (int)(((float)i + 0.5f) * invJ);
        const int j = i - fr * J;
        const v4f v = *reinterpret_cast<const v4f *>(sOut + fr * FS + j * 4);
        if (VEC) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(gout) + i);
        else { gout[4 * i] = v.x; gout[4 * i + 1] = v.y; gout[4 * i + 2] = v.z; gout[4 * i + 3] = v.w; }
    }
}

template <int FPW>
static int launch_ik(const IkArgs &a, bool vec, hipStream_t s) {
    const size_t lds = ((size_t)FPW * (2 * (4 * a.J + 4) + 3 * a.J) + 3 * a.J + 3 * a.J + 2) * sizeof(float);
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
    const size_t per_frame = (size_t)(2 * (4 * J + 4) + 3 * J) * sizeof(float), fixed = (size_t)(6 * J + 2) * sizeof(float) + 256;
    {
        const char *e = getenv("PM_IK_FPW");  // tuning aid
        const int v = e ? atoi(e) : 0;
        if (v == 64 && 64 * per_frame + fixed <= kMaxLds) return launch_ik<64>(a, vec, s);
        if (v == 32 && 32 * per_frame + fixed <= kMaxLds) return launch_ik<32>(a, vec, s);
        if (v == 16) return launch_ik<16>(a, vec, s);
        if (v == 8) return launch_ik<8>(a, vec, s);
    }
    if (5 * (32 * per_frame + fixed) <= kMaxLds) return launch_ik<32>(a, vec, s);
    if (2 * (16 * per_frame + fixed) <= kMaxLds) return launch_ik<16>(a, vec, s);
    if (4 * per_frame + fixed <= kMaxLds) return launch_ik<4>(a, vec, s);
    set_error("from_root_positions: J=%d does not fit the LDS tile", J);
    return PM_EUNSUPPORTED;
}
