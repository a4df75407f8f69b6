// dq.hip -- root-centred dual-quaternion encode / decode and from_global_rotations for gfx950.
//   to_root_dual_quat    pymotion/ops/skeleton.py:207-244   (sequential along each parent chain)
//   from_root_dual_quat  pymotion/ops/skeleton.py:173-204   (every joint needs only its parent's INPUT)
//   from_global_rotations pymotion/ops/skeleton.py:64-93    (same gather shape)
// Same wave-private tiling as fk.hip: HBM <-> LDS with contiguous dwordx4, AoS access from LDS.
#include <stdlib.h>

#include "common.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------------
// to_root_dual_quat.  The payload is a quaternion + a translation; a QUAD (4 consecutive lanes) owns a
// frame, lane c holding component c of the running root-space quaternion and of the translation written
// as the pure quaternion (0, t).  Products with a distributed left operand use
//     (a (x) b)_c = sum_k S[c][k] a_k b_{c xor k}          (Klein-group structure of the Hamilton product)
// with a_k a DPP quad broadcast and the b's read from LDS at per-lane permuted addresses; cross products
// of the rotate-a-vector formula (quat.py:320-334) use DPP rotations of lanes 1..3.  A chain step is ~40
// instructions for 16 frames, never waits on LDS, and a wave needs only 16 frames of LDS image (11.5 KiB
// at J = 22 -> 13 waves per CU).
// The LDS image is the tile's OUTPUT (32 J B per frame) with one 16-byte pad per frame: bank-conflict
// free column access, and the copy-out stays a dwordx4 stream.
//   phase A  lane per (frame, joint): quaternion straight from HBM (coalesced dwordx4, loads pipelined)
//            into the first half of the joint's 32-byte output slot;
//   walk     per joint: compose with the parent (registers when parents[j] == j-1, else its slot), or stay
//            local when the parent is the root (skeleton.py:236-241); slot <- root-space (q, t, 0);
//   phase C  lane per (frame, joint): (q, t) -> [q, 0.5 (0,t) (x) q] in place (dual_quat.py:28-36);
//   out      contiguous dwordx4 streaming stores.
// Skeleton constants {parent, offset} sit in a small LDS table.
// ---------------------------------------------------------------------------------------------------
struct ToRootArgs {
    const float *rot;       // [F,J,4]
    const float *root_pos;  // [F,3]
    const float *offsets;   // [J,3]
    float *dq;              // [F,J,8]
    int64_t F;
    int32_t J;
    int32_t ablate;  // tuning aid (env PM_DQ_ABLATE): 1 = skip the walk, 2 = skip phase C
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
    const int n = nf * J;
    const int FS = 8 * J + 4;           // padded frame stride in floats
    float *sDq = smem;                  // [FPW * FS]
    float *sConst = sDq + FPW * FS;     // [(J+1)*4]
    const float invJ = 1.0f / (float)J;

    // all global loads first: root position, constants, then the rotations in batches of 4
    // lanes >= 4*FPW shadow lanes 0.. ; frames past a partial tile walk their own (unused) slots: no masking
    const int wl = lane % (4 * FPW);
    const int fq = wl >> 2, c = wl & 3;
    const float rp = (c > 0 && fq < nf) ? a.root_pos[(f0 + fq) * 3 + c - 1] : 0.0f;  // (0, root_pos) component c
    for (int j = lane; j <= J; j += PM_WAVE) {
        const int jc = j < J ? j : J - 1;
        v4f c;
        c.x = __int_as_float(a.parents.p[jc]);
        c.y = a.offsets[3 * jc]; c.z = a.offsets[3 * jc + 1]; c.w = a.offsets[3 * jc + 2];
        reinterpret_cast<v4f *>(sConst)[j] = c;
    }
    const float *gsrc = a.rot + f0 * J * 4;
    auto load_batch = [&](const int e0, v4f (&q)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            if (e < n) {
                if (VEC) q[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(gsrc) + e);
                else q[u] = v4f{gsrc[4 * e], gsrc[4 * e + 1], gsrc[4 * e + 2], gsrc[4 * e + 3]};
            }
        }
    };
    auto park_batch = [&](const int e0, const v4f (&q)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * PM_WAVE + lane;
            if (e < n) {
                const int f = (int)(((float)e + 0.5f) * invJ);  // e / J, exact for e < 2^22
                const int j = e - f * J;
                *reinterpret_cast<v4f *>(sDq + f * FS + j * 8) = q[u];
            }
        }
    };
    {   // two batches (8 KiB per wave) in flight, batch k+2 requested before batch k is consumed
        constexpr int B = 4 * PM_WAVE;
        v4f qa[4], qb[4];
        load_batch(0, qa);
        load_batch(B, qb);
        for (int e0 = 0; e0 < n; e0 += 2 * B) {
            park_batch(e0, qa);
            load_batch(e0 + 2 * B, qa);
            park_batch(e0 + B, qb);
            load_batch(e0 + 3 * B, qb);
        }
    }
    wave_sync();

    float *fD = sDq + fq * FS;
    const float *cstf = sConst;
    // per-lane constants of the component layout
    const float s1 = (c == 0 || c == 2) ? -1.0f : 1.0f;   // S[c][1]:  - + - +
    const float s2 = (c == 0 || c == 3) ? -1.0f : 1.0f;   // S[c][2]:  - + + -
    const float s3 = (c == 0 || c == 1) ? -1.0f : 1.0f;   // S[c][3]:  - - + +
    const int x1 = c ^ 1, x2 = c ^ 2, x3 = c ^ 3;         // b_{c xor k}
    const int cn1 = (c == 3) ? 1 : c + 1, cn2 = (c == 2) ? 1 : ((c == 3) ? 2 : c + 2);  // next / next-next of x,y,z
    const float live = (c == 0) ? 0.0f : 1.0f;            // lane 0 carries the zero scalar part of (0, t)
    const int toff = (c == 0) ? 7 : 3 + c;                // where component c of (0,t) sits in a slot: t0 t1 t2 0

    float gq = (c == 0) ? 1.0f : 0.0f, gt = 0.0f;  // joint j-1, root space
    // look-ahead state for joint 0
    float b0 = fD[c], b1 = fD[x1], b2 = fD[x2], b3 = fD[x3];
    int par = 0;
    float vc = 0.0f, vn = 0.0f, vnn = 0.0f;
    for (int j = (a.ablate & 1) ? J : 0; j < J; ++j) {
        // joint j+1's inputs do not depend on the chain: request them now (slot j+1 still holds its quaternion)
        const int jn = j + 1;  // table has J+1 entries; the slot past the last joint is inside the allocation
        const float nb0 = fD[jn * 8 + c], nb1 = fD[jn * 8 + x1], nb2 = fD[jn * 8 + x2], nb3 = fD[jn * 8 + x3];
        const int parn = __builtin_amdgcn_readfirstlane(__float_as_int(cstf[jn * 4]));
        const float nvc = cstf[jn * 4 + c], nvn = cstf[jn * 4 + cn1], nvnn = cstf[jn * 4 + cn2];  // lane 0: unused

        float q, t;
        if (j == 0) {
            q = b0; t = rp;                       // skeleton.py:232
        } else if (par == 0) {
            q = b0; t = vc * live;                // joints hanging off the root stay local (:236-237)
        } else {
            float pq = gq, pt = gt;
            if (par != j - 1) { pq = fD[par * 8 + c]; pt = fD[par * 8 + toff]; }  // finished slot: (q, t, 0)
            const float pw = quad_bcast<0>(pq), px = quad_bcast<1>(pq), py = quad_bcast<2>(pq), pz = quad_bcast<3>(pq);
            // q = pq (x) q_j   (quat.py:337-361 in component-parallel form)
            q = pw * b0 + s1 * (px * b1) + s2 * (py * b2) + s3 * (pz * b3);
            // t = pq . t_j + pt   (quat.py:320-334: tt = 2 (pv x v); v + w tt + pv x tt)
            const float an = quad_perm<0, 2, 3, 1>(pq), ann = quad_perm<0, 3, 1, 2>(pq);
            const float tt = 2.0f * (an * vnn - ann * vn);
            const float ttn = quad_perm<0, 2, 3, 1>(tt), ttnn = quad_perm<0, 3, 1, 2>(tt);
            t = (vc + pw * tt + (an * ttnn - ann * ttn) + pt) * live;
        }
        fD[j * 8 + c] = q;
        fD[j * 8 + toff] = t;
        gq = q; gt = t;
        b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
        par = parn; vc = nvc; vn = nvn; vnn = nvnn;
    }
    wave_sync();
    // phase C, lane per (frame, joint): (q, t) -> [q, 0.5 (0,t) (x) q]  (dual_quat.py:28-36), off the chain
    for (int e = (a.ablate & 2) ? n : lane; e < n; e += PM_WAVE) {
        const int f = (int)(((float)e + 0.5f) * invJ);
        const int j = e - f * J;
        float *slot = sDq + f * FS + j * 8;
        float qt[8], d[8];
        lds_get<8>(slot, 0, qt);
        const float q[4] = {qt[0], qt[1], qt[2], qt[3]}, t[3] = {qt[4], qt[5], qt[6]};
        rt2dq(q, t, d);
        *reinterpret_cast<v4f *>(slot + 4) = v4f{d[4], d[5], d[6], d[7]};
    }
    wave_sync();
    // copy-out: dwordx4 i of the tile lives at frame i / 2J, chunk i % 2J of the padded image
    float *gout = a.dq + f0 * J * 8;
    const int n4 = n * 2, J2 = 2 * J;
    const float invJ2 = 1.0f / (float)J2;
    for (int i = lane; i < n4; i += PM_WAVE) {
        const int f = (int)(((float)i + 0.5f) * invJ2);
        const int r = i - f * J2;
        const v4f v = *reinterpret_cast<const v4f *>(sDq + f * FS + r * 4);
        if (VEC) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(gout) + i);
        else { gout[4 * i] = v.x; gout[4 * i + 1] = v.y; gout[4 * i + 2] = v.z; gout[4 * i + 3] = v.w; }
    }
}

template <int FPW>
static int launch_to_root(const ToRootArgs &a, bool vec, hipStream_t s) {
    const size_t lds = ((size_t)FPW * (8 * a.J + 4) + 4 * (a.J + 1)) * sizeof(float);
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
    int *sPar = reinterpret_cast<int *>(sO1 + FJ * (MODE == 0 ? 4 : 0));  // [J] parents, staged once
    for (int j = lane; j < J; j += PM_WAVE) sPar[j] = (j == 0) ? 0 : a.parents.p[j];
    wave_sync();
    const int n = nf * J;
    const float invJ = 1.0f / (float)J;
    for (int e = lane; e < n; e += PM_WAVE) {
        const int f = (int)(((float)e + 0.5f) * invJ);  // e / J without an integer divide (exact for e < 2^22)
        const int j = e - f * J;
        const int par = sPar[j];
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
    // Frames per wave: a multiple of 4 (16-byte tile bases) giving ~160-210 (frame,joint) items per wave.
    // Small tiles win here -- there is no chain to amortise and many resident waves hide the load latency
    // (measured at 2^20 x 22: 8/12/16/20 frames per wave -> 250/266/312/314 us; at 2^18 x 52: 4/8 -> 174/192).
    const size_t per_frame = (size_t)a.J * gather_lds_w<MODE>() * sizeof(float);
    const size_t extra = (size_t)a.J * sizeof(int);  // parents table
    int fpw = ((160 + a.J - 1) / a.J + 3) & ~3;
    if (fpw < 4) fpw = 4;
    {
        const char *e = getenv("PM_GATHER_FPW");  // tuning aid
        if (e && atoi(e) >= 4) fpw = atoi(e) & ~3;
    }
    while (fpw > 4 && fpw * per_frame + extra > kMaxLds / 4) fpw -= 4;
    if (fpw * per_frame + extra > kMaxLds) { set_error("gather: J too large for LDS"); return PM_EUNSUPPORTED; }
    const size_t lds = fpw * per_frame + extra;
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
    { const char *ab = getenv("PM_DQ_ABLATE"); a.ablate = ab ? atoi(ab) : 0; }
    if (int e = pack_parents(parents, J, a.parents)) return e;
    const bool vec = aligned16(rot) && aligned16(dq);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t per_frame = ((size_t)J * 8 + 4) * sizeof(float), fixed = 4 * ((size_t)J + 1) * sizeof(float) + 256;
    int pick = (7 * (16 * per_frame + fixed) <= kMaxLds) ? 16 : 8;  // 4 lanes per frame; keep >= 7 waves per CU if possible
    {
        const char *e = getenv("PM_DQ_FPW");  // tuning aid
        if (e && (atoi(e) == 16 || atoi(e) == 8 || atoi(e) == 4)) pick = atoi(e);
    }
    while (pick > 4 && pick * per_frame + fixed > kMaxLds) pick >>= 1;
    if (pick * per_frame + fixed <= kMaxLds) {
        if (pick == 16) return launch_to_root<16>(a, vec, s);
        if (pick == 8) return launch_to_root<8>(a, vec, s);
        return launch_to_root<4>(a, vec, s);
    }
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
