// unroll.hip -- quat.unroll (pymotion/rotations/quat.py:426-462) for gfx950: the first frame-COUPLED op.
//
// Reference: walk the unroll axis; flip quaternion i when d0 = dot(q_i, q'_{i-1}) < d1 = -d0, q'_{i-1} being the
// already-corrected one.  With s_i the sign frame i ends up with (s_0 = +1) and dot_i the dot product of the ORIGINAL
// neighbours, d0 = s_{i-1} dot_i, so
//     dot_i > 0: s_i = s_{i-1}      dot_i < 0: s_i = -s_{i-1}      dot_i == 0 or NaN: s_i = +1   (`d0 < d1` is false)
// i.e. every frame applies one of three maps to the running sign -- keep, negate, RESET to + -- and maps of that kind
// compose associatively (affine maps over GF(2): (a, b): s -> a s xor b; keep = (1,0), negate = (1,1), reset = (0,0)).
// Without resets that is a prefix XOR of the flip bits; a reset (a zero-padded row, an exactly orthogonal step, a NaN)
// forgets everything before it.  A scan, not a loop:
//   pass 1  each wave streams a chunk of 256 consecutive frames (one record per lane, the predecessor row an
//           L1 / L2 hit) and XORs the flip bits into per-series parities (LDS atomics) -> workspace;
//   pass 2  exclusive prefix XOR over chunks (lane = series; a few thousand independent loads);
//   pass 3  each wave owns a chunk of up to 24 series; per 64-frame sub-tile the rows are staged in LDS
//           (coalesced), lane = frame computes its flip bit per series, one wave ballot per series gives every
//           lane its inclusive prefix parity (+ the chunk's carry-in); the corrected records go back into the
//           LDS rows and leave as contiguous dwordx4 streams.
// (LDS: 65 rows x 25 records x 16|32 B = 26|52 KiB at most.)
// Layout: q [T, S, 4] (unroll axis first; the front-end moves it there), out same.
// Algorithmic HBM bytes: 16 (pass 1) + 16 + 16 (pass 3) = 48 B per quaternion.
#include "common.hpp"

namespace pm {

constexpr int UR_SUB = PM_WAVE;   // frames per sub-tile (lane = frame)
constexpr int UR_CHUNK = 256;     // frames per wave (4 sub-tiles): 4096 waves at 2^20 frames
constexpr int UR_SB_MAX = 24;     // series per block at most: 65 rows x 25 x 16 B = 26 KiB of LDS -> 6 waves per CU

struct UnrollArgs {
    const float *q;
    float *out;
    int32_t *ws;      // [nchunks][S] chunk summaries (pass 1 out: bit 0 = parity of the flips after the chunk's last
                      // reset, bit 1 = the chunk holds a reset), then exclusive prefixes (pass 2, in place: bit 0)
    int64_t T;
    int32_t S;
    int32_t nchunks;
    int32_t sbsize;   // series per block of the apply pass (<= UR_SB_MAX, balanced over the blocks)
    int32_t sblocks;  // number of such blocks; the grid is 1-D: block = chunk * sblocks + series block
    int32_t p1_sb, p1_blocks;  // the same for the parity pass (series per block chosen so that the grid fills the chip)
};

// W = 4: quaternions; W = 8: dual quaternions (sign decided by the real part, applied to all 8 floats,
// rotations/dual_quat.py:139-167).
template <int W>
__global__ __launch_bounds__(PM_WAVE) void unroll_apply_kernel(const UnrollArgs a) {
    constexpr int V = W / 4;  // dwordx4 per record
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x / a.sblocks;
    const int s0 = (blockIdx.x - chunk * a.sblocks) * a.sbsize;
    const int sb = (a.S - s0) < a.sbsize ? (a.S - s0) : a.sbsize;  // series in this block
    const int64_t t0 = (int64_t)chunk * UR_CHUNK;
    const int64_t t1 = (t0 + UR_CHUNK) < a.T ? (t0 + UR_CHUNK) : a.T;
    const int rs = sb | 1;  // row stride in quaternions, odd: per-lane ds_read_b128 down a column is conflict-free
    v4f *rows = reinterpret_cast<v4f *>(smem);  // [(UR_SUB + 1)][rs][V]: row 0 = the frame before the sub-tile

    // carry-in parity per series (lane = series): exclusive prefix over earlier chunks
    unsigned long long carry = 0;  // bit j = parity of series s0 + j
    {
        const int c = (lane < sb) ? a.ws[(int64_t)chunk * a.S + s0 + lane] : 0;
        carry = __ballot(c & 1);
    }
    for (int64_t ts = t0; ts < t1; ts += UR_SUB) {
        const int nfr = (int)((t1 - ts) < UR_SUB ? (t1 - ts) : UR_SUB);
        // stage rows ts-1 .. ts+nfr-1 (row segments of sb quaternions are contiguous in HBM)
        const int first = (ts == 0) ? 1 : 0;  // no predecessor for the very first frame
        const int rowlen = sb * V;  // dwordx4 per staged row segment
        const float inv_rowlen = 1.0f / (float)rowlen;
        const int i_end = (nfr + 1) * rowlen;
        constexpr int UR_DEPTH = 8;  // loads in flight per lane (8 KiB per wave: with 6 resident waves, 4 starved HBM)
        for (int i0 = lane + first * rowlen; i0 < i_end; i0 += UR_DEPTH * PM_WAVE) {
            v4f v[UR_DEPTH];
            int dst[UR_DEPTH];
#pragma unroll
            for (int u = 0; u < UR_DEPTH; ++u) {
                const int i = i0 + u * PM_WAVE, ic = i < i_end ? i : i_end - 1;  // clamped: loads are unconditional
                const int r = (int)(((float)ic + 0.5f) * inv_rowlen), c = ic - r * rowlen;
                dst[u] = r * rs * V + c;
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a.q) + ((ts - 1 + r) * a.S + s0) * V + c);
            }
#pragma unroll
            for (int u = 0; u < UR_DEPTH; ++u)
                if (i0 + u * PM_WAVE < i_end) rows[dst[u]] = v[u];
        }
        wave_sync();
        const bool act = lane < nfr;
        unsigned long long newcarry = carry;
        for (int j = 0; j < sb; ++j) {
            const v4f cur = rows[((lane + 1) * rs + j) * V];
            bool flip = false, reset = false;
            if (act && !(ts == 0 && lane == 0)) {
                const v4f prv = rows[(lane * rs + j) * V];
                const float d = cur.x * prv.x + cur.y * prv.y + cur.z * prv.z + cur.w * prv.w;
                flip = d < 0.0f;
                reset = !(d < 0.0f) && !(d > 0.0f);  // 0, -0 or NaN: the reference's `d0 < d1` is false whatever came before
            }
            const unsigned long long m = __ballot(flip), mr = __ballot(reset);
            {
                const unsigned long long upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
                int par = (__popcll(m & upto) + (int)((carry >> j) & 1ull)) & 1;
                if (mr != 0) {  // wave-uniform, rare: lanes at or after a reset only see the flips after the LAST reset before them
                    const unsigned long long rb = mr & upto;
                    if (rb != 0) {
                        const int r = 63 - __builtin_clzll(rb);
                        const unsigned long long after = upto & ~((r == 63) ? ~0ull : ((2ull << r) - 1ull));
                        par = __popcll(m & after) & 1;
                    }
                }
                if (act) {
                    // corrected record back into its row, in place: every lane has read this series' `cur` and
                    // `prv` above (in-order DS), and the flip bits only ever use ORIGINAL neighbours
                    const float sg = par ? -1.0f : 1.0f;
                    rows[((lane + 1) * rs + j) * V] = v4f{cur.x * sg, cur.y * sg, cur.z * sg, cur.w * sg};
                    if constexpr (V == 2) {
                        const v4f du = rows[((lane + 1) * rs + j) * V + 1];
                        rows[((lane + 1) * rs + j) * V + 1] = v4f{du.x * sg, du.y * sg, du.z * sg, du.w * sg};
                    }
                }
            }
            if (mr != 0) {
                const int r = 63 - __builtin_clzll(mr);
                const unsigned long long after = (r == 63) ? 0ull : ~((2ull << r) - 1ull);
                newcarry = (newcarry & ~(1ull << j)) | ((unsigned long long)(__popcll(m & after) & 1) << j);
            } else if (__popcll(m) & 1) {
                newcarry ^= (1ull << j);
            }
        }
        {
            // rows 1..nfr leave the way they came: contiguous dwordx4, 4 stores in flight per lane
            wave_sync();
            const int o_end = nfr * rowlen;
            for (int i = lane; i < o_end; i += PM_WAVE) {
                const int r = (int)(((float)i + 0.5f) * inv_rowlen), c = i - r * rowlen;
                const v4f v = rows[(r + 1) * rs * V + c];
                __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(a.out) + ((ts + r) * a.S + s0) * V + c);
            }
        }
        carry = newcarry;
        wave_sync();
    }
}

// pass 1 as a plain stream (no LDS tile, one record per lane, consecutive lanes on consecutive records): the
// flip bit of record (t, s) needs record (t-1, s), which the same wave fetched S records earlier (an L1 / L2
// hit), and the chunk's parity per series is an LDS atomic XOR.  Series are taken in blocks of at most UR_P1_SB,
// fewer when the clip is short and wide, so that chunks x series blocks still fill the chip.
constexpr int UR_P1_SB = 8192;

template <int W>
__global__ __launch_bounds__(PM_WAVE) void unroll_parity_kernel(const UnrollArgs a) {
    constexpr int V = W / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *par = reinterpret_cast<int *>(smem);
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x / a.p1_blocks;
    const int s0 = (blockIdx.x - chunk * a.p1_blocks) * a.p1_sb;
    const int sb = (a.S - s0) < a.p1_sb ? (a.S - s0) : a.p1_sb;
    const int64_t t0 = (int64_t)chunk * UR_CHUNK;
    const int64_t t1 = (t0 + UR_CHUNK) < a.T ? (t0 + UR_CHUNK) : a.T;
    int *lastr = par + sb;  // local frame of the chunk's last reset per series, -1 = none
    for (int i = lane; i < sb; i += PM_WAVE) { par[i] = 0; lastr[i] = -1; }
    wave_sync();
    const int n = (int)(t1 - t0) * sb;  // records of this chunk x series block (< 2^22: UR_CHUNK * UR_P1_SB = 2^21)
    const float inv_sb = 1.0f / (float)sb;
    const v4f *q = reinterpret_cast<const v4f *>(a.q);
    // `second`: the (rare) re-count after a reset was seen: only flips AFTER the series' last reset make the summary
    auto sweep = [&](const bool second) {
        for (int i0 = 0; i0 < n; i0 += 4 * PM_WAVE) {
            v4f cur[4], prv[4];
            int ser[4], row[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * PM_WAVE + lane, ic = i < n ? i : n - 1;
                const int r = (int)(((float)ic + 0.5f) * inv_sb), c = ic - r * sb;
                const int64_t t = t0 + r;
                ser[u] = c;
                row[u] = r;
                ok[u] = (i < n) && (t > 0);
                const int64_t e = (t * a.S + s0 + c) * V;
                cur[u] = __builtin_nontemporal_load(q + e);
                prv[u] = q[t > 0 ? e - (int64_t)a.S * V : e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = cur[u].x * prv[u].x + cur[u].y * prv[u].y + cur[u].z * prv[u].z + cur[u].w * prv[u].w;
                const bool flip = ok[u] && d < 0.0f;
                const bool reset = ok[u] && !(d < 0.0f) && !(d > 0.0f);
                if (!second) {
                    if (flip) __hip_atomic_fetch_xor(par + ser[u], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (reset) __hip_atomic_fetch_max(lastr + ser[u], row[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (flip && row[u] > lastr[ser[u]]) {
                    __hip_atomic_fetch_xor(par + ser[u], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    sweep(false);
    wave_sync();
    bool any = false;
    for (int i = lane; i < sb; i += PM_WAVE) any = any || lastr[i] >= 0;
    if (__ballot(any) != 0) {  // wave-uniform; zero-padded rows, exactly orthogonal steps, NaN: not on the usual path
        for (int i = lane; i < sb; i += PM_WAVE) par[i] = 0;
        wave_sync();
        sweep(true);
        wave_sync();
    }
    for (int i = lane; i < sb; i += PM_WAVE) a.ws[(int64_t)chunk * a.S + s0 + i] = (par[i] & 1) | (lastr[i] >= 0 ? 2 : 0);
}

// exclusive prefix over chunks of the composed sign maps, in place: one 16-wave workgroup per series, 4096 chunks per
// trip -- every wave requests its 4 x 64 summaries up front (the one-wave form paid a memory latency per 64 chunks:
// 28 us of the 280 at 2^20 frames), prefixes inside a group of 64 are a ballot + popcount cut at the last reset, and the
// waves' totals meet in LDS.
__global__ __launch_bounds__(1024) void unroll_scan_kernel(int32_t *ws, int nchunks, int S) {
    __shared__ int wtot[16];
    const int s = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int carry = 0;  // sign parity entering this trip (uniform over the workgroup)
    for (int base = 0; base < nchunks; base += 4096) {
        int v[4], pre[4];
        bool cut[4];  // a reset lies between the start of this wave's range and the element: its prefix is absolute
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = base + wave * 256 + g * 64 + lane;
            v[g] = (c < nchunks) ? ws[(int64_t)c * S + s] : 0;
        }
        int run = 0;        // composition of the groups of this wave so far, entered with parity 0
        bool seen = false;  // ... contained a reset
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const unsigned long long m = __ballot(v[g] & 1), mr = __ballot(v[g] & 2);
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long rb = mr & below;
            if (rb != 0) {
                const int r = 63 - __builtin_clzll(rb);
                pre[g] = __popcll(m & below & ~((1ull << r) - 1ull)) & 1;
                cut[g] = true;
            } else {
                pre[g] = (__popcll(m & below) + run) & 1;
                cut[g] = seen;
            }
            if (mr != 0) {
                const int r = 63 - __builtin_clzll(mr);
                run = __popcll(m & ~((1ull << r) - 1ull)) & 1;
                seen = true;
            } else {
                run ^= __popcll(m) & 1;
            }
        }
        if (lane == 0) wtot[wave] = (seen ? 2 : 0) | run;
        __syncthreads();
        int cin = carry, all = carry;
        for (int w = 0; w < 16; ++w) {
            const int t = wtot[w];
            const int nxt = (t & 2) ? (t & 1) : (all ^ (t & 1));
            if (w < wave) cin = nxt;
            all = nxt;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = base + wave * 256 + g * 64 + lane;
            if (c < nchunks) ws[(int64_t)c * S + s] = cut[g] ? pre[g] : (pre[g] ^ cin);
        }
        carry = all;
        __syncthreads();
    }
}

// the same for short, wide clips (few chunks, many series): one THREAD per series walks the chunks; consecutive
// threads touch consecutive words of a chunk's row
__global__ __launch_bounds__(256) void unroll_scan_wide_kernel(int32_t *ws, int nchunks, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    int carry = 0;
    for (int c = 0; c < nchunks; ++c) {
        const int v = ws[(int64_t)c * S + s];
        ws[(int64_t)c * S + s] = carry;
        carry = (v & 2) ? (v & 1) : (carry ^ (v & 1));
    }
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_quat_unroll_workspace_bytes(int64_t T, int32_t S) {
    if (T <= 0 || S <= 0) return 0;
    return ((T + UR_CHUNK - 1) / UR_CHUNK) * (int64_t)S * (int64_t)sizeof(int32_t);
}

template <int W>
static int unroll_launch(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    PM_CHECK_ARGS(T >= 0 && S >= 0, "quat_unroll: negative size");
    if (T == 0 || S == 0) return PM_OK;
    PM_CHECK_ARGS(q && out && workspace, "quat_unroll: null pointer");
    PM_CHECK_ARGS(aligned16(q) && aligned16(out), "quat_unroll: q and out must be 16-byte aligned");
    const int64_t nchunks = (T + UR_CHUNK - 1) / UR_CHUNK;
    const int nsb = (S + UR_SB_MAX - 1) / UR_SB_MAX;
    const int sbsize = (S + nsb - 1) / nsb;
    const int sblocks = (S + sbsize - 1) / sbsize;
    // parity pass: series per block from the size of the grid it leaves (>= ~8 K waves if the problem has them)
    int p1_sb = UR_P1_SB;
    while (p1_sb > 64 && nchunks * ((S + p1_sb - 1) / p1_sb) < 8192) p1_sb >>= 1;
    const int64_t p1_blocks = (S + p1_sb - 1) / p1_sb;
    if (nchunks * sblocks > 0x7fffffffLL || nchunks * p1_blocks > 0x7fffffffLL) { set_error("quat_unroll: problem too large"); return PM_EUNSUPPORTED; }
    UnrollArgs a;
    a.q = q; a.out = out; a.ws = static_cast<int32_t *>(workspace); a.T = T; a.S = S; a.nchunks = (int)nchunks; a.sbsize = sbsize; a.sblocks = sblocks; a.p1_sb = p1_sb; a.p1_blocks = (int)p1_blocks;
    const size_t lds = (size_t)(UR_SUB + 1) * (sbsize | 1) * 4 * W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int e = allow_lds(unroll_apply_kernel<W>, lds)) return e;
    const dim3 grid((unsigned)(nchunks * sblocks));
    {   // pass 1: chunk parities
        const size_t p1_lds = 2 * (size_t)(S < p1_sb ? S : p1_sb) * sizeof(int);
        hipLaunchKernelGGL((unroll_parity_kernel<W>), dim3((unsigned)(nchunks * p1_blocks)), dim3(PM_WAVE), p1_lds, s, a);
    }
    if (nchunks <= 256) hipLaunchKernelGGL(unroll_scan_wide_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, a.ws, (int)nchunks, (int)S);
    else hipLaunchKernelGGL(unroll_scan_kernel, dim3((unsigned)S), dim3(1024), 0, s, a.ws, (int)nchunks, (int)S);
    hipLaunchKernelGGL((unroll_apply_kernel<W>), grid, dim3(PM_WAVE), lds, s, a);
    return PM_AFTER_LAUNCH("quat_unroll");
}

extern "C" int pm_quat_unroll_f32(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<4>(q, T, S, out, workspace, stream);
}
extern "C" int pm_dq_unroll_f32(const float *dq, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<8>(dq, T, S, out, workspace, stream);
}
