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
//   pass 1  (unroll_mask_kernel) each wave streams a chunk of 256 consecutive frames (one record per lane, the
//           predecessor row an L1 / L2 hit), ORs flip / reset bits into per-series LDS masks and turns them into PREFIX
//           parities relative to the chunk's entry (a shift-XOR ladder per 32 frames); masks + a 2-bit chunk summary
//           per series go to the workspace (T S / 4 bytes in all);
//   pass 2  exclusive prefix of the chunk summaries (composition of keep / negate / reset maps);
//   pass 3  (unroll_apply_kernel) a pure stream: a record's sign is one mask bit XOR its chunk's entering parity.
// Round 1 staged 64-frame sub-tiles of the rows in LDS in pass 3 and re-derived the flips there with ballots (185 us of
// the 254 at 2^20 x 22); as a stream it runs at the element-wise kernels' rate.
// Layout: q [T, S, 4] (unroll axis first; the front-end moves it there), out same.
// Algorithmic HBM bytes: 16 (pass 1) + 16 + 16 (pass 3) = 48 B per quaternion.
#include "common.hpp"

namespace pm {

constexpr int UR_CHUNK = 256;     // frames per chunk: 8 mask words per (chunk, series)
constexpr int UR_WORDS = UR_CHUNK / 32;
constexpr int UR_P1_SB = 512;     // series per block of the mask pass at most (72 B of LDS per series)

// Workspace (pm_quat_unroll_workspace_bytes), all int32 / uint32, nchunks = ceil(T / 256):
//   sum  [nchunks][S]      pass 1: bit 0 = sign parity a chunk leaves behind when entered with +, bit 1 = it holds a reset;
//                          pass 2 (in place): bit 0 = parity ENTERING the chunk, bit 1 kept
//   pre  [nchunks][S][8]   bit r: parity of frame 256 chunk + r relative to the chunk's entry (flips since the chunk's
//                          start, or since the last reset at or before r)
//   abs  [nchunks][S][8]   bit r: a reset lies at or before frame r inside the chunk -> the entering parity does not apply
struct UnrollArgs {
    const float *q;
    float *out;
    int32_t *sum;
    uint32_t *pre, *abs;
    int64_t T;
    int32_t S;
    int32_t nchunks;
    int32_t p1_sb, p1_blocks;  // mask pass: series per block (chosen so that the grid fills the chip) and blocks per chunk
};

// pass 1: one wave per (chunk, series block) streams its records (one record per lane, consecutive lanes on consecutive
// records; the predecessor row was fetched S records earlier: an L1 / L2 hit), ORs each record's flip / reset bit into
// the LDS masks of its series, then turns the flip masks into prefix parities -- a shift-XOR ladder per word, a word-to-word
// carry, and a bit-serial walk only for series that hold a reset -- and writes masks and chunk summaries.
// W = 4: quaternions; W = 8: dual quaternions (sign decided by the real part, rotations/dual_quat.py:139-167).
template <int W>
__global__ __launch_bounds__(PM_WAVE) void unroll_mask_kernel(const UnrollArgs a) {
    constexpr int V = W / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned *flip = reinterpret_cast<unsigned *>(smem);  // [sb][8]
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x / a.p1_blocks;
    const int s0 = (blockIdx.x - chunk * a.p1_blocks) * a.p1_sb;
    const int sb = (a.S - s0) < a.p1_sb ? (a.S - s0) : a.p1_sb;
    unsigned *rst = flip + sb * UR_WORDS;                  // [sb][8]
    const int64_t t0 = (int64_t)chunk * UR_CHUNK;
    const int64_t t1 = (t0 + UR_CHUNK) < a.T ? (t0 + UR_CHUNK) : a.T;
    for (int i = lane; i < 2 * sb * UR_WORDS; i += PM_WAVE) flip[i] = 0u;
    wave_sync();
    const int n = (int)(t1 - t0) * sb;  // records of this chunk x series block (< 2^22)
    const float inv_sb = 1.0f / (float)sb;
    const v4f *q = reinterpret_cast<const v4f *>(a.q);
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
            const bool f = ok[u] && d < 0.0f;
            const bool z = ok[u] && !(d < 0.0f) && !(d > 0.0f);  // 0, -0 or NaN: the reference's `d0 < d1` is false whatever came before
            const int w = ser[u] * UR_WORDS + (row[u] >> 5);
            const unsigned bit = 1u << (row[u] & 31);
            if (f) __hip_atomic_fetch_or(flip + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (z) __hip_atomic_fetch_or(rst + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    wave_sync();
    for (int c = lane; c < sb; c += PM_WAVE) {  // lane = series
        unsigned *fw = flip + c * UR_WORDS, *rw = rst + c * UR_WORDS;
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < UR_WORDS; ++k) any |= rw[k];
        int par = 0;  // parity entering the next word
        if (any == 0) {
#pragma unroll
            for (int k = 0; k < UR_WORDS; ++k) {
                unsigned x = fw[k];
                x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;  // bit r = parity of bits 0..r
                x ^= par ? 0xffffffffu : 0u;
                fw[k] = x;
                par = (int)(x >> 31);
            }
        } else {  // resets in this chunk: bit-serial (rare)
            int seen = 0;
            for (int k = 0; k < UR_WORDS; ++k) {
                const unsigned x = fw[k], z = rw[k];
                unsigned o = 0u, ab = 0u;
                for (int b = 0; b < 32; ++b) {
                    if ((z >> b) & 1u) { par = 0; seen = 1; }
                    else par ^= (int)((x >> b) & 1u);
                    o |= (unsigned)par << b;
                    ab |= (unsigned)seen << b;
                }
                fw[k] = o;
                rw[k] = ab;
            }
        }
        // the frames past the end of a short last chunk carry bit (t1 - t0 - 1)'s parity forward (flip bits there are 0)
        a.sum[(int64_t)chunk * a.S + s0 + c] = par | (any ? 2 : 0);
        uint32_t *gp = a.pre + ((int64_t)chunk * a.S + s0 + c) * UR_WORDS, *ga = a.abs + ((int64_t)chunk * a.S + s0 + c) * UR_WORDS;
#pragma unroll
        for (int k = 0; k < UR_WORDS; ++k) { gp[k] = fw[k]; if (any) ga[k] = rw[k]; }
    }
}

// pass 3: a pure stream.  One dwordx4 per lane, consecutive lanes on consecutive dwordx4; the sign of a record is
// one bit of `pre`, XOR the parity entering its chunk unless a reset inside the chunk came first -- two or three cached
// 4-byte loads (the masks total T S / 4 bytes) next to the 16 bytes streamed each way.
template <int W>
__global__ __launch_bounds__(256) void unroll_apply_kernel(const UnrollArgs a) {
    constexpr int V = W / 4;
    const int64_t nv = a.T * (int64_t)a.S * V;
    const int64_t base = (int64_t)blockIdx.x * 1024;  // 4 dwordx4 per thread
    // record index of the block's first dwordx4 -> (t, s), once; inside the block offsets stay below 2^22
    const int64_t rec0 = base / V;
    const int64_t tb = rec0 / a.S;
    const int sb0 = (int)(rec0 - tb * a.S);
    const float invS = 1.0f / (float)a.S;
    const v4f *src = reinterpret_cast<const v4f *>(a.q);
    v4f *dst = reinterpret_cast<v4f *>(a.out);
    v4f val[4];
    unsigned sg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        const int64_t ic = i < nv ? i : nv - 1;
        val[u] = __builtin_nontemporal_load(src + ic);
        const int rl = (int)((ic - rec0 * V) / V) + sb0;          // record offset from (tb, 0)
        const int dt = (int)(((float)rl + 0.5f) * invS);           // rl / S, exact below 2^22
        const int s_ = rl - dt * a.S;
        const int64_t t = tb + dt;
        const int64_t cs = (t >> 8) * a.S + s_;
        const int r = (int)(t & 255);
        const int sm = a.sum[cs];
        unsigned bit = (a.pre[cs * UR_WORDS + (r >> 5)] >> (r & 31)) & 1u;
        unsigned use_carry = 1u;
        if (sm & 2) use_carry = ((a.abs[cs * UR_WORDS + (r >> 5)] >> (r & 31)) & 1u) ^ 1u;  // rare
        bit ^= (unsigned)(sm & 1) & use_carry;
        sg[u] = bit << 31;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < nv) {
            v4f o;
            o.x = __uint_as_float(__float_as_uint(val[u].x) ^ sg[u]); o.y = __uint_as_float(__float_as_uint(val[u].y) ^ sg[u]);
            o.z = __uint_as_float(__float_as_uint(val[u].z) ^ sg[u]); o.w = __uint_as_float(__float_as_uint(val[u].w) ^ sg[u]);
            __builtin_nontemporal_store(o, dst + i);
        }
    }
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
            if (c < nchunks) ws[(int64_t)c * S + s] = (cut[g] ? pre[g] : (pre[g] ^ cin)) | (v[g] & 2);  // bit 1 (the chunk holds a reset) stays
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
        ws[(int64_t)c * S + s] = carry | (v & 2);
        carry = (v & 2) ? (v & 1) : (carry ^ (v & 1));
    }
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_quat_unroll_workspace_bytes(int64_t T, int32_t S) {
    if (T <= 0 || S <= 0) return 0;
    return ((T + UR_CHUNK - 1) / UR_CHUNK) * (int64_t)S * (int64_t)sizeof(int32_t) * (1 + 2 * UR_WORDS);
}

template <int W>
static int unroll_launch(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    PM_CHECK_ARGS(T >= 0 && S >= 0, "quat_unroll: negative size");
    if (T == 0 || S == 0) return PM_OK;
    PM_CHECK_ARGS(q && out && workspace, "quat_unroll: null pointer");
    PM_CHECK_ARGS(aligned16(q) && aligned16(out), "quat_unroll: q and out must be 16-byte aligned");
    const int64_t nchunks = (T + UR_CHUNK - 1) / UR_CHUNK;
    // mask pass: series per block from the size of the grid it leaves (>= ~8 K waves if the problem has them)
    int p1_sb = UR_P1_SB;
    while (p1_sb > 64 && nchunks * ((S + p1_sb - 1) / p1_sb) < 8192) p1_sb >>= 1;
    const int64_t p1_blocks = (S + p1_sb - 1) / p1_sb;
    const int64_t nv = T * (int64_t)S * (W / 4), ablocks = (nv + 1023) / 1024;
    if (nchunks * p1_blocks > 0x7fffffffLL || ablocks > 0x7fffffffLL) { set_error("quat_unroll: problem too large"); return PM_EUNSUPPORTED; }
    UnrollArgs a;
    a.q = q; a.out = out; a.T = T; a.S = S; a.nchunks = (int)nchunks; a.p1_sb = p1_sb; a.p1_blocks = (int)p1_blocks;
    a.sum = static_cast<int32_t *>(workspace);
    a.pre = reinterpret_cast<uint32_t *>(a.sum + nchunks * S);
    a.abs = a.pre + nchunks * S * UR_WORDS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        const size_t p1_lds = 2 * (size_t)(S < p1_sb ? S : p1_sb) * UR_WORDS * sizeof(unsigned);
        PM_SET_LDS(p1_lds);
        hipLaunchKernelGGL((unroll_mask_kernel<W>), dim3((unsigned)(nchunks * p1_blocks)), dim3(PM_WAVE), p1_lds, s, a);
    }
    if (nchunks <= 256) hipLaunchKernelGGL(unroll_scan_wide_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, a.sum, (int)nchunks, (int)S);
    else hipLaunchKernelGGL(unroll_scan_kernel, dim3((unsigned)S), dim3(1024), 0, s, a.sum, (int)nchunks, (int)S);
    hipLaunchKernelGGL((unroll_apply_kernel<W>), dim3((unsigned)ablocks), dim3(256), 0, s, a);
    return PM_AFTER_LAUNCH("quat_unroll");
}

extern "C" int pm_quat_unroll_f32(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<4>(q, T, S, out, workspace, stream);
}
extern "C" int pm_dq_unroll_f32(const float *dq, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<8>(dq, T, S, out, workspace, stream);
}
