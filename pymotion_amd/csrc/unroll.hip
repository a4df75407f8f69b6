// unroll.hip -- quat.unroll (pymotion/rotations/quat.py:426-462) for gfx950: the first frame-COUPLED op.
//
// Reference: walk the unroll axis; flip quaternion i when d0 = dot(q_i, q'_{i-1}) < d1 = -d0, q'_{i-1} being the
// already-corrected one.  With s_i the sign frame i ends up with (s_0 = +1) and dot_i the dot product of the ORIGINAL
// neighbours, d0 = s_{i-1} dot_i, so
//     dot_i > 0: s_i = s_{i-1}      dot_i < 0: s_i = -s_{i-1}      dot_i == 0 or NaN: s_i = +1   (`d0 < d1` is false)
// i.e. every frame applies one of three maps to the running sign -- keep, negate, RESET to + -- and maps of that kind
// compose associatively (affine maps over GF(2): (a, b): s -> a s xor b; keep = (1,0), negate = (1,1), reset = (0,0)).
// Without resets that is a prefix XOR of the flip bits; a reset (a zero-padded row, an exactly orthogonal step, a NaN)
// forgets everything before it.  A scan, not a loop.  Two forms:
//   * at most 64 series (a clip, or a few): ONE kernel, a decoupled look-back scan -- 16 B read and 16 B written per
//     quaternion, see unroll_onepass_kernel below (2^20 x 22: 151 us; a 65 536-frame clip: 19 us against 79 us); batches of
//     clips [B, T, S, 4] are B independent chains in the same launch (16 384 clips of 64 frames x 22: 123 us);
//   * wide batches (S > 64), three passes:
//   pass 1  (unroll_mask_kernel) each wave streams a chunk of 256 consecutive frames (one record per lane, the
//           predecessor row an L1 / L2 hit), ORs flip / reset bits into per-series LDS masks and turns them into PREFIX
//           parities relative to the chunk's entry (a shift-XOR ladder per 32 frames); masks + a 2-bit chunk summary
//           per series go to the workspace (T S / 4 bytes in all);
//   pass 2  exclusive prefix of the chunk summaries (composition of keep / negate / reset maps);
//   pass 3  (unroll_apply_kernel) a pure stream: a record's sign is one mask bit XOR its chunk's entering parity.
// Round 1 staged 64-frame sub-tiles of the rows in LDS in pass 3 and re-derived the flips there with ballots (185 us of
// the 254 at 2^20 x 22); as a stream it runs at the element-wise kernels' rate.
// Layout: q [T, S, 4] (unroll axis first; the front-end moves it there), out same.
// Algorithmic HBM bytes: 32 B per quaternion (one pass); the three-pass form moves 16 (pass 1) + 16 + 16 (pass 3) = 48 B.
#include "common.hpp"
#include "trig.hpp"

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


// ---- one pass (S <= 64 series: a clip, or a few) ------------------------------------------------------------------------
// The three passes above move 48 B per quaternion for a 32 B operation.  With at most 64 series the whole state of the scan
// at any frame is ONE 64-bit word (the running sign of every series), so the chunks can be chained inside a single kernel
// (a decoupled look-back scan): a 256-thread workgroup takes the next tile of 256 R consecutive dwordx4 -- the ticket
// counter hands tiles out in the order workgroups START, so every tile a workgroup may wait for is already running --
// keeps its records in registers, builds the flip / reset masks of the tile in LDS like the mask pass does, publishes the
// tile's map {parity it adds per series, series it resets} and then looks back over its predecessors, 64 tiles per
// load: the nearest tile whose ENTERING + own state is already known ends the search, the maps of the tiles in between
// compose on top.  Signs are applied to the registers and the tile leaves: 16 B read, 16 B written per dwordx4.
// Workspace: uint32 ticket (own 64 B line), then per tile one 64-bit status WORD per group of 31 series, zeroed by a small
// kernel ahead of the launch:   bits 0..30 parity, bits 31..61 reset mask, bits 62..63 state -- 0 = nothing yet, 1 = the tile's
// own map {parity it adds, series it resets}, 2 = parity LEAVING the tile (bits 0..30).  A word is written and read with one
// relaxed agent-scope atomic and is self-contained, so no fence orders anything (a release / acquire pair at agent scope
// writes back / invalidates the XCD's L2, per tile).
constexpr int OP_GROUP = 31;
#ifndef PM_OP_MINW
#define PM_OP_MINW 1  // waves per SIMD the compiler must fit (no cap needed: 95 VGPRs once the record indices stopped living across the phases)
#endif

struct OnePassArgs {
    const float *q;
    float *out;
    uint32_t *ticket;
    unsigned long long *st;   // [nclips][tpc][ngroups]    level 0: one word per tile
    unsigned long long *st1;  // [nclips][bpc][ngroups]    level 1: one word per aligned block of 64 tiles of a clip
    int64_t nv;        // dwordx4 per clip
    int64_t nclips, tpc, bpc;  // clips [B][T][S][W] scanned independently: tiles and 64-tile blocks per clip
    int32_t S, words;  // mask words per series: ceil(rows / 32), rows = frames a tile can touch
    int32_t ngroups;   // ceil(S / 31)
    int32_t static_order;  // PM_TUNING build only: tiles by blockIdx (what the ticket costs)
    int32_t single;        // every clip is ONE tile: nobody waits for anybody, tiles by blockIdx and no ticket
    unsigned long long *zero_next;  // pm_unroll_onepass_f32: the OTHER workspace of the caller's pair, zeroed on the way (null: nothing)
    int64_t n_zero_next, ntiles;    // its words in use; workgroups of this launch (each zeroes a slice)
    uint32_t order_pk[64]; // EULER source (pm_bvh_rotations_f32): per series the three axis codes of its Euler order, o0 | o1 << 8 | o2 << 16
};

// ---- look-back ----
// The tiles in flight ahead of a tile publish their own maps long before any of them knows its entering parity, so the
// nearest tile with a LEAVING parity is typically several hundred tiles back while the chip finishes 40-80 tiles per
// microsecond: walking back 64 tiles per memory round trip never catches up (measured: 45 of 180 us, and fetching eight
// windows per round trip instead only adds traffic on the few channels that hold the status words).  So the words form
// a two-level skip list: the last tile of every aligned block of 64 tiles also publishes the block's map (level 1), and a
// tile looks at the tiles of its own block before it (<= 63 level-0 words) and at the 64 blocks before that (4096 tiles)
// in ONE round trip.
__device__ __forceinline__ unsigned wave_xor(unsigned x) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x ^= (unsigned)__shfl_xor((int)x, m, 64);
    return x;
}

// One window of status words, lane 0 nearest: the map of lanes 0 .. min(L, n - 1), L = nearest lane with a leaving parity
// (64: none).  Returns false while a needed word is still empty.  (p, r) then (p', r')  =  ((p & ~r') ^ p', r | r').
__device__ __forceinline__ bool fold_window(const unsigned long long w, const int n, const int lane, unsigned &cp, unsigned &cr, bool &absolute) {
    const bool in = lane < n;
    const unsigned long long m_inc = __builtin_amdgcn_ballot_w64(in && (w >> 62) == 2ull), m_zero = __builtin_amdgcn_ballot_w64(in && (w >> 62) == 0ull);
    const int L = m_inc ? __builtin_ctzll(m_inc) : 64;
    const unsigned long long near = (L < 64) ? ((1ull << L) - 1ull) : ~0ull;
    if ((m_zero & near) != 0ull) return false;
    const unsigned par = (in && lane <= L) ? (unsigned)w & 0x7fffffffu : 0u;  // tiles beyond the leaving parity do not matter
    const unsigned rst = (in && lane < L) ? (unsigned)(w >> 31) & 0x7fffffffu : 0u;
    if (__builtin_amdgcn_ballot_w64(rst != 0u) == 0ull) {
        cp = wave_xor(par);
        cr = 0u;
    } else {  // resets among the nearer tiles: in order (rare)
        const int last = L < 64 ? L : (n < 64 ? n - 1 : 63);
        cp = (unsigned)__builtin_amdgcn_readlane((int)par, last);
        cr = (unsigned)__builtin_amdgcn_readlane((int)rst, last);
        for (int l = last - 1; l >= 0; --l) {
            const unsigned pl = (unsigned)__builtin_amdgcn_readlane((int)par, l), rl = (unsigned)__builtin_amdgcn_readlane((int)rst, l);
            cp = (cp & ~rl) ^ pl;
            cr |= rl;
        }
    }
    absolute = L < 64;
    if (absolute) cr = 0x7fffffffu;  // a leaving parity is absolute
    return true;
}

// Parity of the 31 series of group `grp` ENTERING tile `tile` (wave-uniform; all 64 lanes of one wave call this).  `own_p`,
// `own_r`: this tile's map; the last tile of a block publishes the block's map on the way.
__device__ __forceinline__ unsigned unroll_look_back(unsigned long long *st0, unsigned long long *st1, const int ngroups, const int grp, const int64_t tile,
                                                     const unsigned own_p, const unsigned own_r, const int lane) {
    constexpr unsigned long long kBefore = 2ull << 62;  // before tile 0: every series enters with +
    const int pos = (int)(tile & 63);                   // tiles of the own block before this one
    const int64_t blk = tile >> 6;
    const unsigned long long *p0 = st0 + (tile - 1 - lane) * ngroups + grp;
    unsigned long long w0 = 1ull << 62, w1;
    unsigned c0p = 0u, c0r = 0u;
    bool abs0 = false;
    {
        const int64_t b = blk - 1 - lane;
        if (lane < pos) w0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1 = b >= 0 ? __hip_atomic_load(st1 + b * ngroups + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kBefore;
    }
    while (!fold_window(w0, pos, lane, c0p, c0r, abs0)) {
        __builtin_amdgcn_s_sleep(1);
        if (lane < pos) w0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pos == 63 && lane == 0) {  // the block's map, for everybody behind: an absolute parity if the walk ended inside the block
        const unsigned bp = (c0p & ~own_r) ^ own_p, br = c0r | own_r;
        const unsigned long long word = abs0 ? ((2ull << 62) | bp) : ((1ull << 62) | ((unsigned long long)br << 31) | bp);
        __hip_atomic_store(st1 + blk * ngroups + grp, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (abs0) return c0p;
    unsigned acc_par = c0p, acc_rst = c0r;
    for (int64_t hi = blk - 1;; hi -= 64) {
        unsigned cp = 0u, cr = 0u;
        bool abs1 = false;
        const int64_t b = hi - lane;
        if (hi != blk - 1) w1 = b >= 0 ? __hip_atomic_load(st1 + b * ngroups + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kBefore;
        while (!fold_window(w1, 64, lane, cp, cr, abs1)) {
            __builtin_amdgcn_s_sleep(1);
            if (b >= 0) w1 = __hip_atomic_load(st1 + b * ngroups + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned np = (cp & ~acc_rst) ^ acc_par, nr = cr | acc_rst;
        acc_par = np; acc_rst = nr;
        if (abs1) return acc_par;
    }
}

// ticket + status words back to zero ahead of every scan (a kernel of our own rather than a memset node: under HIP graph
// replay the memset node of ROCm 7.0 faulted on the second replay, tests/test_gpu_parity.py)
__global__ __launch_bounds__(256) void unroll_reset_kernel(unsigned long long *w, const int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = 0ull;
}

// EULER (W = 4): the fused BVH ingest of io/bvh.py:352-359 -- quat.normalize(quat.unroll(quat.from_euler(radians(rotations), order), axis = 0)) in
// ONE pass: `q` holds Euler angles in DEGREES, [T][S][3] (12 B per record: consecutive lanes on consecutive records, dwordx3), the
// load stage turns each record into its unit quaternion (trig.hpp: euler2q with the series' order from a table in LDS, then
// q / (|q| + 1e-8), quat.py:423 -- a positive scale commutes with the sign decision) and everything downstream is the scan as
// it stands: 12 B read + 16 B written per joint and frame where the three launches move 28 + 32 + 32.
template <int W, int R, int NT, bool EULER = false>
__global__ __launch_bounds__(NT, PM_OP_MINW) void unroll_onepass_kernel(const OnePassArgs a) {
    static_assert(!EULER || W == 4, "the Euler source produces quaternions");
    constexpr int V = W / 4, TILE = NT * R;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned s_tile;
    __shared__ unsigned long long s_enter;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.S, words = a.words;
    unsigned *flip = reinterpret_cast<unsigned *>(smem);  // [S][words]  flip bits, then prefix parities
    unsigned *rst = flip + S * words;                      // [S][words]  reset bits, then "a reset at or before"
    unsigned *sOrd = rst + S * words;                      // [S]  EULER: the series' axis codes
    if (EULER && tid < S) sOrd[tid] = a.order_pk[tid];
    if (tid == 0) s_tile = (a.single || PM_ABLATED_FLAG(a.static_order & 1)) ? blockIdx.x : __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < 2 * S * words; i += NT) flip[i] = 0u;
    __syncthreads();
    if (a.zero_next != nullptr) {  // (tickets / block indices 0 .. ntiles - 1, one each: the slices cover the words whatever order the workgroups start in)
        const int64_t per = (a.n_zero_next + a.ntiles - 1) / a.ntiles, lo = (int64_t)s_tile * per, hi = (lo + per) < a.n_zero_next ? (lo + per) : a.n_zero_next;
        for (int64_t i = lo + tid; i < hi; i += NT) a.zero_next[i] = 0ull;
    }
    // tickets run over (clip, tile of the clip): a tile only ever waits for earlier tiles of its own clip, i.e. lower tickets
    const int64_t clip = (int64_t)s_tile / a.tpc;
    if (clip >= a.nclips) return;  // (uniform)
    const int64_t tile = (int64_t)s_tile - clip * a.tpc;
    const int64_t base = tile * TILE;
    const int64_t rec0 = base / V;          // record of the tile's first dwordx4 -> (tb, sb0), once
    const int64_t tb = rec0 / S;
    const int sb0 = (int)(rec0 - tb * S);
    const float invS = 1.0f / (float)S;
    const v4f *src = reinterpret_cast<const v4f *>(a.q) + clip * a.nv;
    v4f *dst = reinterpret_cast<v4f *>(a.out) + clip * a.nv;
    unsigned long long *st0 = a.st + clip * a.tpc * a.ngroups, *st1 = a.st1 + clip * a.bpc * a.ngroups;

    // A wave owns 64 R consecutive dwordx4 of the tile (row u: 64 of them, one per lane), so the record one frame earlier --
    // D = S V dwordx4 back -- sits in the SAME wave's registers, D lanes to the left in this row or, for the first D lanes, at
    // the far end of the row before: one select and four ds_bpermute per dwordx4 instead of a second (L2) load of every record
    // (measured at 2^20 x 22: 57 of 232 us).  Only the first D dwordx4 of a wave's range are fetched again.  (D > 64 --
    // dual quaternions of more than 32 series -- keeps the second load.)
    const int D = S * V;
    const bool shuffled = D <= 64 && !PM_ABLATED_FLAG(a.static_order & 8);
    // (indices inside the wave's range stay 32-bit and nothing per record is kept besides the record itself: sixteen records
    // per thread took 178 VGPRs before this, 95 now -- see lane_b / lane_c below)
    const int64_t wbase = base + (int64_t)wave * (64 * R);
    const int left = (int)((a.nv - wbase) < (int64_t)(64 * R) ? (a.nv - wbase > 0 ? a.nv - wbase : 0) : (int64_t)(64 * R));  // dwordx4 of this wave's range that exist
    const v4f *wsrc = src + wbase;
    const int lastoff = (int)(a.nv - 1 - wbase);  // the array's last dwordx4 seen from this range (negative for a wave past the end): what idle lanes read
    const int rl0 = (int)((wbase - rec0 * V) / V) + sb0;  // record offset of the range's first dwordx4 from (tb, 0); V divides wbase
    auto locate = [&](const int li, int &dt, int &s_) {  // dwordx4 li of the range -> (row, series)
        const int rl = rl0 + (V == 1 ? li : li >> 1);
        dt = (int)(((float)rl + 0.5f) * invS);  // rl / S, exact below 2^22
        s_ = rl - dt * S;
    };
    // dwordx4 `li` of the wave's range as the scan sees it: the stored quaternion, or (EULER) the unit quaternion of the record's angles
    // (s_: the record's series -- the record one frame earlier, which may lie ahead of the tile, has the same)
    auto raw3 = [&](const int li) { return __builtin_nontemporal_load(reinterpret_cast<const v3f_a4 *>(a.q + (clip * a.nv + wbase + li) * 3)); };  // (sizeof(v3f_a4) is 16: index in floats)
    auto convert = [&](const v3f_a4 d, const int s_) -> v4f {
        const unsigned pk = sOrd[s_];
        const int o[3] = {(int)(pk & 0xffu), (int)((pk >> 8) & 0xffu), (int)((pk >> 16) & 0xffu)};
        // np.radians in float64 like the reference's (bvh.py:353), rounded once
        const float e[3] = {(float)((double)d.x * 0.017453292519943295), (float)((double)d.y * 0.017453292519943295), (float)((double)d.z * 0.017453292519943295)};
        float q[4], qn[4];
        euler2q<false>(e, o, q);  // (no libm detour beyond 1e8 rad = 5.7e9 degrees: nobody's channel value; NaN / Inf still give NaN)
        qnormalize(q, 1e-8f, qn);
        return v4f{qn[0], qn[1], qn[2], qn[3]};
    };
    v4f val[R];
    if constexpr (!EULER) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int li = u * 64 + lane;
            val[u] = __builtin_nontemporal_load(wsrc + (li < left ? li : lastoff));
        }
    } else {
        // every load of the wave's range first (one memory latency), then the conversions, two at a time
        v3f_a4 raw[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int li = u * 64 + lane;
            raw[u] = raw3(li < left ? li : lastoff);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int li = u * 64 + lane;
            int dt0, s0_;
            locate(li < left ? li : lastoff, dt0, s0_);
            val[u] = convert(raw[u], s0_);
            if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // (the lane id through an opaque copy per phase: left to itself the compiler keeps the sixteen record indices of the load
    // phase -- and their clamps -- alive to the last store, ~30 registers of the 164)
    int lane_b = lane, lane_c = lane;
    asm volatile("" : "+v"(lane_b));
    const int from = ((lane - D) & 63) << 2;  // ds_bpermute address of the lane D to the left
    const bool own = lane < 64 - D;           // lanes that hand their own row to the right; the rest hand the previous row to the head of this one
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int li = u * 64 + lane_b;
        int dt, s_;
        locate(li, dt, s_);
        const bool ok = (li < left) && (tb + dt > 0) && (V == 1 || (li & 1) == 0);  // the real part decides (dual_quat.py:139-167)
        const bool outside = !shuffled || (u == 0 && lane < D);  // the row before lies ahead of this wave's range
        v4f prv = val[u];
        if (outside && ok && !PM_ABLATED_FLAG(a.static_order & 2)) {  // (L1 / L2: fetched 16 S bytes earlier by a neighbour)
            if constexpr (EULER) prv = convert(raw3(li - D), s_);
            else prv = wsrc[li - D];
        }
        if (shuffled) {
            const v4f mine = val[u], before = val[u > 0 ? u - 1 : 0];
            v4f got;
            got.x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(own ? mine.x : before.x)));
            got.y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(own ? mine.y : before.y)));
            got.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(own ? mine.z : before.z)));
            got.w = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(own ? mine.w : before.w)));
            if (!(u == 0 && lane < D)) prv = got;
        }
        const v4f c = val[u];
        const float d = c.x * prv.x + c.y * prv.y + c.z * prv.z + c.w * prv.w;
        const bool f = ok && d < 0.0f;
        const bool z = ok && !(d < 0.0f) && !(d > 0.0f);  // 0, -0 or NaN: reset (see the top of the file)
        const int w = s_ * words + (dt >> 5);
        const unsigned bit = 1u << (dt & 31);
        if (f) __hip_atomic_fetch_or(flip + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (z) __hip_atomic_fetch_or(rst + w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two records in flight, not sixteen (the scheduler would hoist every exchange: 164 VGPRs)
    }
    __syncthreads();
    if (wave == 0) {
        int par = 0;
        unsigned any = 0u;
        if (lane < S) {  // lane = series: flips -> prefix parities relative to the tile's entry
            unsigned *fw = flip + lane * words, *rw = rst + lane * words;
            for (int k = 0; k < words; ++k) any |= rw[k];
            if (any == 0u) {
                for (int k = 0; k < words; ++k) {
                    unsigned x = fw[k];
                    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
                    x ^= par ? 0xffffffffu : 0u;
                    fw[k] = x;
                    par = (int)(x >> 31);
                }
            } else {  // resets in this tile: bit-serial (rare)
                int seen = 0;
                for (int k = 0; k < words; ++k) {
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
        }
        const unsigned long long agg_par = __builtin_amdgcn_ballot_w64(par != 0), agg_rst = __builtin_amdgcn_ballot_w64(any != 0u);
        unsigned long long *me = st0 + tile * a.ngroups;
        for (int g = 0; g < a.ngroups; ++g) {
            const unsigned long long p = (agg_par >> (OP_GROUP * g)) & 0x7fffffffull, r = (agg_rst >> (OP_GROUP * g)) & 0x7fffffffull;
            if (lane == 0 && !a.single) __hip_atomic_store(me + g, (1ull << 62) | (r << 31) | p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a clip of one tile: nobody reads its words, and a caller's workspace pair stays clean)
        }
        unsigned long long enter = 0ull;
        for (int g = 0; g < a.ngroups; ++g) {
            const unsigned long long p = (agg_par >> (OP_GROUP * g)) & 0x7fffffffull, r = (agg_rst >> (OP_GROUP * g)) & 0x7fffffffull;
            const unsigned long long e = PM_ABLATED_FLAG(a.static_order & 4) ? 0ull : unroll_look_back(st0, st1, a.ngroups, g, tile, (unsigned)p, (unsigned)r, lane);
            if (lane == 0 && !a.single) {
                const unsigned long long leaving = (2ull << 62) | ((e & ~r) ^ p);
                __hip_atomic_store(me + g, leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((tile & 63) == 63) __hip_atomic_store(st1 + (tile >> 6) * a.ngroups + g, leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the block's word becomes absolute
            }
            enter |= e << (OP_GROUP * g);
        }
        if (lane == 0) s_enter = enter;
    }
    __syncthreads();
    const unsigned long long enter = s_enter;
    v4f *wdst = dst + wbase;
    asm volatile("" : "+v"(lane_c));
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int li = u * 64 + lane_c;
        int dt, s_;
        locate(li, dt, s_);
        const int w = s_ * words + (dt >> 5);
        unsigned bit = (flip[w] >> (dt & 31)) & 1u;
        const unsigned absolute = (rst[w] >> (dt & 31)) & 1u;  // a reset inside the tile came first: the entering parity does not apply
        bit ^= (unsigned)((enter >> s_) & 1ull) & (absolute ^ 1u);
        const unsigned sg = bit << 31;
        if (li < left) {
            v4f o;
            o.x = __uint_as_float(__float_as_uint(val[u].x) ^ sg); o.y = __uint_as_float(__float_as_uint(val[u].y) ^ sg);
            o.z = __uint_as_float(__float_as_uint(val[u].z) ^ sg); o.w = __uint_as_float(__float_as_uint(val[u].w) ^ sg);
            __builtin_nontemporal_store(o, wdst + li);
        }
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_quat_unroll_batched_workspace_bytes(int64_t B, int64_t T, int32_t S) {
    if (B <= 0 || T <= 0 || S <= 0) return 0;
    // three passes (S > 64; clips one after the other through the same workspace)
    const int64_t three = ((T + UR_CHUNK - 1) / UR_CHUNK) * (int64_t)S * (int64_t)sizeof(int32_t) * (1 + 2 * UR_WORDS);
    // one pass (S <= 64): ticket line + per clip a status word per tile of 1024 dwordx4 (the smaller tile; dual quaternions: 2 per
    // record) and per block of 64 tiles, for every group of 31 series
    const int64_t t1 = (T * S * 2 + 1023) / 1024, one = 64 + B * (t1 + (t1 + 63) / 64) * (int64_t)sizeof(unsigned long long) * ((S + OP_GROUP - 1) / OP_GROUP);
    return three > one ? three : one;
}
extern "C" int64_t pm_quat_unroll_workspace_bytes(int64_t T, int32_t S) { return pm_quat_unroll_batched_workspace_bytes(1, T, S); }

// `pair` (pm_unroll_onepass_f32): `workspace` is all zero on entry -- no reset launch in front of the scan --, *pair->dirtied receives the 8-byte
// words this call leaves non-zero in it, and the first pair->other_words words of pair->other are zeroed on the way (a slice per workgroup).
struct UnrollPair { int64_t *dirtied; void *other; int64_t other_words; };
template <int W, bool EULER = false>
static int unroll_launch(const float *q, int64_t B, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream, const uint8_t *order = nullptr,
                         const UnrollPair *pair = nullptr) {
    PM_CHECK_ARGS(B >= 0 && T >= 0 && S >= 0, "quat_unroll: negative size");
    if (pair) *pair->dirtied = 0;
    if (B == 0 || T == 0 || S == 0) {
        // nothing to scan -- but a successful pm_unroll_onepass_f32 promises the other workspace clean: the caller swaps the two on PM_OK
        if (pair && pair->other && pair->other_words > 0)
            hipLaunchKernelGGL(unroll_reset_kernel, dim3((unsigned)((pair->other_words + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                               static_cast<unsigned long long *>(pair->other), pair->other_words);
        return pair ? PM_AFTER_LAUNCH("unroll_onepass") : PM_OK;
    }
    PM_CHECK_ARGS(q && out && workspace, "quat_unroll: null pointer");
    PM_CHECK_ARGS((EULER || aligned16(q)) && aligned16(out), "quat_unroll: q and out must be 16-byte aligned");
    PM_CHECK_ARGS((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "quat_unroll: the workspace must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if constexpr (EULER) {
        PM_CHECK_ARGS(order, "bvh_rotations: null order table");
        if (S > 64) { set_error("bvh_rotations: the one-pass kernel takes at most 64 joints (J = %d): run from_euler, unroll and normalize", S); return PM_EUNSUPPORTED; }
        for (int i = 0; i < 3 * S; ++i) PM_CHECK_ARGS(order[i] <= 2, "bvh_rotations: axis codes are 0, 1, 2 (x, y, z)");
    }
    if (S <= 64 && (EULER || tune_env("PM_UNROLL_ONEPASS", 1))) {
        const int64_t nv = T * (int64_t)S * (W / 4);
        // tile: 4096 dwordx4 (64 KiB; 16 per thread) from 64 such tiles on, 1024 below.  Measured, S = 22, T = 2^10 / 2^12 / 2^14 /
        // 2^16 / 2^18 / 2^20 frames: three passes 17 / 20 / 34 / 79 / 60 / 195 us; one pass with 1024-dwordx4 tiles 9 / 8 / 13 / 32 /
        // 91 / 301 us, with 4096: 12 / 13 / 13 / 19 / 52 / 158 us (2048 / 3072 per tile at 2^20: 187 / 166 us; 512-thread workgroups: +5 %).
        constexpr int NT = 256;
        // Clips of a batch ([B][T][S][W]) are independent chains in one launch.  A clip that fits one tile gets the smallest tile
        // that holds it (1024 / 2048 / 4096 / 8192 dwordx4) and no look-back at all; longer clips the big tile once the launch has 64 of them.
        int Rauto = 16;
        if (nv <= 1024) Rauto = 4;
        else if (nv <= 2048) Rauto = 8;
        else if (nv > 4096 && nv <= 8192 && B >= 64) Rauto = 32;  // still one tile per clip (4096 clips of 256 frames x 22: 146 -> 131 us)
        else if (nv > 4096 && B * ((nv + 4095) / 4096) < 64) Rauto = 4;
        else if (nv >= (int64_t)4 << 20) Rauto = 32;  // very long chains: fewer, bigger tiles (2^18 / 2^20 frames x 22: 51 / 159 -> 46 / 153 us; 2^16: 19.6 -> 21.3)
        else if (B == 1 && nv < ((int64_t)1 << 19)) Rauto = 8;  // a clip of a few hundred thousand records: more, smaller tiles (2^14 x 22: 13.2 -> 11.5 us; 2^16 x 22: 19.3 with 16, 22.7 with 8)
        // (round 4, asked for by the review: 128-thread workgroups -- more look-back chains in flight per CU -- are slower at every length,
        // 2^14 / 2^16 / 2^18 / 2^20 x 22 with R = 16: 13.6 / 24.5 / 69.9 / 214 us against 13.2 / 19.3 / 51.0 / 158.5 us; tools/unroll_nt_sweep.py)
        if constexpr (EULER) {
            // the conversion (three sincos + the Euler product + normalize: ~170 VALU instructions per record) sits between a tile's loads
            // and its map: smaller tiles spread it over more waves.  Measured (S = 22, T = 2^10 ... 2^20 frames, us; R = 4 / 8 / 16 / 32):
            // 9.6 / 12.2 / 18.1 / 31.3,  10.6 / 12.6 / 18.4 / 32,  16.2 / 15.4 / 20.2 / 32.8,  37 / 27.3 / 27.4 / 35,  102 / 79 / 70.6 / 83,
            // 340 / 243 / 210 / 242 (with the long-chain LDS reservation below)
            Rauto = nv < ((int64_t)1 << 18) ? 4 : (nv < ((int64_t)1 << 20) ? 8 : 16);
        }
        const int R = tune_env("PM_UNROLL_R", Rauto);
        if (R != 32 && R != 16 && R != 8 && R != 4) { set_error("PM_UNROLL_R must be 4, 8, 16 or 32"); return PM_EINVAL; }
        const int NTsel = (!EULER && W == 4) ? tune_env("PM_UNROLL_NT", NT) : NT;  // PM_TUNING build only: 128-thread workgroups
        const int64_t tile = NTsel * (int64_t)R, tpc = (nv + tile - 1) / tile, bpc = (tpc + 63) / 64, ntiles = B * tpc;
        if (ntiles > 0x7fffffffLL) { set_error("quat_unroll: problem too large"); return PM_EUNSUPPORTED; }
        OnePassArgs a;
        a.q = q; a.out = out; a.nv = nv; a.S = S; a.nclips = B; a.tpc = tpc; a.bpc = bpc;
        a.ticket = static_cast<uint32_t *>(workspace);
        a.st = reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace) + 64);
        a.ngroups = (S + OP_GROUP - 1) / OP_GROUP;
        a.st1 = a.st + ntiles * a.ngroups;  // B * bpc blocks
        a.static_order = tune_env("PM_UNROLL_STATIC", 0);
        a.single = tpc == 1;
        if constexpr (EULER)
            for (int i = 0; i < S; ++i) a.order_pk[i] = (uint32_t)order[3 * i] | ((uint32_t)order[3 * i + 1] << 8) | ((uint32_t)order[3 * i + 2] << 16);
        const int64_t rows = (tile / (W / 4) + S - 1) / S + 1;  // frames a tile can touch
        a.words = (int)((rows + 31) / 32);
        size_t lds = (2 * (size_t)S * a.words + (EULER ? S : 0)) * sizeof(unsigned);
        // Long chains run best with THREE workgroups per CU: every tile in flight ahead of a tile is a word its look-back has to fold, so
        // more resident tiles lengthen every walk (2^18 x 22 with 5 / 4 / 3 / 2 workgroups per CU: 60.5 / 56.2 / 51.0 / 53.6 us; batches of
        // short clips want them all: 4096 clips of 256 frames 145 / 145 / 153 / 187 us).  The kernel needs 95 VGPRs (five workgroups per
        // CU), so long chains reserve a third of the CU's LDS each.
        // (the Euler source too: without the reservation its 2^20-frame time is bimodal, 218 / 366 / 453 us run to run -- five workgroups per CU
        // make the look-back chains as long as they can get)
        if (tpc > 64 && lds < 52 * 1024 && tune_env("PM_UNROLL_RESERVE", 1)) lds = 52 * 1024;
        lds += (size_t)tune_env("PM_UNROLL_LDS_PAD", 0);  // PM_TUNING build only: more unused LDS
        const int64_t nwords = 8 + (ntiles + B * bpc) * a.ngroups;  // the ticket's 64-byte line, then the words
        a.zero_next = nullptr; a.n_zero_next = 0; a.ntiles = ntiles;
        if (pair) {
            *pair->dirtied = a.single ? 0 : nwords;
            if (pair->other && pair->other_words > 0) { a.zero_next = static_cast<unsigned long long *>(pair->other); a.n_zero_next = pair->other_words; }
        } else if (!a.single)  // (single-tile clips read neither the ticket nor a status word)
            hipLaunchKernelGGL(unroll_reset_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, s, static_cast<unsigned long long *>(workspace), nwords);
        PM_SET_LDS(lds);
        if (NTsel == 128) {
            if constexpr (!EULER && W == 4) {
                if (R == 32) hipLaunchKernelGGL((unroll_onepass_kernel<W, 32, 128, EULER>), dim3((unsigned)ntiles), dim3(128), lds, s, a);
                else if (R == 16) hipLaunchKernelGGL((unroll_onepass_kernel<W, 16, 128, EULER>), dim3((unsigned)ntiles), dim3(128), lds, s, a);
                else if (R == 8) hipLaunchKernelGGL((unroll_onepass_kernel<W, 8, 128, EULER>), dim3((unsigned)ntiles), dim3(128), lds, s, a);
                else hipLaunchKernelGGL((unroll_onepass_kernel<W, 4, 128, EULER>), dim3((unsigned)ntiles), dim3(128), lds, s, a);
                return PM_AFTER_LAUNCH("quat_unroll");
            }
        }
        if (R == 32) hipLaunchKernelGGL((unroll_onepass_kernel<W, 32, NT, EULER>), dim3((unsigned)ntiles), dim3(NT), lds, s, a);
        else if (R == 16) hipLaunchKernelGGL((unroll_onepass_kernel<W, 16, NT, EULER>), dim3((unsigned)ntiles), dim3(NT), lds, s, a);
        else if (R == 8) hipLaunchKernelGGL((unroll_onepass_kernel<W, 8, NT, EULER>), dim3((unsigned)ntiles), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL((unroll_onepass_kernel<W, 4, NT, EULER>), dim3((unsigned)ntiles), dim3(NT), lds, s, a);
        return PM_AFTER_LAUNCH("quat_unroll");
    }
    if (pair) { set_error("unroll_onepass: more than 64 series take the three-pass scan (pm_quat_unroll_f32 / pm_dq_unroll_f32)"); return PM_EUNSUPPORTED; }
    const int64_t nchunks = (T + UR_CHUNK - 1) / UR_CHUNK;
    // mask pass: series per block from the size of the grid it leaves (>= ~8 K waves if the problem has them)
    int p1_sb = UR_P1_SB;
    while (p1_sb > 64 && nchunks * ((S + p1_sb - 1) / p1_sb) < 8192) p1_sb >>= 1;
    const int64_t p1_blocks = (S + p1_sb - 1) / p1_sb;
    const int64_t nv = T * (int64_t)S * (W / 4), ablocks = (nv + 1023) / 1024;
    if (nchunks * p1_blocks > 0x7fffffffLL || ablocks > 0x7fffffffLL) { set_error("quat_unroll: problem too large"); return PM_EUNSUPPORTED; }
    UnrollArgs a;
    a.T = T; a.S = S; a.nchunks = (int)nchunks; a.p1_sb = p1_sb; a.p1_blocks = (int)p1_blocks;
    a.sum = static_cast<int32_t *>(workspace);
    a.pre = reinterpret_cast<uint32_t *>(a.sum + nchunks * S);
    a.abs = a.pre + nchunks * S * UR_WORDS;
    for (int64_t b = 0; b < B; ++b) {  // wide clips of a batch: one after the other (stream order makes the shared workspace safe)
        a.q = q + b * T * (int64_t)S * W; a.out = out + b * T * (int64_t)S * W;
        {
            const size_t p1_lds = 2 * (size_t)(S < p1_sb ? S : p1_sb) * UR_WORDS * sizeof(unsigned);
            PM_SET_LDS(p1_lds);
            hipLaunchKernelGGL((unroll_mask_kernel<W>), dim3((unsigned)(nchunks * p1_blocks)), dim3(PM_WAVE), p1_lds, s, a);
        }
        if (nchunks <= 256) hipLaunchKernelGGL(unroll_scan_wide_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, a.sum, (int)nchunks, (int)S);
        else hipLaunchKernelGGL(unroll_scan_kernel, dim3((unsigned)S), dim3(1024), 0, s, a.sum, (int)nchunks, (int)S);
        hipLaunchKernelGGL((unroll_apply_kernel<W>), dim3((unsigned)ablocks), dim3(256), 0, s, a);
    }
    return PM_AFTER_LAUNCH("quat_unroll");
}

extern "C" int pm_quat_unroll_f32(const float *q, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<4>(q, 1, T, S, out, workspace, stream);
}
extern "C" int pm_dq_unroll_f32(const float *dq, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<8>(dq, 1, T, S, out, workspace, stream);
}
extern "C" int pm_quat_unroll_batched_f32(const float *q, int64_t B, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<4>(q, B, T, S, out, workspace, stream);
}
extern "C" int pm_dq_unroll_batched_f32(const float *dq, int64_t B, int64_t T, int32_t S, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<8>(dq, B, T, S, out, workspace, stream);
}

// io/bvh.py:352-359 get_data: quat.normalize(quat.unroll(quat.from_euler(np.radians(rotations), order), axis = 0)) as ONE launch.
extern "C" int pm_bvh_rotations_f32(const float *euler_deg, const uint8_t *order, int64_t T, int32_t J, float *out, void *workspace, pm_stream_t stream) {
    return unroll_launch<4, true>(euler_deg, 1, T, J, out, workspace, stream, order);
}

// The one-pass scans without the reset launch in front of them (a clip of real length is 7-19 us of which that launch is ~3): the caller owns a
// PAIR of workspaces per stream, both zeroed once (pm_memset), and alternates them -- `ws_zeroed` is the clean one, `ws_other` the one the call
// before dirtied: this launch zeroes its first `ws_other_words` words on the way (every workgroup a slice) and reports in *ws_words_dirtied how
// many 8-byte words of `ws_zeroed` it leaves non-zero.  kind: 0 quat.unroll, 1 dual_quat.unroll (batched: [B, T, S, 4 | 8]), 2 the BVH ingest
// (B = 1, `order` as in pm_bvh_rotations_f32).  At most 64 series (PM_EUNSUPPORTED beyond: the three-pass scan has no reset to save).
extern "C" int pm_unroll_onepass_f32(int32_t kind, const float *in, const uint8_t *order, int64_t B, int64_t T, int32_t S, float *out, void *ws_zeroed,
                                     int64_t *ws_words_dirtied, void *ws_other, int64_t ws_other_words, pm_stream_t stream) {
    PM_CHECK_ARGS(ws_words_dirtied && ws_other_words >= 0 && (ws_other || ws_other_words == 0), "unroll_onepass: bad workspace pair");
    PM_CHECK_ARGS(ws_other == nullptr || (reinterpret_cast<uintptr_t>(ws_other) & 7) == 0, "unroll_onepass: the workspaces must be 8-byte aligned");
    PM_CHECK_ARGS(ws_other == nullptr || ws_other != ws_zeroed, "unroll_onepass: the two workspaces must differ");
    const UnrollPair pair = {ws_words_dirtied, ws_other, ws_other_words};
    if (S > 64) { *ws_words_dirtied = 0; set_error("unroll_onepass: more than 64 series take the three-pass scan"); return PM_EUNSUPPORTED; }
    switch (kind) {
        case 0: return unroll_launch<4>(in, B, T, S, out, ws_zeroed, stream, nullptr, &pair);
        case 1: return unroll_launch<8>(in, B, T, S, out, ws_zeroed, stream, nullptr, &pair);
        case 2: PM_CHECK_ARGS(B == 1, "unroll_onepass: the BVH ingest takes one clip"); return unroll_launch<4, true>(in, 1, T, S, out, ws_zeroed, stream, order, &pair);
        default: set_error("unroll_onepass: kind must be 0, 1 or 2"); return PM_EINVAL;
    }
}
