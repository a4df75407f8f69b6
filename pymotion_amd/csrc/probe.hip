// probe.hip -- store-pattern probe (measurement helper, not part of the reference's surface).
//
// Three quarters of fk's bytes are stores and the LDS-free pure-write stream of round 3 read 4.5 TB/s (56 % of the HBM spec)
// where pure reads reach 6.7 TB/s.  Is that the chip or the pattern?  This kernel writes the same bytes under every knob the
// store path offers, one knob at a time (tools/store_probe.py -> profiles/r04_store_patterns.txt):
//   * per-wave contiguous burst: BURST KiB = BURST back-to-back dwordx4 stores of one wave (1 KiB each);
//   * cache policy of the store: plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt / sc0 (inline asm: the builtins only reach plain and nt),
//     plus buffer_store_dwordx4 plain / nt;
//   * chunk -> address placement: where the chunks of the workgroups of one XCD (workgroup b runs on XCD b % 8) lie:
//     linear (neighbouring chunks on different XCDs), one contiguous eighth per XCD (xcd_tile, what every kernel of this
//     library does), or XCD-interleaved runs of `granule` chunks;
//   * workgroup size (64 / 256 threads), persistent grid-stride vs one chunk per wave;
//   * an optional read in front of the stores (rd4 dwordx4 per lane and chunk, nt loads: fk's 1 : 3 mix);
//   * one output stream or two (fk writes `rotmats` and `pos`: 3/4 + 1/4 of the chunk to two arrays), or the two arrays one
//     after the other chip-wide (split = 2).
#include "common.hpp"

namespace pm {

template <int POL>
__device__ __forceinline__ void probe_store(v4f *p, const v4f v) {
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

struct ProbeArgs {
    const v4f *src;
    v4f *dst, *dst2;     // dst2: the second output array (split != 0)
    int64_t nbc;         // block-chunks (a block-chunk = one chunk per wave of the workgroup, contiguous)
    int64_t nbc_pad;     // nbc rounded up to a multiple of 8 * granule
    int32_t rd4;         // dwordx4 per lane read in front of a chunk's stores
    int32_t placement;   // 0 linear, 1 one contiguous range per XCD, 2 XCD-interleaved runs of `granule` block-chunks
    int32_t granule;
    int32_t split;       // 0: one array; 1: 3/4 of the chunk to dst, 1/4 to dst2; 2: as 1, but all of dst first, then all of dst2
    int32_t data;        // what is stored: 0 a per-chunk pattern (+ what was read), 1 one constant (the reads still happen), 2 random bits
    int32_t plain_loads; // 0: nt loads, 1: plain loads
    int32_t serial_loads;  // 1: wait for every load before the next is issued
    int32_t lds;           // bytes of (unused) LDS per workgroup: bounds the resident workgroups per CU like a kernel's LDS tile does
};

// 128 pseudo-random bits per (store, lane): does WHAT is written change the rate (toggling data lines, power)?
__device__ __forceinline__ v4f probe_bits(const int64_t i) {
    unsigned x = (unsigned)i * 2654435761u + (unsigned)(i >> 32);
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return v4f{__uint_as_float(x), __uint_as_float(x * 3266489917u), __uint_as_float((x ^ 0x9e3779b9u) * 668265263u), __uint_as_float(~x * 374761393u)};
}

__device__ __forceinline__ int64_t probe_place(const ProbeArgs &a, const int64_t vb) {
    if (a.placement == 0) return vb;
    const int64_t xcd = vb % PM_NXCD, i = vb / PM_NXCD;
    if (a.placement == 1) return xcd * (a.nbc_pad / PM_NXCD) + i;
    const int64_t g = a.granule;
    return (i / g) * (PM_NXCD * g) + xcd * g + (i % g);
}

template <int POL, int BURST, bool BUF>
__global__ void store_probe_kernel(const ProbeArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    constexpr int BA = (BURST * 3) / 4, BB = BURST - BA;
    for (int pass = 0; pass < (a.split == 2 ? 2 : 1); ++pass) {
        for (int64_t vb = blockIdx.x; vb < a.nbc_pad; vb += gridDim.x) {
            const int64_t bc = probe_place(a, vb);
            if (bc >= a.nbc) continue;
            const int64_t c = bc * wpb + wv;
            v4f acc = v4f{(float)(int)c, 1.0f, 2.0f, 3.0f};
            if (pass == 0 && a.rd4 > 0) {
                if (a.serial_loads) {  // one load in flight per wave: a dependent chain of rd4 memory latencies
                    for (int u = 0; u < a.rd4; ++u) {
                        const v4f *q = a.src + (c * a.rd4 + u) * 64 + lane;
                        acc += a.plain_loads ? *q : __builtin_nontemporal_load(q);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                } else {               // every load of the chunk issued before the first use: one latency per chunk (what fk does)
                    v4f ld[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        ld[u] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                        if (u < a.rd4) {
                            const v4f *q = a.src + (c * a.rd4 + u) * 64 + lane;
                            ld[u] = a.plain_loads ? *q : __builtin_nontemporal_load(q);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += ld[u];
                }
            }
            if (a.data == 1) {
                asm volatile("" ::"v"(acc));  // the loads stay, their values go nowhere
                acc = v4f{1.0f, 0.0f, 0.0f, 0.0f};
            } else if (a.data == 2) {
                const v4f r = probe_bits(c * 64 + lane);
                asm volatile("" ::"v"(acc));
                acc = r;
            }
            if (a.split == 0) {
                v4f *p = a.dst + c * (BURST * 64) + lane;
                if constexpr (BUF) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.dst + c * (BURST * 64), 0, BURST * 1024, 0x00020000);
#pragma unroll
                    for (int u = 0; u < BURST; ++u)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, acc), rs,
                                                               (u * 64 + lane) * 16, 0, POL == 1 ? 2 : 0);
                } else {
#pragma unroll
                    for (int u = 0; u < BURST; ++u) probe_store<POL>(p + u * 64, acc);
                }
            } else {
                if (a.split == 1 || pass == 0) {
                    v4f *p = a.dst + c * (BA * 64) + lane;
#pragma unroll
                    for (int u = 0; u < BA; ++u) probe_store<POL>(p + u * 64, acc);
                }
                if (a.split == 1 || pass == 1) {
                    v4f *p = a.dst2 + c * (BB * 64) + lane;
#pragma unroll
                    for (int u = 0; u < BB; ++u) probe_store<POL>(p + u * 64, acc);
                }
            }
        }
    }
}

template <int POL, int BURST, bool BUF>
static int launch_probe(const ProbeArgs &a, const int grid, const int threads, hipStream_t s) {
    auto k = store_probe_kernel<POL, BURST, BUF>;
    if (int e = allow_lds(k, (size_t)a.lds)) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3((unsigned)threads), (size_t)a.lds, s, a);
    return PM_AFTER_LAUNCH("store_probe");
}

template <int BURST>
static int launch_probe_b(const ProbeArgs &a, const int pol, const int grid, const int threads, hipStream_t s) {
    switch (pol) {
        case 0: return launch_probe<0, BURST, false>(a, grid, threads, s);
        case 1: return launch_probe<1, BURST, false>(a, grid, threads, s);
        case 2: return launch_probe<2, BURST, false>(a, grid, threads, s);
        case 3: return launch_probe<3, BURST, false>(a, grid, threads, s);
        case 4: return launch_probe<4, BURST, false>(a, grid, threads, s);
        case 5: return launch_probe<5, BURST, false>(a, grid, threads, s);
        case 6: return launch_probe<6, BURST, false>(a, grid, threads, s);
        case 10: return launch_probe<0, BURST, true>(a, grid, threads, s);
        case 11: return launch_probe<1, BURST, true>(a, grid, threads, s);
    }
    set_error("store_probe: policy must be 0..6, 10 or 11");
    return PM_EINVAL;
}

// ---- the latency floor of a one-pass scan (tools/unroll_probe.py, profiles/r06_unroll_sweep.txt) -----------------------------------------
// quat.unroll on a clip of real length (2^14 ... 2^18 frames) is 10-45 us: a handful of DEPENDENT memory round trips, not a stream -- "% of 8 TB/s"
// is the wrong ruler there.  This kernel has the scan's grid and its chain of dependencies and none of its work: a workgroup takes a ticket, loads
// ONE dwordx4 per thread (its tile's first row), publishes a status word, waits for its predecessor's (the look-back's one read, at the depth the
// real scan pays when nothing is aggregated yet), and stores one dwordx4 per thread.  `epoch` makes the words self-resetting (a word counts as
// published when it holds this call's epoch), so the probe needs no reset launch either.
struct FloorArgs { const v4f *src; v4f *dst; unsigned *ticket; unsigned *status; int64_t stride4; unsigned ntiles, epoch; int mode; };
// mode 0: one ticket counter for the launch (what the scans did up to round 5); 1: no ticket, tile = blockIdx.x (only safe while every workgroup of the
// launch is resident); 2: one counter per XCD, tile = 8 * ticket + XCC_ID (see unroll.hip)
__global__ __launch_bounds__(256) void scan_floor_kernel(const FloorArgs a) {
    __shared__ unsigned s_tile;
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (a.mode == 1) s_tile = blockIdx.x;
        else if (a.mode == 2) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            const unsigned per = (a.ntiles + 7u - xcc) / 8u;  // tiles of this XCD: xcc, xcc + 8, ...
            s_tile = (__hip_atomic_fetch_add(a.ticket + 16u * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.epoch * per) * 8u + xcc;
        } else s_tile = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.epoch * a.ntiles;
    }
    __syncthreads();
    const unsigned tile = s_tile;
    if (tile >= a.ntiles) return;  // (mode 2: an XCD that started more workgroups than it has tiles -- see pm_scan_floor_probe)
    const v4f x = __builtin_nontemporal_load(a.src + (int64_t)tile * a.stride4 + tid);
    // (the status word is published after the tile's own load has returned, like the scan's map of its tile; relaxed agent-scope words like the scan's)
    const unsigned dep = __float_as_uint(x.x) & 0u;
    if (tid == 0) {
        __hip_atomic_store(a.status + tile, a.epoch + 1u + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile > 0)
            while (__hip_atomic_load(a.status + tile - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch + 1u) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_nontemporal_store(x, a.dst + (int64_t)tile * a.stride4 + tid);
}

}  // namespace pm

using namespace pm;

// ntiles workgroups of `threads` (64 ... 256) threads; tile t touches dwordx4 [t * stride4, t * stride4 + threads) of src / dst; ws: ntiles + 129 words,
// zero-filled ONCE; epoch: 0, 1, 2, ... call by call on the same ws with the same ntiles and mode.  mode: see scan_floor_kernel.
// (mode 2 counts on the dispatcher's round robin -- workgroup b on XCD b % 8 -- for every XCD to start exactly its share of the workgroups; an XCD that
// starts more drops the surplus and the tiles of the XCD that started fewer are never taken: a probe, not a scan -- unroll.hip's version hands the
// surplus on.)
extern "C" int pm_scan_floor_probe(const float *src, float *dst, void *ws, int64_t ntiles, int64_t stride4, int32_t threads, uint32_t epoch, int32_t mode,
                                   pm_stream_t stream) {
    PM_CHECK_ARGS(src && dst && ws && ntiles >= 1 && ntiles <= 0x7fffffffLL && stride4 >= threads && threads >= 64 && threads <= 256 && threads % 64 == 0 &&
                  aligned16(src) && aligned16(dst) && mode >= 0 && mode <= 2, "scan_floor_probe: bad arguments");
    FloorArgs a;
    a.src = reinterpret_cast<const v4f *>(src); a.dst = reinterpret_cast<v4f *>(dst);
    a.ticket = reinterpret_cast<unsigned *>(ws); a.status = a.ticket + 128;
    a.stride4 = stride4; a.ntiles = (unsigned)ntiles; a.epoch = epoch; a.mode = mode;
    hipLaunchKernelGGL(scan_floor_kernel, dim3((unsigned)ntiles), dim3((unsigned)threads), 0, static_cast<hipStream_t>(stream), a);
    return PM_AFTER_LAUNCH("scan_floor_probe launch");
}

// cfg (host ints): [0] burst KiB per wave and chunk (1, 2, 3, 4, 6, 8, 12, 16, 24, 32), [1] rd4, [2] store policy (0 plain, 1 nt, 2 sc1, 3 sc0 sc1,
// 4 sc0 sc1 nt, 5 sc0, 6 sc1 nt, 10 / 11 buffer_store plain / nt; -1: hipMemsetD32Async of the same bytes instead of a kernel),
// [3] placement, [4] granule (block-chunks), [5] threads per workgroup (64 or 256), [6] grid (0: one block-chunk per workgroup),
// [7] split, [8] data (0 per-chunk pattern + what was read, 1 one constant, 2 random bits), [9] 1 = plain instead of nt loads,
// [10] 1 = one load in flight per wave (each waited for) instead of all of a chunk's loads up front, [11] bytes of unused LDS per workgroup
// (bounds the workgroups resident per CU the way a kernel's LDS tile does).  n4 = dwordx4 to write (rounded down to whole block-chunks); src must hold n4 * rd4 / burst dwordx4 when rd4 > 0.
extern "C" int pm_store_probe_f32(const float *src, float *dst, int64_t n4, const int32_t *cfg, pm_stream_t stream) {
    PM_CHECK_ARGS(dst && cfg && n4 >= 0 && aligned16(dst) && (!src || aligned16(src)), "store_probe: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int burst = cfg[0], rd4 = cfg[1], pol = cfg[2], placement = cfg[3], granule = cfg[4] > 0 ? cfg[4] : 1, threads = cfg[5];
    const int split = cfg[7];
    if (pol == -1) return check_hip(hipMemsetD32Async((hipDeviceptr_t)dst, 0x3f800000, (size_t)n4 * 4, s), "hipMemsetD32Async");
    PM_CHECK_ARGS((threads == 64 || threads == 256) && rd4 >= 0 && (rd4 == 0 || src) && placement >= 0 && placement <= 2 && split >= 0 && split <= 2,
                  "store_probe: bad configuration");
    bool burst_ok = false;
    for (const int b : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}) burst_ok |= burst == b;
    PM_CHECK_ARGS(burst_ok, "store_probe: burst must be 1, 2, 3, 4, 6, 8, 12, 16, 24 or 32 KiB");   // before it divides anything
    PM_CHECK_ARGS(split == 0 || burst % 4 == 0, "store_probe: split needs a burst that is a multiple of 4");
    ProbeArgs a;
    const int wpb = threads / 64;
    a.src = reinterpret_cast<const v4f *>(src);
    a.dst = reinterpret_cast<v4f *>(dst);
    a.nbc = n4 / ((int64_t)burst * 64 * wpb);
    if (a.nbc == 0) return PM_OK;
    a.dst2 = a.dst + a.nbc * wpb * ((burst * 3) / 4) * 64;
    const int64_t unit = (int64_t)PM_NXCD * (placement == 2 ? granule : 1);
    a.nbc_pad = (a.nbc + unit - 1) / unit * unit;
    a.rd4 = rd4; a.placement = placement; a.granule = granule; a.split = split;
    a.data = cfg[8]; a.plain_loads = cfg[9]; a.serial_loads = cfg[10]; a.lds = cfg[11];
    PM_CHECK_ARGS(rd4 <= 16 && a.lds >= 0 && (size_t)a.lds <= kMaxLds, "store_probe: rd4 <= 16, 0 <= lds <= 160 KiB");
    const int64_t grid = cfg[6] > 0 ? cfg[6] : a.nbc_pad;
    if (grid > 0x7fffffffLL) { set_error("store_probe: grid too large"); return PM_EUNSUPPORTED; }
    switch (burst) {
        case 1: return launch_probe_b<1>(a, pol, (int)grid, threads, s);
        case 2: return launch_probe_b<2>(a, pol, (int)grid, threads, s);
        case 3: return launch_probe_b<3>(a, pol, (int)grid, threads, s);
        case 4: return launch_probe_b<4>(a, pol, (int)grid, threads, s);
        case 6: return launch_probe_b<6>(a, pol, (int)grid, threads, s);
        case 12: return launch_probe_b<12>(a, pol, (int)grid, threads, s);
        case 24: return launch_probe_b<24>(a, pol, (int)grid, threads, s);
        case 8: return launch_probe_b<8>(a, pol, (int)grid, threads, s);
        case 16: return launch_probe_b<16>(a, pol, (int)grid, threads, s);
        case 32: return launch_probe_b<32>(a, pol, (int)grid, threads, s);
    }
    return PM_EINVAL;  // (unreachable: burst_ok above)
}
