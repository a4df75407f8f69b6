#!/usr/bin/env python3
"""Is the ~4.5 TB/s pure-write rate the chip or the pattern?  (round-3 review, item 1)

Writes fk's output size (2^20 x 22 x 48 B = 1.107 GB) under every knob of the store path -- per-wave burst, cache policy,
XCD -> address placement, workgroup size / persistence, one or two output arrays, with and without fk's reads in front --
through `pm_store_probe_f32` (csrc/probe.hip), and times hipMemsetD32Async and torch's fill as outside references.
Sustained timing (back-to-back launches between two HIP events after a warm-up under the same kernel), like bench.py.
    python tools/store_probe.py > profiles/r04_store_patterns.txt
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

PEAK = 8000.0
POL = {0: "plain", 1: "nt", 2: "sc1", 3: "sc0 sc1", 4: "sc0 sc1 nt", 5: "sc0", 6: "sc1 nt", 10: "buffer plain", 11: "buffer nt", -1: "hipMemsetD32Async"}
PLACE = {0: "linear", 1: "xcd-contiguous", 2: "xcd-interleaved"}


def p(t):
    return C.c_void_p(t.data_ptr())


def sustained(fn, n=60, warm=120):
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(3):
        _lib.call("pm_event_record", ev[0], None)
        for _ in range(n):
            fn()
        _lib.call("pm_event_record", ev[1], None)
        ms = C.c_float()
        _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        best = min(best, ms.value / n)
    for e in ev:
        _lib.call("pm_event_destroy", e)
    return best


def main():
    dev = torch.device("cuda:0")
    F, J = 1 << 20, 22
    n4 = F * J * 3                                     # dwordx4 of fk's outputs (48 J B per frame)
    dst = torch.empty(n4 * 4 + (1 << 22), device=dev)  # + slack for the rounding of block-chunks
    src = torch.randn(n4 * 2 + (1 << 22), device=dev)         # up to 1 : 2 read : write
    warm = torch.empty(1 << 26, device=dev)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(20):
            warm.add_(1.0)
        torch.cuda.synchronize()
    del warm

    def run(burst, rd4=0, pol=1, place=1, gran=1, threads=64, grid=0, split=0, data=0, plain_loads=0, serial=0, lds=0):
        cfg = (C.c_int32 * 12)(burst, rd4, pol, place, gran, threads, grid, split, data, plain_loads, serial, lds)
        ms = sustained(lambda: _lib.call("pm_store_probe_f32", p(src) if rd4 else None, p(dst), n4, cfg, None))
        wpb = threads // 64
        nbc = n4 // (burst * 64 * wpb)
        wbytes = nbc * wpb * burst * 1024 if pol != -1 else n4 * 16
        rbytes = nbc * wpb * rd4 * 1024
        return ms, wbytes, rbytes

    def line(label, ms, wbytes, rbytes=0):
        tot = (wbytes + rbytes) / (ms * 1e-3) / 1e9
        w = wbytes / (ms * 1e-3) / 1e9
        print(f"{label:78s} {ms * 1e3:8.1f} us  total {tot:7.1f} GB/s {tot / PEAK * 100:5.1f} %   writes {w:7.1f} GB/s {w / PEAK * 100:5.1f} %", flush=True)

    print(f"# store-pattern probe, {n4 * 16 / 1e9:.3f} GB written per launch (fk's outputs at 2^20 x 22); % of the 8 TB/s HBM spec")
    print("# (a) outside references")
    ms, wb, _ = run(4, pol=-1)
    line("hipMemsetD32Async", ms, wb)
    flat = dst[: n4 * 4]
    ms = sustained(lambda: flat.fill_(1.0))
    line("torch.Tensor.fill_", ms, n4 * 16)
    ms = sustained(lambda: _lib.call("pm_stream_plain_f32", p(src), p(dst), n4, -1, 8192, None))
    line("round-3 pure-write stream (pm_stream_plain_f32, 8192 x 256 threads, grid-stride, nt)", ms, n4 * 16)

    print("# (b) per-wave contiguous burst (nt stores, one contiguous range per XCD, one chunk per wave)")
    for threads in (64, 256):
        for burst in (1, 2, 4, 8, 16, 32):
            ms, wb, _ = run(burst, threads=threads)
            line(f"burst {burst:2d} KiB, {threads:3d}-thread workgroups", ms, wb)
    print("# (b') the same, persistent grid-stride workgroups (256 threads)")
    for grid in (1024, 2048, 4096, 8192):
        for burst in (1, 4, 16):
            ms, wb, _ = run(burst, threads=256, grid=grid, place=0)
            line(f"burst {burst:2d} KiB, grid {grid:5d} x 256 threads, linear placement", ms, wb)

    print("# (c) store cache policy (one chunk per wave, 64-thread workgroups, one contiguous range per XCD)")
    for burst in (4, 16):
        for pol in (0, 1, 2, 3, 4, 5, 6, 10, 11):
            ms, wb, _ = run(burst, pol=pol)
            line(f"burst {burst:2d} KiB, policy {POL[pol]}", ms, wb)

    print("# (d) chunk -> address placement (burst 4 KiB, 64-thread workgroups = 4 KiB block-chunks, nt; workgroup b runs on XCD b % 8)")
    for pol in (1, 0):
        ms, wb, _ = run(4, pol=pol, place=0)
        line(f"linear: neighbouring 4 KiB chunks on different XCDs, policy {POL[pol]}", ms, wb)
        for gran in (4, 16, 64, 256, 4096):
            ms, wb, _ = run(4, pol=pol, place=2, gran=gran)
            line(f"XCD-interleaved runs of {gran * 4:6d} KiB, policy {POL[pol]}", ms, wb)
        ms, wb, _ = run(4, pol=pol, place=1)
        line(f"one contiguous eighth per XCD (xcd_tile, what the library does), policy {POL[pol]}", ms, wb)
    print("# (d') the same with 16 KiB chunks (fk's tile is 16.5 KB)")
    ms, wb, _ = run(16, place=0)
    line("linear, 16 KiB chunks", ms, wb)
    for gran in (4, 64, 1024):
        ms, wb, _ = run(16, place=2, gran=gran)
        line(f"XCD-interleaved runs of {gran * 16:6d} KiB", ms, wb)
    ms, wb, _ = run(16, place=1)
    line("one contiguous eighth per XCD", ms, wb)

    print("# (e) fk's mix: reads (nt loads, ALL of a chunk's loads issued before the first use) in front of the stores; one chunk per wave")
    for burst, rd4 in ((4, 1), (8, 2), (8, 3), (16, 5), (32, 10)):
        ms, wb, rb = run(burst, rd4=rd4)
        line(f"{rd4} KiB read + {burst} KiB written per wave, one array, nt", ms, wb, rb)
    for pol in (1, 0, 2):
        for split, name in ((0, "one output array"), (1, "two arrays, 12 + 4 KiB back to back (rotmats + pos)"), (2, "two arrays, one after the other chip-wide")):
            ms, wb, rb = run(16, rd4=5, pol=pol, split=split)
            line(f"5 + 16 KiB per wave, {name}, policy {POL[pol]}", ms, wb, rb)
    for threads, grid in ((256, 0), (256, 4096)):
        ms, wb, rb = run(16, rd4=5, threads=threads, grid=grid)
        line(f"5 + 16 KiB per wave, {threads}-thread workgroups, grid {grid or 'one chunk per wave'}", ms, wb, rb)
    print("# (e') bytes in flight: the same 5 + 16 KiB per wave with the waves per CU bounded by an (unused) LDS allocation per wave,")
    print("#      as a kernel's LDS tile bounds them (fk at J = 22: 16.9 KB per wave = 9 waves per CU), and with one load in flight per wave")
    for lds_kb in (0, 5, 8, 12, 17, 22, 33, 66):
        ms, wb, rb = run(16, rd4=5, lds=lds_kb * 1024)
        wpc = min(32, (160 * 1024) // (lds_kb * 1024)) if lds_kb else 32
        line(f"5 + 16 KiB per wave, {lds_kb:2d} KB of LDS per wave (<= {wpc:2d} waves per CU)", ms, wb, rb)
    for lds_kb in (0, 17):
        ms, wb, rb = run(16, rd4=5, lds=lds_kb * 1024, serial=1)
        line(f"5 + 16 KiB per wave, {lds_kb:2d} KB of LDS per wave, ONE load in flight per wave", ms, wb, rb)
    for lds_kb in (0, 4, 8, 17):
        ms, wb, rb = run(4, rd4=1, lds=lds_kb * 1024)
        line(f"1 + 4 KiB per wave, {lds_kb:2d} KB of LDS per wave", ms, wb, rb)
    for lds_kb in (0, 8, 17, 33):
        ms, wb, _ = run(16, lds=lds_kb * 1024)
        line(f"pure write, 16 KiB per wave, {lds_kb:2d} KB of LDS per wave", ms, wb)

    print("# (h) the mix by chunk size, read : write ratio, waves per CU (LDS bound) and placement -- nt loads up front, nt stores")
    for ratio, pairs in (("1 : 4", ((4, 1), (8, 2), (16, 4), (32, 8))), ("1 : 3", ((3, 1), (6, 2), (12, 4), (24, 8))), ("1 : 2", ((2, 1), (4, 2), (8, 4), (16, 8))),
                         ("other", ((12, 3), (16, 5), (16, 6), (8, 3), (24, 7)))):
        for burst, rd4 in pairs:
            row = []
            for lds_kb in (0, 8, 17, 33):
                ms, wb, rb = run(burst, rd4=rd4, lds=lds_kb * 1024)
                row.append(f"{(wb + rb) / ms / 1e6 / 80:5.1f}")
            print(f"ratio {ratio:6s} {rd4:2d} KiB read + {burst:2d} KiB written per wave:  % of spec with 0 / 8 / 17 / 33 KB of LDS per wave (<= 32 / 20 / 9 / 4 waves per CU): " + " / ".join(row), flush=True)
    for burst, rd4 in ((16, 5), (12, 4), (4, 1)):
        for lds_kb in (0, 17):
            row = []
            for place, gran in ((0, 1), (2, 4), (2, 64), (2, 1024), (1, 1)):
                ms, wb, rb = run(burst, rd4=rd4, lds=lds_kb * 1024, place=place, gran=gran)
                row.append(f"{(wb + rb) / ms / 1e6 / 80:5.1f}")
            print(f"{rd4} + {burst} KiB per wave, {lds_kb:2d} KB LDS: placement linear / XCD-interleaved runs of 4 / 64 / 1024 chunks / one range per XCD: " + " / ".join(row), flush=True)

    print("# (g) does WHAT is written matter?  (one chunk per wave, 64-thread workgroups, one contiguous range per XCD)")
    DATA = {0: "per-chunk pattern (c, 1, 2, 3) [+ what was read]", 1: "one constant (1, 0, 0, 0)", 2: "random bits"}
    for burst in (4, 16):
        for pol in (1, 0):
            for data in (0, 1, 2):
                ms, wb, _ = run(burst, pol=pol, data=data)
                line(f"pure write, burst {burst:2d} KiB, policy {POL[pol]}, data: {DATA[data]}", ms, wb)
    for burst, rd4 in ((4, 1), (16, 5)):
        for data in (0, 1, 2):
            for plain in (0, 1):
                ms, wb, rb = run(burst, rd4=rd4, data=data, plain_loads=plain)
                line(f"{rd4} KiB read ({'plain' if plain else 'nt'} loads of N(0,1) floats) + {burst} KiB written, nt stores, data: {DATA[data]}", ms, wb, rb)
    zsrc = torch.zeros_like(src)
    keep = src
    src = zsrc
    for burst, rd4 in ((4, 1), (16, 5)):
        for data in (0, 1):
            ms, wb, rb = run(burst, rd4=rd4, data=data)
            line(f"{rd4} KiB read (nt loads of ZEROS) + {burst} KiB written, nt stores, data: {DATA[data]}", ms, wb, rb)
    src = keep
    del zsrc

    print("# (f) the kernels themselves, same session")
    rot, root, off, parents = syn.fk_workload(F, seed=0)
    rot_d, root_d, off_d = (torch.from_numpy(x).to(dev) for x in (rot, root, off))
    pos = torch.empty((F, J, 3), device=dev)
    rm = torch.empty((F, J, 3, 3), device=dev)
    pp = parents.astype(np.int32).ctypes.data_as(C.c_void_p)
    ms = sustained(lambda: _lib.call("pm_fk_f32", p(rot_d), p(root_d), p(off_d), 0, pp, F, J, p(pos), p(rm), None))
    line("pm_fk_f32 2^20 x 22", ms, F * 48 * J, F * (16 * J + 12))
    ms = sustained(lambda: _lib.call("pm_stream_ceiling_f32", p(rot_d), p(dst), F, 4 * J, 12 * J, None))
    line("pm_stream_ceiling_f32 (fk's tiling, no arithmetic)", ms, F * 48 * J, F * 16 * J)
    rot_d.zero_(); rot_d[..., 0] = 1.0; root_d.zero_()
    ms = sustained(lambda: _lib.call("pm_fk_f32", p(rot_d), p(root_d), p(off_d), 0, pp, F, J, p(pos), p(rm), None))
    line("pm_fk_f32 2^20 x 22 on IDENTITY rotations, zero roots (outputs: 0 / 1 and the offsets)", ms, F * 48 * J, F * (16 * J + 12))
    N = F * J
    t3 = torch.randn((N, 3), device=dev)
    d8 = torch.empty((N, 8), device=dev)
    ms = sustained(lambda: _lib.call("pm_dq_from_t_f32", p(t3), N, p(d8), None))
    line("dq.from_translation (12 B in, 32 B out per record: 5 of 8 output floats are constants)", ms, N * 32, N * 12)
    t3.zero_()
    ms = sustained(lambda: _lib.call("pm_dq_from_t_f32", p(t3), N, p(d8), None))
    line("dq.from_translation of zeros", ms, N * 32, N * 12)
    q4 = torch.randn((N, 4), device=dev)
    m9 = torch.empty((N, 9), device=dev)
    ms = sustained(lambda: _lib.call("pm_quat_to_matrix_f32", p(q4), N, p(m9), None))
    line("quat.to_matrix of N(0,1) quaternions (16 B in, 36 B out)", ms, N * 36, N * 16)
    q4.zero_(); q4[:, 0] = 1.0
    ms = sustained(lambda: _lib.call("pm_quat_to_matrix_f32", p(q4), N, p(m9), None))
    line("quat.to_matrix of identity quaternions", ms, N * 36, N * 16)


if __name__ == "__main__":
    main()
