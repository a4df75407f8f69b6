#!/usr/bin/env python3
"""quat.unroll over clip lengths (S = 22 series unless given): the one-pass look-back kernel against the three-pass scan
(PMHIP_VARIANT=tuning PM_UNROLL_ONEPASS=0) and its tile sizes (PM_UNROLL_R, PM_UNROLL_NT).  Tuning aid."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib

pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = int(sys.argv[1]) if len(sys.argv) > 1 else 22
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PM_UNROLL"))
for lg in (10, 12, 14, 16, 18, 20):
    T = 1 << lg
    q = torch.randn((T, S, 4), device="cuda")
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_f32", P(q), T, S, P(out), P(ws), None))
    # the same scan on a caller-owned pair of zeroed workspaces, alternated call by call: no reset launch (pm_unroll_onepass_f32)
    pair = torch.zeros((2, 1 << 17), dtype=torch.int64, device="cuda")
    st = {"cur": 0, "dirty": [0, 0]}

    def no_reset(kind=0, x=q, o=out, order=None):
        c = st["cur"]
        n = C.c_int64(0)
        _lib.call("pm_unroll_onepass_f32", kind, P(x), order, 1, T, S, P(o), P(pair[c]), C.byref(n), P(pair[1 - c]), st["dirty"][1 - c], None)
        st["dirty"][1 - c], st["dirty"][c], st["cur"] = 0, n.value, 1 - c

    fits = (T * S * 2 + 1023) // 1024 * 10 < (1 << 17)
    ms_p = pp.timeit(no_reset)[0] if fits else float("nan")
    # the latency floor of this launch (pm_scan_floor_probe): the scan's grid -- the tile size dispatch picks -- and its chain of dependencies
    # (ticket, one load, publish, wait for the predecessor's word, one store), none of its work
    nv = T * S
    R = 4 if nv <= 1024 else (8 if nv <= 2048 else (4 if (nv > 4096 and (nv + 4095) // 4096 < 64) else (32 if nv >= (4 << 20) else (8 if nv < (1 << 19) else 16))))
    ntl = (nv + 256 * R - 1) // (256 * R)
    floors = []
    for mode in (0, 1, 2):
        fws = torch.zeros(ntl + 129, dtype=torch.int32, device="cuda")
        ep = {"n": 0}

        def floor(mode=mode, fws=fws, ep=ep):
            _lib.call("pm_scan_floor_probe", P(q), P(out), P(fws), ntl, 256 * R, 256, ep["n"], mode, None)
            ep["n"] += 1

        floors.append(pp.timeit(floor)[0] if (mode != 1 or ntl <= 1024) else float("nan"))  # (no ticket: only while every workgroup is resident)
    ms_f = floors[0]
    nores = (f"{ms_p * 1e3:8.1f} us  {T * S * 32 / ms_p / 1e6 / 80:5.1f}%  = {ms_p / ms_f:4.2f} x floor" if fits else
             "   (the pair of this probe, 1 MB each, is too small for this clip: the doors fall back to the plain entry point too)")
    print(f"[{tag}] T=2^{lg} S={S}: {ms * 1e3:8.1f} us  {T * S * 32 / ms / 1e6 / 80:5.1f}% of 8 TB/s on 32 B/quaternion;  latency floor of the launch "
          f"({ntl} tiles: ticket, load, publish, look back once, store) {ms_f * 1e3:6.1f} us (no ticket {floors[1] * 1e3:5.1f}, a ticket counter per XCD {floors[2] * 1e3:5.1f});  without the reset launch (workspace pair) {nores}", flush=True)
# batches of clips [B, T, S, 4] along T: one launch, nothing transposed (raw ABI), and the torch door end to end
import pymotion_amd.rotations.quat_torch as quat_t

for B, T in ((16384, 64), (4096, 256), (64, 16384)):
    q = torch.randn((B, T, S, 4), device="cuda")
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_batched_workspace_bytes(B, T, S)) + 16, dtype=torch.uint8, device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_batched_f32", P(q), B, T, S, P(out), P(ws), None))
    ms2, _ = pp.timeit(lambda: quat_t.unroll(q, 1))
    print(f"[{tag}] batch B={B} T={T} S={S}: {ms * 1e3:8.1f} us  {B * T * S * 32 / ms / 1e6 / 80:5.1f}% of 8 TB/s on 32 B/quaternion;"
          f" quat_torch.unroll(q, 1) end to end {ms2 * 1e3:8.1f} us", flush=True)
# BVH ingest (io/bvh.py:352-359): the fused kernel (12 B in + 16 B out per joint and frame) against the three launches it replaces
# (from_euler 12 -> 16, unroll 16 -> 16, normalize 16 -> 16: 92 B), device-resident, raw ABI
import numpy as np

order_h = np.tile(np.array([2, 0, 1], np.uint8), (S, 1))
order_d = torch.from_numpy(order_h).cuda()
for lg in (10, 12, 14, 16, 18, 20):
    T = 1 << lg
    deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
    rad = torch.deg2rad(deg)
    q1, q2, out = (torch.empty((T, S, 4), device="cuda") for _ in range(3))
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")

    def three():
        _lib.call("pm_quat_from_euler_f32", P(rad), P(order_d), S, T * S, P(q1), None)
        _lib.call("pm_quat_unroll_f32", P(q1), T, S, P(q2), P(ws), None)
        _lib.call("pm_quat_normalize_f32", P(q2), T * S, C.c_float(1e-8), P(out), None)

    fused = lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None)  # noqa: E731
    pair = torch.zeros((2, 1 << 17), dtype=torch.int64, device="cuda")
    st = {"cur": 0, "dirty": [0, 0]}

    def fused_pair():
        c = st["cur"]
        n = C.c_int64(0)
        _lib.call("pm_unroll_onepass_f32", 2, P(deg), order_h.ctypes.data_as(C.c_void_p), 1, T, S, P(out), P(pair[c]), C.byref(n), P(pair[1 - c]), st["dirty"][1 - c], None)
        st["dirty"][1 - c], st["dirty"][c], st["cur"] = 0, n.value, 1 - c

    ms1p, _ = pp.timeit(fused_pair)
    ms1, _ = pp.timeit(fused)
    ms3, _ = pp.timeit(three)
    ms1b, _ = pp.timeit(fused)  # (again, after the three launches: the order must not matter)
    # (a timing window is only the kernel while the host stays ahead of the device; pp.timeit repeats a window in which the launching thread was
    # descheduled -- round 4: 15-35 ms stalls a few times per process made a 10 us kernel read 900 us in one window and 10 us in the next.  The
    # before / after pair is kept as a second guard.)
    note = ""
    if max(ms1, ms1b) > 2.0 * min(ms1, ms1b):
        ms1c, _ = pp.timeit(fused)
        note = f"  [one timing window of three was slow: {ms1 * 1e3:.1f} / {ms1b * 1e3:.1f} / {ms1c * 1e3:.1f} us]"
        ms1, ms1b = sorted((ms1, ms1b, ms1c))[:2]
    ms1, ms1b = min(ms1, ms1b), max(ms1, ms1b)
    print(f"[{tag}] get_data fused T=2^{lg} S={S}: {ms1 * 1e3:8.1f} us ({T * S * 28 / ms1 / 1e6 / 80:5.1f}% of 8 TB/s on 28 B per joint-frame)"
          f"  three launches {ms3 * 1e3:8.1f} us  -> {ms3 / ms1:4.2f}x   (fused before / after: {ms1 * 1e3:.1f} / {ms1b * 1e3:.1f}){note};"
          f"  without the reset launch {ms1p * 1e3:8.1f} us -> {ms3 / ms1p:4.2f}x", flush=True)
