#!/usr/bin/env python3
"""Per-call cost of the two Python front doors on small batches (config 1: one 1000-frame clip), i.e. the
part that is Python / ctypes / allocation rather than kernel time.  Tuning aid, not part of the bench."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import pymotion_amd.ops.skeleton as sk  # noqa: E402
import pymotion_amd.ops.skeleton_torch as skt
import pymotion_amd.rotations.quat_torch as qt
from pymotion_amd import synthetic as syn


_spin = torch.empty(1 << 26, device="cuda")


def bench(fn, n=200, warm=20):
    t_end = time.perf_counter() + 0.1   # 100 ms of device work first: clocks up (see tools/dvfs_idle_probe.py)
    while time.perf_counter() < t_end:
        _spin.add_(1.0)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per = []
    for _ in range(5):  # median of 5 batches: a process sees the odd one-off 10-40 ms host stall
        t0 = time.perf_counter()
        for _ in range(n // 5):
            fn()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / (n // 5) * 1e6)
    return sorted(per)[2]


for F in (1000, 100_000):
    rot, root, off, par = syn.fk_workload(F, normalized=True)
    tr, tg, to = (torch.from_numpy(a).cuda() for a in (rot, root, off))
    tp = torch.from_numpy(par)
    print(f"F={F}")
    print(f"  torch door fk                 {bench(lambda: skt.fk(tr, tg, to, tp)):9.1f} us/call")
    print(f"  torch door to_root_dual_quat  {bench(lambda: skt.to_root_dual_quat(tr, tg, tp, to)):9.1f} us/call")
    print(f"  torch door quat.to_matrix     {bench(lambda: qt.to_matrix(tr)):9.1f} us/call")
    print(f"  numpy door fk (f32 in, f64 out){bench(lambda: sk.fk(rot, root, off, par), n=20, warm=3):9.1f} us/call")
