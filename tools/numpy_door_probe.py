#!/usr/bin/env python3
"""End-to-end time of the NumPy front door (host arrays in, host arrays out: H2D + kernel + D2H + float64
cast), next to the device-resident kernel time.  For the PCIe-inclusive note in DESIGN.md §6."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pymotion_amd.ops.skeleton as sk
from pymotion_amd import synthetic as syn

for F in (1000, 1 << 17, 1 << 20):
    rot, root, off, par = syn.fk_workload(F, seed=1)
    sk.fk(rot[:64], root[:64], off, par)
    ts, keep = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        keep.append(sk.fk(rot, root, off, par))  # results stay alive: handing 2.2 GB of float64 back to the OS is the
        ts.append(time.perf_counter() - t0)      # caller's cost (reported separately), not the call's
    t = min(ts)
    med = sorted(ts)[len(ts) // 2]
    t0 = time.perf_counter()
    del keep
    tf = [(time.perf_counter() - t0) / 5]
    moved = F * (64 * 22 + 12)
    print(f"NumPy door fk F={F}: {t * 1e3:.2f} ms (median of 5: {med * 1e3:.2f})  {F / t:.3e} frames/s  ({moved / t / 1e9:.1f} GB/s of fp32 payload over PCIe, output cast to float64; freeing the result: {min(tf) * 1e3:.1f} ms)")
