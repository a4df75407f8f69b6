#!/usr/bin/env python3
"""is the 'fast mode' of the J = 52 kernels a clock boost after an idle gap?  the same launch timed in consecutive windows of 100 launches (no warm-up in
front of a window), with idle gaps of 0 / 0.5 / 3 s between groups of windows (SMPL-H fk at 2^18 frames; the 22-joint body at 2^20)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for J, F, par in ((52, 1 << 18, syn.PARENTS_52), (22, 1 << 20, syn.PARENTS_22)):
    par = np.ascontiguousarray(par, dtype=np.int32)
    src = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    call = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def window(n=100):
        e0.record()
        for _ in range(n): call()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for gap in (0.0, 3.0, 0.5, 3.0, 0.0):
        torch.cuda.synchronize(); time.sleep(gap)
        ws = [window() for _ in range(12)]
        print(f"J={J} after {gap:3.1f} s idle: " + " ".join(f"{w:6.1f}" for w in ws) + " us", flush=True)
