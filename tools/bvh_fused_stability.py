#!/usr/bin/env python3
"""run-to-run stability of pm_bvh_rotations_f32 at 2^20 x 22 per tile size / reservation (fresh buffers and data every repeat)"""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S, T = 22, 1 << 20
order_h = np.tile(np.array([2, 0, 1], np.uint8), (S, 1))
for R in (8, 16, 32):
    for res in (1, 0):
        os.environ["PM_UNROLL_R"] = str(R); os.environ["PM_UNROLL_RESERVE"] = str(res)
        row = []
        for rep in range(5):
            junk = torch.empty(int(np.random.default_rng(rep).integers(1, 1 << 22)), device="cuda")  # move the allocations around
            deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
            out = torch.empty((T, S, 4), device="cuda")
            ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
            ms, _ = pp.timeit(lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None))
            row.append(f"{ms * 1e3:6.1f}")
            del junk, deg, out, ws
        print(f"R={R:2d} reserve={res}: " + " ".join(row), flush=True)
