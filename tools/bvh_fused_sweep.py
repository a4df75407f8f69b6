#!/usr/bin/env python3
"""pm_bvh_rotations_f32 over tile sizes (PM_UNROLL_R) and the long-chain LDS reservation (PM_UNROLL_RESERVE), tuning build."""
import ctypes as C
import os
import sys

os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib

pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = 22
order_h = np.tile(np.array([2, 0, 1], np.uint8), (S, 1))
for lg in (10, 12, 14, 16, 18, 20):
    T = 1 << lg
    deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
    out = torch.empty((T, S, 4), device="cuda")
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
    row = []
    for R in (4, 8, 16, 32):
        for res in (1, 0):
            os.environ["PM_UNROLL_R"] = str(R)
            os.environ["PM_UNROLL_RESERVE"] = str(res)
            ms, _ = pp.timeit(lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None))
            row.append(f"R{R}{'r' if res else ' '} {ms * 1e3:7.1f}")
    print(f"T=2^{lg}: " + " | ".join(row), flush=True)
