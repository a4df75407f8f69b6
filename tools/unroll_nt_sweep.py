#!/usr/bin/env python3
"""quat.unroll one pass over workgroup size (PM_UNROLL_NT), records per thread (PM_UNROLL_R) and the long-chain LDS reservation, tuning build."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = 22
for lg in (14, 16, 18, 20):
    T = 1 << lg
    q = torch.randn((T, S, 4), device="cuda")
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) * 4 + 64, dtype=torch.uint8, device="cuda")
    row = []
    for NT in (256, 128):
        for R in (8, 16, 32):
            for res in (1, 0):
                os.environ.update({"PM_UNROLL_NT": str(NT), "PM_UNROLL_R": str(R), "PM_UNROLL_RESERVE": str(res)})
                ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_f32", P(q), T, S, P(out), P(ws), None))
                row.append(f"NT{NT} R{R}{'r' if res else ' '} {ms * 1e3:6.1f}")
    print(f"T=2^{lg}: " + " | ".join(row), flush=True)
