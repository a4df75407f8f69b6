#!/usr/bin/env python3
"""how much of a config-4 sized fk launch is its last, partly filled round of workgroups?  SMPL-H, two tiles of four frames a workgroup, 3584 workgroups
resident (256 CUs x 14): frames = rounds x 3584 x 8.  Time per frame at whole and fractional round counts, one process."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
J = 52
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
Fmax = 3584 * 8 * 40
rot = torch.randn((Fmax, J, 4), device="cuda"); root = torch.rand((Fmax, 3), device="cuda") * 4 - 2
off = torch.randn((J, 3), device="cuda") * 0.1
pos = torch.empty((Fmax, J, 3), device="cuda"); rm = torch.empty((Fmax, J, 3, 3), device="cuda")
for rep in range(2):
    for rounds in (8.0, 8.5, 9.0, 9.14, 9.5, 10.0, 18.0, 18.3, 36.0, 36.5):
        F = int(round(rounds * 3584 * 8)) if rounds != 9.14 else 1 << 18
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None))
        print(f"rounds {F / (3584 * 8):6.2f}  F = {F:8d}: {ms * 1e3:7.1f} us  {ms * 1e6 / F:6.3f} ns a frame  {F * (64 * J + 12) / ms / 1e6 / 80:5.1f} %", flush=True)
