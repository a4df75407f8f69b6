#!/usr/bin/env python3
"""fk over a sweep of joint counts (chains with two branch points, 2^19 frames): where the kernel shapes switch
(23|24 joints, 64|65) and what multiples of 8 joints -- frame strides that alias in LDS -- cost.  Tuning aid."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib

pp.SUSTAINED = 60
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F = 1 << 19
for J in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,16,22,23,24,28,32,40,48,52,56,64,65,72".split(","))]:
    par = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    par[J // 2] = 0
    par[3 * J // 4] = J // 4
    rot = torch.randn((F, J, 4), device="cuda")
    root = torch.randn((F, 3), device="cuda")
    off = torch.randn((J, 3), device="cuda")
    pos = torch.empty((F, J, 3), device="cuda")
    rm = torch.empty((F, J, 3, 3), device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None))
    gb = F * (64 * J + 12) / ms / 1e6
    print(f"FPW={os.environ.get('PM_FK_FPW', 'auto'):>4} J={J:3d}: {ms * 1e3:7.1f} us  {gb:6.0f} GB/s ({gb / 80:.1f}% of 8 TB/s)", flush=True)
    del rot, pos, rm
