#!/usr/bin/env python3
"""from_root_positions over joint counts, sustained timing (28 J bytes per frame against the 8 TB/s HBM spec)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

pp.SUSTAINED = 60
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PM_IK"))
for J in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "22,52,128".split(","))]:
    par = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if par is None:
        par = syn.random_parents(J, np.random.default_rng(J))
    F = (1 << 20) if J <= 24 else (1 << 18)
    pos = torch.randn((F, J, 3), device="cuda")
    off = torch.randn((J, 3), device="cuda")
    out = torch.empty((F, J, 4), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(out), None))
    print(f"[{tag}] J={J:3d} F=2^{F.bit_length() - 1}: from_root_positions {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}%", flush=True)
