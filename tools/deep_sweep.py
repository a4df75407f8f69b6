#!/usr/bin/env python3
"""to_root_dual_quat and fk over joint counts on the joint sweep's chain-like skeleton (2^19 frames): the lane-per-frame kernels of
deep.hip against the tile kernels (PMHIP_VARIANT=tuning PM_DQ_DEEP=0/1 forces either; fk rides along for the tile kernels' numbers).   tools/deep_sweep.py 64,65,128"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib

pp.SUSTAINED = int(os.environ.get("SUSTAINED", "40"))
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F = 1 << 19
which = os.environ.get("DEEP_SWEEP", "dq,fk").split(",")
for J in [int(x) for x in sys.argv[1].split(",")]:
    par = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    par[J // 2] = 0
    par[3 * J // 4] = J // 4
    rot = torch.randn((F, J, 4), device="cuda")
    root = torch.randn((F, 3), device="cuda")
    off = torch.randn((J, 3), device="cuda") * 0.15
    off[0] = 0
    pp_ = par.ctypes.data_as(C.c_void_p)
    line = f"J={J:3d}:"
    if "dq" in which:
        dq = torch.empty((F, J, 8), device="cuda")
        ms, _ = pp.timeit(lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None))
        line += f" to_root {ms * 1e3:7.1f} us {F * (48 * J + 12) / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().split('(')[0].split('::')[-1]:28s}"
        del dq
    if "fk" in which:
        pos = torch.empty((F, J, 3), device="cuda")
        rm = torch.empty((F, J, 3, 3), device="cuda")
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        line += f" | fk {ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().split('(')[0].split('::')[-1]}"
        del pos, rm
    if "mirror" in which:
        mi = torch.empty((F, J, 4), device="cuda")
        ms, _ = pp.timeit(lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, None, 0, F, J, P(mi), None))
        line += f" | mirror(all) {ms * 1e3:7.1f} us {F * 32 * J / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().split('(')[0].split('::')[-1]}"
        del mi
    print(line, flush=True)
    del rot
