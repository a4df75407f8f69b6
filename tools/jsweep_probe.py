#!/usr/bin/env python3
"""fk over a sweep of joint counts (chains with two branch points, 2^19 frames): where the kernel shapes switch
(23|24 joints, 64|65) and what multiples of 8 joints -- frame strides that alias in LDS -- cost.  Tuning aid.
PM_SWEEP_TOPO=bushy: random trees (parents[i] uniform in [0, i): depth ~ 2 ln J) instead of the chain-like skeleton."""
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
    if os.environ.get("PM_SWEEP_TOPO") == "bushy":
        from pymotion_amd import synthetic as syn
        par = syn.random_parents(J, np.random.default_rng(J))
    rot = torch.randn((F, J, 4), device="cuda")
    root = torch.randn((F, 3), device="cuda")
    off = torch.randn((J, 3), device="cuda") * 0.15  # human-scale bones in metres: the fp32 walk (bones >= 1 m take the fixed-point one, see fk.hip PREC_DYN)
    pos = torch.empty((F, J, 3), device="cuda")
    rm = torch.empty((F, J, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
    gb = F * (64 * J + 12) / ms / 1e6
    line = f"{os.environ.get('PM_SWEEP_TOPO', 'chain-like'):>10} J={J:3d}: fk {ms * 1e3:7.1f} us {gb / 80:5.1f}%"
    if os.environ.get("PM_SWEEP_ALL"):
        off[0] = 0
        dq = torch.empty((F, J, 8), device="cuda")
        tr = torch.empty((F, J, 3), device="cuda")
        qo = torch.empty((F, J, 4), device="cuda")
        for name, fn, nb in (
            ("to_root", lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None), 48 * J + 12),
            ("from_root", lambda: _lib.call("pm_from_root_dq_f32", P(dq), pp_, F, J, P(tr), P(qo), None), 60 * J),
            ("mirror", lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, None, 0, F, J, P(qo), None), 32 * J),
            ("ik", lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(qo), None), 28 * J),
        ):
            ms, _ = pp.timeit(fn)
            line += f" | {name} {ms * 1e3:7.1f} us {F * nb / ms / 1e6 / 80:5.1f}%"
        del dq, tr, qo
    print(line, flush=True)
    del rot, pos, rm
