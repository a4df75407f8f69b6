#!/usr/bin/env python3
"""to_root_dual_quat over joint counts (SMPL-H tree at 52, random trees elsewhere), sustained timing; with the tuning build
(PMHIP_VARIANT=tuning) PM_DQ_ABLATE / PM_DQ_FPW / PM_DQ_CHAINS apply."""
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
F = 1 << 18
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PM_DQ"))
for J in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "22,40,48,52,56,64,65,96,128".split(","))]:
    par = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if par is None:
        par = syn.random_parents(J, np.random.default_rng(J))
    Fj = F * 4 if J <= 24 else F
    rot = torch.randn((Fj, J, 4), device="cuda")
    rot /= rot.norm(dim=-1, keepdim=True)
    root = torch.randn((Fj, 3), device="cuda")
    # metre-scale bones (|t| < 1: the fp32 step) and the same skeleton in centimetres (the precise step, dq.hip)
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(J), 0.3)).cuda()
    dq = torch.empty((Fj, J, 8), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    ms, _ = pp.timeit(lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), Fj, J, P(dq), None))
    off_cm, root_cm = off * 100.0, root * 100.0
    ms_cm, _ = pp.timeit(lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root_cm), pp_, P(off_cm), Fj, J, P(dq), None))
    pct = lambda t: Fj * (48 * J + 12) / t / 1e6 / 80  # noqa: E731
    print(f"[{tag}] J={J:3d} depth={int(syn.depth_of(par).max()):2d}: to_root {ms * 1e3:7.1f} us {pct(ms):5.1f}%   centimetre data {ms_cm * 1e3:7.1f} us {pct(ms_cm):5.1f}%  "
          f"{_lib.last_kernel_name().split('(')[0][8:]}", flush=True)
    del rot, dq
