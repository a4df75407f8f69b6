#!/usr/bin/env python3
"""fk_wide_kernel with parts switched off (tuning build, PM_FK_ABLATE: 2 no walk, 8 no phase-A parks, 16 no copy-out, 32 the next frame requested after the copy-out instead of before the walk; PM_FKW_NT frames per workgroup): where a frame's time goes."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 20
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "160,256,384,512").split(",")]:
    par = syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
    F = 1 << 18
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    row = []
    for ab in (0, 2, 8, 16, 2 | 8, 2 | 16, 2 | 8 | 16):
        os.environ["PM_FK_ABLATE"] = str(ab)
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        row.append(f"ablate {ab:2d}: {ms * 1e3:7.1f} us")
    os.environ.pop("PM_FK_ABLATE")
    for nt, ab in ((1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (4, 32)):
        os.environ["PM_FKW_NT"] = str(nt); os.environ["PM_FK_ABLATE"] = str(ab)
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        row.append(f"nt {nt}{' no prefetch' if ab else ''}: {ms * 1e3:7.1f} us")
    os.environ.pop("PM_FKW_NT"); os.environ.pop("PM_FK_ABLATE")
    print(f"J={J:3d} bushy: " + " | ".join(row), flush=True)
    del rot, pos, rm
