#!/usr/bin/env python3
"""fk_stream_kernel with parts of it switched off (tuning build, PM_FK_ABLATE: 2 the walk, 16 the partial first / last vectors of a
segment, 32 phase A's LDS writes), joint counts either side of a multiple of four: where do the 13 points between J = 128 and 129 go?"""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
os.environ["PM_FK_STREAM"] = "1"
for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "96,100,104,128,129,132,144,160,250,252,256").split(",")]:
    par = chain_like(J)
    F = (1 << 19) if J <= 128 else (1 << 18)
    Jp = (J + 31) // 32 * 32
    rot = torch.randn((F, Jp, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, Jp, 3), device="cuda"); rm = torch.empty((F, Jp, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    row = []
    for abl in (0, 2, 64, 66, 194):
        os.environ["PM_FK_ABLATE"] = str(abl)
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        row.append(f"abl {abl:3d}: {ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%")
    print(f"J={J:3d}: " + " | ".join(row), flush=True)
    del rot, pos, rm
