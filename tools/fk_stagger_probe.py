#!/usr/bin/env python3
"""tuning build, PM_FK_ABLATE=64: the first round of the SMPL-H fk launch staggered (one / two / three tiles per workgroup in turn) against the plain
launch -- same process, alternating; results must be identical to the bit."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
J = 52
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
for F in (1 << 18, (1 << 18) + 3, 1 << 17, 1 << 20):
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1
    x6 = torch.randn((F, J, 3, 2), device="cuda")
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    fns = {"fk": lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None),
           "fused ortho6d": lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, C.c_float(0), P(pos), P(rm), None, None)}
    for name, fn in fns.items():
        outs = {}
        for rep in range(2):
            for ab in ("0", "64"):
                os.environ["PM_FK_ABLATE"] = ab
                pos.zero_(); rm.zero_()
                ms, _ = pp.timeit(fn)
                outs[ab] = (pos.clone(), rm.clone())
                print(f"F={F:8d} {name:14s} PM_FK_ABLATE={ab:2s}: {ms * 1e3:7.1f} us", flush=True)
        print("   bits equal:", bool(torch.equal(outs["0"][0].view(torch.int32), outs["64"][0].view(torch.int32)) and torch.equal(outs["0"][1].view(torch.int32), outs["64"][1].view(torch.int32))))
    os.environ.pop("PM_FK_ABLATE")
    del rot, x6, pos, rm
