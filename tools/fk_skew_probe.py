#!/usr/bin/env python3
"""does it matter that the eight XCDs' tile ranges start at power-of-two distances?  fk and the fused ortho6d kernel at 2^18 (SMPL-H) / 2^20 (22 joints)
frames and a few frames more (each XCD's eighth then starts a tile or more later in every array), production library, one process, two passes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for J, F0, par in ((52, 1 << 18, syn.PARENTS_52), (22, 1 << 20, syn.PARENTS_22)):
    par = np.ascontiguousarray(par, dtype=np.int32)
    Fm = F0 + 70000
    rot = torch.randn((Fm, J, 4), device="cuda"); root = torch.rand((Fm, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1
    x6 = torch.randn((Fm, J, 3, 2), device="cuda")
    pos = torch.empty((Fm, J, 3), device="cuda"); rm = torch.empty((Fm, J, 3, 3), device="cuda")
    for rep in range(2):
        for dF in (0, 3, 64, 128, 1000, 4096, 65536, -4096):
            F = F0 + dF
            row = []
            for name, fn, bpj in (("fk", lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None), 64),
                                  ("fused ortho6d", lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, C.c_float(0), P(pos), P(rm), None, None), 72)):
                ms, _ = pp.timeit(fn)
                row.append(f"{name} {ms * 1e3:7.1f} us {F * (bpj * J + 12) / ms / 1e6 / 80:5.1f} %")
            print(f"J={J} F = 2^{int(np.log2(F0))} {dF:+6d}: " + " | ".join(row), flush=True)
    del rot, x6, pos, rm
