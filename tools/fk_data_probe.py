#!/usr/bin/env python3
"""does fk's time (SMPL-H, 2^18 frames) depend on the DATA?  the same three buffers, the source filled with zeros / random quaternions / unit quaternions,
alternating, one process (a follow-up of tools/fk_align_probe.py, whose 'one buffer' rows ran on uninitialised memory)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 30
J, F = 52, 1 << 18
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
root = torch.rand((F, 3), device="cuda") * 4 - 2
off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
src = torch.empty((F, J, 4), device="cuda"); pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
call = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
for rep in range(3):
    for tag, fill in (("zeros", lambda: src.zero_()), ("randn", lambda: src.normal_()), ("unit", lambda: src.copy_(torch.nn.functional.normalize(torch.randn_like(src), dim=-1))),
                      ("identity", lambda: src.copy_(torch.tensor([1.0, 0, 0, 0], device="cuda").expand_as(src))), ("randn x 1e-3", lambda: src.normal_().mul_(1e-3))):
        fill()
        ms, _ = pp.timeit(call)
        print(f"rep {rep} source = {tag:12s}: {ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%", flush=True)
