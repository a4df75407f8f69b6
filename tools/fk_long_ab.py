#!/usr/bin/env python3
"""fk on long chain-like skeletons, SAME process / box / buffers, several builds of the library (PMHIP variants: prod, ab, ab2 --
tools/ab_file.sh): the streamed walk's timing depends on where the allocator put the arrays (boxes and runs differ by 5-8 points
on this kernel while the tile kernels repeat to a point), so variants are only compared inside one process.
    python tools/fk_long_ab.py 96,100,128,129,... prod,ab,ab2"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
Js = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "96,100,112,128,129,130,144,160,200,250,256,300,400,511,512").split(",")]
variants = (sys.argv[2] if len(sys.argv) > 2 else "prod,ab,ab2").split(",")
for J in Js:
    par = chain_like(J)
    F = (1 << 19) if J <= 128 else (1 << 18)
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    row = []
    for rep in range(2):
        for v in variants:
            with _lib.variant(v):
                ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
                name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
            row.append(f"{v}: {ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%")
    print(f"J={J:3d} {name[:34]:34s} | " + " | ".join(row), flush=True)
    del rot, pos, rm
