#!/usr/bin/env python3
"""fk on long skeletons (chain-like and random trees, 2^19 / 2^18 frames): the streamed walk (fk_stream_kernel, chunks of 24 / 32 joints) against the
tile kernels it replaces (PM_FK_STREAM=0), same box, tuning build; the last column is the production dispatch."""
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
from pymotion_amd import synthetic as syn
for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,80,96,97,100,104,112,120,127,128,129,130,132,144,160,161,192,200,250,252,256,300,384,400,511,512").split(",")]:
    for kind in ("chain", "bushy"):
        par = chain_like(J) if kind == "chain" else syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
        F = (1 << 19) if J <= 128 else (1 << 18)
        rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        pp_ = par.ctypes.data_as(C.c_void_p)
        row = []
        for env in ({"PM_FK_STREAM": "0"}, {"PM_FK_STREAM": "1", "PM_FKS_CHS": "24"}, {"PM_FK_STREAM": "1", "PM_FKS_CHS": "32"}, {}):
            for k in ("PM_FK_STREAM", "PM_FKS_CHS"): os.environ.pop(k, None)
            os.environ.update(env)
            ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
            name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
            row.append(f"{ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}% {name[:38]:38s}")
        print(f"J={J:3d} {kind:5s}: " + " | ".join(row), flush=True)
        del rot, pos, rm
