#!/usr/bin/env python3
"""fk's four-frame pipelined kernel: four JOINTS of a frame at a time (tree_walk_w4, PM_FK_W4=1) against one joint at a time with twelve lanes
(col 1: PM_FK_W4=0, whatever shape the dispatch picks; col 2: the four-frame kernel forced with PM_FK_W4=1), same box, same arrays, tuning build; the last column is the production dispatch.  FKW_KINDS=humanoid,bushy,chain,
FKW_SRC=quat|o6d."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from tools.fk_wide_sweep import humanoid, chain_like  # noqa: E402
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
kinds = (os.environ.get("FKW_KINDS") or "humanoid,bushy,chain").split(",")
src = os.environ.get("FKW_SRC", "quat")
for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "24,28,32,36,40,44,48,52,56,64,72,80,92").split(",")]:
    for kind in kinds:
        par = syn.PARENTS_52 if (kind == "humanoid" and J == 52) else chain_like(J) if kind == "chain" else humanoid(J) if kind == "humanoid" else syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
        par = np.ascontiguousarray(par, dtype=np.int32)
        depth = int(syn.depth_of(par).max())
        F = 1 << 18
        rot = torch.randn((F, J, 4), device="cuda"); x6 = torch.randn((F, J, 3, 2), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        pp_ = par.ctypes.data_as(C.c_void_p)
        row, outs = [], []
        for env in ({"PM_FK_W4": "0"}, {"PM_FK_W4": "1", "PM_FK_FPW": "4", "PM_FK_WIDE": "0", "PM_FK_STREAM": "0"}, {}):
            for k in ("PM_FK_W4", "PM_FK_FPW", "PM_FK_WIDE", "PM_FK_STREAM"): os.environ.pop(k, None)
            os.environ.update(env)
            if src == "quat":
                call = lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None)  # noqa: E731
                bpf = 64 * J + 12
            else:
                call = lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(off), 0, pp_, F, J, C.c_float(1e-12), P(pos), P(rm), None, None)  # noqa: E731
                bpf = 72 * J + 12
            ms, _ = pp.timeit(call)
            name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
            row.append(f"{ms * 1e3:7.1f} us {F * bpf / ms / 1e6 / 80:5.1f}% {name[:26]:26s}")
            outs.append((pos.clone(), rm.clone()))
        same = bool(torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)) and torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32)))
        print(f"J={J:3d} {kind:8s} depth {depth:3d} {src}: " + " | ".join(row) + f" | bits equal: {same}", flush=True)
        del rot, pos, rm, outs, x6
