#!/usr/bin/env python3
"""from_root_positions: chains per frame (tuning build, PM_IK_CHAINS = 0 dispatch / 2 / 4; eight were built and measured in round 5, profiles/r05_ik_chains8.txt) on random trees, humanoids with hands and SMPL-H;
2^18 frames (2^20 up to 24 joints), % of 8 TB/s on 28 J bytes per frame; the results of the forced variants against the dispatch's, bit for bit."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
os.environ["PM_IK_ORDER"] = os.environ.get("PM_IK_ORDER", "0")  # the tile kernels only (the lane-per-frame order kernel has no chains)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from tools.fk_wide_sweep import humanoid
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "22,40,52,64,80,96,128,200").split(",")]:
    for kind in ("bushy", "humanoid"):
        par = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J) if kind == "humanoid" else None
        if par is None:
            par = humanoid(J) if kind == "humanoid" else syn.random_parents(J, np.random.default_rng(J))
        par = np.ascontiguousarray(par, dtype=np.int32)
        F = (1 << 20) if J <= 24 else (1 << 18)
        # positions of a real pose (fk of random rotations), root-centred: what the op is for
        rot = torch.randn((F, J, 4), device="cuda"); root = torch.zeros((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.2; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        pp_ = par.ctypes.data_as(C.c_void_p)
        _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None)
        del rot, rm
        out = torch.empty((F, J, 4), device="cuda")
        row, outs = [], []
        for ch in ("0", "2", "4"):
            os.environ["PM_IK_CHAINS"] = ch
            ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(out), None))
            name = _lib.last_kernel_name().replace("void pm::from_root_positions_kernel", "k").split("(")[0]
            row.append(f"chains {ch}: {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}% {name:18s}")
            outs.append(out.clone())
        same = all(bool(torch.equal(outs[0].view(torch.int32), o.view(torch.int32))) for o in outs[1:])
        print(f"J={J:3d} {kind:8s}: " + " | ".join(row) + f" | bits equal: {same}", flush=True)
        del pos, out, outs
