#!/usr/bin/env python3
"""from_root_positions on SMPL-H's level-order 52-joint table (and a DFS relabelling of the same tree) under every walk shape of the tuning build."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

def dfs_relabel(par):
    J = len(par); kids = [[] for _ in range(J)]
    for j in range(1, J): kids[par[j]].append(j)
    order = []
    def go(j):
        order.append(j)
        for c in kids[j]: go(c)
    go(0)
    new = {o: i for i, o in enumerate(order)}
    p2 = np.zeros(J, np.int32)
    for o in range(1, J): p2[new[o]] = new[par[o]]
    return p2

F = 1 << 18
for name, par in (("SMPL-H level order", syn.PARENTS_52), ("SMPL-H relabelled depth first", dfs_relabel(syn.PARENTS_52)), ("22-joint body", syn.PARENTS_22)):
    J = len(par)
    Fj = F if J > 24 else F * 4
    pos = torch.randn((Fj, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((Fj, J, 4), device="cuda")
    pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
    for env in ({}, {"PM_IK_CHAINS": "1"}, {"PM_IK_CHAINS": "2"}, {"PM_IK_CHAINS": "4"}, {"PM_IK_FPW": "64"}, {"PM_IK_FPW": "32"}, {"PM_IK_FPW": "16"}, {"PM_IK_ORDER": "1"}, {"PM_IK_ORDER": "0"},
                {"PM_IK_CHAINS": "2", "PM_IK_NT": "1"}, {"PM_IK_CHAINS": "2", "PM_IK_NT": "2"}, {"PM_IK_CHAINS": "4", "PM_IK_NT": "2"}):
        for k in list(os.environ):
            if k.startswith("PM_IK"): del os.environ[k]
        os.environ.update(env)
        try:
            ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), Fj, J, P(out), None))
            print(f"{name:30s} {str(env):44s} {ms * 1e3:7.1f} us {Fj * 28 * J / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')}", flush=True)
        except Exception as e:
            print(name, env, "error", str(e)[:80])
