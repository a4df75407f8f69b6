#!/usr/bin/env python3
"""from_root_positions on tables that are parents-first but not depth first: SMPL-H as stored (52 joints, level order), the SMPL body (24), a 55-joint
SMPL-X-like table, the same trees relabelled depth first, the 22-joint BVH body and chain-like depth-first skeletons (PMHIP_VARIANT=tuning PM_IK_ORDER=0: the tile kernels)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 60
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

hand = lambda w, b: [w if k % 3 == 0 else b + k - 1 for k in range(15)]  # noqa: E731
smpl24 = np.concatenate([syn.PARENTS_52[:22], [20, 21]]).astype(np.int32)
smplx55 = np.asarray(list(syn.PARENTS_52[:22]) + [15, 15, 15] + hand(20, 25) + hand(21, 40), np.int32)
for name, par in (("SMPL-H level order (as stored)", syn.PARENTS_52), ("SMPL-H relabelled depth first", dfs_relabel(syn.PARENTS_52)), ("SMPL body, 24 joints, level order", smpl24),
                  ("SMPL-X-like, 55 joints, level order", smplx55), ("SMPL-X-like relabelled depth first", dfs_relabel(smplx55)), ("22-joint BVH body", syn.PARENTS_22)):
    J = len(par)
    for lf in ((18, 20) if J > 24 else (20,)):
        F = 1 << lf
        pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
        pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
        ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(out), None))
        print(f"{name:36s} 2^{lf} x {J:3d} {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')}", flush=True)
        del pos, out

def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p

for J in (24, 32, 40, 52, 64, 96, 128):
    par = chain_like(J); F = 1 << 19
    pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), par.ctypes.data_as(C.c_void_p), P(off), F, J, P(out), None))
    print(f"{'chain-like (depth first)':36s} 2^19 x {J:3d} {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')}", flush=True)
    del pos, out
