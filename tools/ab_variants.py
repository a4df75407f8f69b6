#!/usr/bin/env python3
"""same-box A/B of two builds of the library in one process: libpmhip.so (prod) against libpmhip_tuning.so, alternating, fk on several skeletons"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymotion_amd import _lib, synthetic as syn
from tools.store_probe import sustained, p
from oracle import c_oracle as co
def chain_like(J):
    q = np.maximum(np.arange(J) - 1, 0).astype(np.int32); q[J // 2] = 0; q[3 * J // 4] = J // 4
    return q
LONG = os.environ.get("AB_LONG") == "1"
for J, F in (((96, 1 << 18), (128, 1 << 18), (130, 1 << 18), (192, 1 << 17), (256, 1 << 17)) if LONG else ((22, 1 << 20), (8, 1 << 21), (16, 1 << 20), (29, 1 << 19), (33, 1 << 19))):
    par = chain_like(J) if LONG else (syn.PARENTS_22 if J == 22 else syn.random_parents(J, np.random.default_rng(J)))
    for scale in (1.0, 100.0):
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=0)
        off = off * scale; root = root * scale
        r_d, g_d, o_d = (torch.from_numpy(x).cuda() for x in (rot, root, off))
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        pp = par.astype(np.int32).ctypes.data_as(C.c_void_p)
        fk = lambda: _lib.call("pm_fk_f32", p(r_d), p(g_d), p(o_d), 0, pp, F, J, p(pos), p(rm), None)
        n = 1 << 13
        p_o, r_o = co.fk(rot[:n].astype(np.float64), root[:n].astype(np.float64), off.astype(np.float64), par)
        row = []
        for rep in range(2):
            for var in ("prod", "tuning"):
                with _lib.variant(var):
                    ms = sustained(fk)
                    ep = np.abs(pos[:n].cpu().numpy() - p_o).max(); er = np.abs(rm[:n].cpu().numpy() - r_o).max()
                row.append(f"{var} {ms * 1e3:6.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}% ({ep:.1e} {er:.1e})")
        print(f"J={J} x{scale:3.0f}: " + " | ".join(row), flush=True)
