#!/usr/bin/env python3
"""from_root_positions vs the oracle under every walk shape of the tuning build: where does the error come from?"""
import os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import c_oracle as co
import pymotion_amd.ops.skeleton as sk
from pymotion_amd import synthetic as syn, _lib
from test_ik import _reference_sensitivity

def star(K):
    return np.zeros(K + 1, dtype=np.int32)

def run(par, F, seed, scale, label):
    J = len(par)
    rot, root, off, par = syn.fk_workload(F, parents=par, seed=seed, normalized=True, offset_scale=scale)
    pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
    pos = pos.astype(np.float32)
    ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
    sens = _reference_sensitivity(pos, par, off, ref, draws=4)
    for env in ({"PM_IK_CHAINS": "1", "PM_IK_ORDER": "0"}, {"PM_IK_CHAINS": "2", "PM_IK_ORDER": "0"}, {"PM_IK_CHAINS": "4", "PM_IK_ORDER": "0"}, {"PM_IK_ORDER": "1"}):
        for k in ("PM_IK_CHAINS", "PM_IK_ORDER"):
            os.environ.pop(k, None)
        os.environ.update(env)
        got = sk.from_root_positions(pos, par, off)
        name = _lib.last_kernel_name().replace("void pm::from_root_positions_", "")
        err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
        over = err > 2e-5 + 8 * sens
        nk = np.bincount(par[1:], minlength=J)
        multi = nk >= 2
        print(f"{label:22s} {str(env):46s} {name:44s} over {int(over.sum()):4d}/{err.size}  max err {err.max():.2e}  p99.9 {np.quantile(err, 0.999):.2e}"
              f"  max err on joints with >= 2 kids {err[:, multi].max() if multi.any() else 0:.2e}, on single-child joints whose ancestors are all single {0:.0e}")

for K in (2, 3, 4, 6):
    run(star(K), 20000, K, 0.1, f"star of {K}")
run(syn.PARENTS_22, 4099, 9, 0.3, "J22 0.3")
run(syn.PARENTS_22, 4099, 9, 0.1, "J22 0.1")
run(syn.PARENTS_52, 3001, 9, 0.15, "J52 0.15")
run(syn.random_parents(72, np.random.default_rng(72)), 700, 72, 0.1, "random 72")
run(syn.random_parents(72, np.random.default_rng(72)), 700, 72, 0.3, "random 72 scale 0.3")
