import os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from oracle import c_oracle as co
import pymotion_amd.ops.skeleton as sk
from pymotion_amd import _lib, synthetic as syn
from test_ik import _windowed_tree, _reference_sensitivity

for kind in ("win6_96", "win2_128", "win4_64", "win3_40"):
    rng = np.random.default_rng(len(kind))
    par = _windowed_tree(int(kind.split("_")[1]), int(kind[3]), rng)
    J = len(par)
    for F in (64, 400, 4000):
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=J + F, normalized=True, offset_scale=0.1)
        pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
        pos = pos.astype(np.float32)
        ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
        sens = _reference_sensitivity(pos, par, off, ref, draws=12)
        for env in ({}, {"PM_IK_ORDER": "0"}):
            for k in list(os.environ):
                if k.startswith("PM_IK"): del os.environ[k]
            os.environ.update(env)
            got = sk.from_root_positions(pos, par, off)
            name = _lib.last_kernel_name().replace("void pm::from_root_positions_", "")[:30]
            err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
            ratio = (err - 2e-5) / np.maximum(sens, 1e-12)
            over = np.argwhere(err > 2e-5)
            print(kind, F, name, "max err %.3g" % err.max(), "n over 2e-5:", len(over), "max ratio %.3g" % ratio.max(), "median %.2g" % np.median(err))
            for (f, j) in over[:6]:
                print("    frame", f, "joint", j, "err %.3g sens %.3g" % (err[f, j], sens[f, j]))
