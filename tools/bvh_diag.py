import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymotion_amd import _ops, _backend
import pymotion_amd.rotations.quat as quat
be = _backend.numpy_backend() if hasattr(_backend, "numpy_backend") else None
from pymotion_amd.io import bvh as B
_be = B._be
rng = np.random.default_rng(0)
for T, J in ((5, 3), (48, 22), (300, 22), (5000, 31)):
    e = np.cumsum(rng.normal(0, 8, (T, J, 3)), axis=0)
    order = np.array([["z", "x", "y"]] * J)
    order[1] = ["x", "y", "z"]
    a = _ops.bvh_rotations(_be(), e, order)
    b = quat.normalize(quat.unroll(_ops.quat_from_euler(_be(), np.radians(e), order, per_joint_table=True), axis=0))
    d = np.abs(a - b)
    s = np.abs(a + b)
    print(T, J, "max |a-b|", d.max(), "max min(|a-b|,|a+b|)", np.minimum(d.max(-1), s.max(-1)).max(), "frames with sign diff", int((d.max(-1) > 1e-3).any(-1).sum()))
    if d.max() > 1e-3:
        f, j = np.argwhere(d.max(-1) > 1e-3)[0]
        print("  first bad", f, j, a[f, j], b[f, j], "prev", a[f - 1, j], b[f - 1, j])
