#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import c_oracle as co
import pymotion_amd.ops.skeleton as sk
from pymotion_amd import synthetic as syn, _lib
from test_ik import _reference_sensitivity

K, F, scale = 3, 200000, 0.1
par = np.zeros(K + 1, dtype=np.int32)
rot, root, off, par = syn.fk_workload(F, parents=par, seed=K, normalized=True, offset_scale=scale)
pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
pos = pos.astype(np.float32)
ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
got = sk.from_root_positions(pos, par, off)
err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))[:, 0]
sens = _reference_sensitivity(pos, par, off, ref, draws=4)[:, 0]
print("offsets", off, "lengths", np.linalg.norm(off, axis=1))
def ang(a, b):
    a = a / np.linalg.norm(a, axis=-1, keepdims=True); b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    return np.degrees(np.arctan2(np.linalg.norm(np.cross(a, b), axis=-1), (a * b).sum(-1)))
P = pos.astype(np.float64)
u1, v1 = off[1].astype(np.float64), P[:, 1] - P[:, 0]
align_angle = ang(np.broadcast_to(u1, v1.shape), v1)
print("quantiles of err: p50 %.2e p99 %.2e p99.9 %.2e p99.99 %.2e max %.2e" % tuple(np.quantile(err, q) for q in (0.5, 0.99, 0.999, 0.9999, 1.0)))
for f in np.argsort(-err)[:15]:
    print(f"frame {f} err {err[f]:.2e} sens {sens[f]:.2e} alignment angle {align_angle[f]:.4f} deg  ref {ref[f, 0]} got {got[f, 0]}")
# error vs alignment angle
for lo, hi in ((0, 1), (1, 5), (5, 30), (30, 150), (150, 175), (175, 179), (179, 179.9), (179.9, 180)):
    m = (align_angle >= lo) & (align_angle < hi)
    if m.any():
        print(f"alignment angle in [{lo}, {hi}): n {int(m.sum()):6d} max err {err[m].max():.2e} max (err - 8 sens) {np.max(err[m] - 8 * sens[m]):.2e}")
