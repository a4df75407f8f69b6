#!/usr/bin/env python3
"""Where the parity error of the outer ops sits (GPU): prints max / quantiles / location of the error against the CPU oracle for
from_root_positions, mirror (sign agreement), to_root_dual_quat at both scales and the fused ortho6d -> fk chain.
A measuring aid for the tolerances asserted in tests/; not a test itself."""
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import c_oracle as co  # noqa: E402
from pymotion_amd import synthetic as syn  # noqa: E402
import pymotion_amd.ops.skeleton as sk  # noqa: E402


def ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def sre(a, b):
    return np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1))


def q(e):
    e = e[e > 0]
    if e.size == 0:
        return "all zero"
    return "max %.2e  p99.99 %.2e  p99.9 %.2e  p99 %.2e  median %.2e" % (e.max(), np.quantile(e, 0.9999), np.quantile(e, 0.999), np.quantile(e, 0.99), np.median(e))


def ik():
    from conftest import golden

    g = golden("ik.npz")
    for case in ("from_root_positions_J22", "from_root_positions_J52", "from_root_positions_topoJ9", "from_root_positions_starJ6"):
        i, want = g.get(case, "in"), g.get(case, "out64")["rot"]
        got = sk.from_root_positions(i["pos"], i["parents"], i["off"])
        e = sre(got, want)
        print(f"ik {case:34s} {q(e)}  worst joint {int(e.max(0).argmax())}")
    for J, F in ((22, 4099), (22, 200_000), (52, 20_000)):
        par = syn.PARENTS_22 if J == 22 else syn.PARENTS_52
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=9, normalized=True, offset_scale=0.3 if J == 22 else 0.15)
        pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
        pos = pos.astype(np.float32)
        got = sk.from_root_positions(pos, par, off)
        ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
        e = sre(got, ref)
        print(f"ik random {F} x {J:3d}                  {q(e)}  per-joint max (1e-6): {np.round(e.max(0) * 1e6, 1).tolist()}")
        # sensitivity of the REFERENCE to its own input rounding: positions moved by one fp32 ulp
        rng = np.random.default_rng(1)
        pos2 = np.nextafter(pos, np.where(rng.random(pos.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
        ref2 = co.from_root_positions(pos2.astype(np.float64), par, off.astype(np.float64))
        s = sre(ref2, ref)
        print(f"   reference moved by 1-ulp inputs:        {q(s)}")
        for k in (2, 3):
            pos3 = np.nextafter(pos, np.where(np.random.default_rng(k).random(pos.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
            s = np.maximum(s, sre(co.from_root_positions(pos3.astype(np.float64), par, off.astype(np.float64)), ref))
        bad = e > 2e-5
        ratio = (e[bad] - 2e-5) / np.maximum(s[bad], 1e-12)
        print(f"   records over 2e-5: {int(bad.sum())} of {bad.size}; (err - 2e-5) / sensitivity (max of 3 draws) over those: max {ratio.max() if bad.any() else 0:.1f}, "
              f"over 4: {int((ratio > 4).sum())}, over 8: {int((ratio > 8).sum())}, over 16: {int((ratio > 16).sum())}")
        p2, _ = sk.fk(got, np.zeros((F, 3), np.float32), off, par)
        p_ref, _ = co.fk(ref, np.zeros((F, 3)), off.astype(np.float64), par)
        print(f"   pose of the recovered rotations vs pose of the reference's: max {np.abs(p2 - p_ref).max():.2e}; vs the input positions: ours {np.abs(p2 - pos).max():.2e}, reference {np.abs(p_ref - pos).max():.2e}")
    g = golden("ik.npz")
    i, want = g.get("mirror_positions_X", "in"), g.get("mirror_positions_X", "out64")
    r, gt, o, _ = sk.mirror(i["rot"], i["root"], i["parents"], i["off"], None, None, "positions", "X")
    print(f"mirror(mode=positions) golden: {q(sre(r, want['rot']))}")


def mirror():
    for J in (22, 52, 130):
        rng = np.random.default_rng(J)
        parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
        if parents is None:
            parents = syn.random_parents(J, rng)
        F = 20_000 if J < 100 else 3000
        rot = rng.standard_normal((F, J, 4)).astype(np.float32)
        rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
        root = rng.uniform(-1, 1, (F, 3)).astype(np.float32)
        off = syn.make_offsets(J, rng, 0.1)
        got, *_ = sk.mirror(rot, root, parents, off, None, None, "all", "Y")
        _, rm = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), parents)
        g = co.quat_from_matrix(rm)
        sq = g * g
        neg = sq[..., 1] + sq[..., 2] > sq[..., 0] + sq[..., 3]
        margin = np.minimum(np.abs(sq[..., 1] + sq[..., 2] - sq[..., 0] - sq[..., 3]), np.where(neg, np.abs(sq[..., 1] - sq[..., 2]), np.abs(sq[..., 0] - sq[..., 3])))
        g[..., 1] *= -1
        g[..., 3] *= -1
        want = co.from_global_rotations(g, parents)
        same = np.abs(got - want).max(-1) <= 1e-5
        par = np.asarray(parents).copy()
        par[0] = 0
        m_el = np.minimum(margin, margin[:, par])
        print(f"mirror J={J:3d}: up-to-sign err {sre(got, want).max():.2e}; sign flips {int((~same).sum())} of {same.size}; "
              f"largest tie margin among flips {m_el[~same].max() if (~same).any() else 0:.2e}; elements with margin < 4e-6: {int((m_el < 4e-6).sum())}")


def dq():
    for J in (22, 52, 128, 12, 31, 96):
        parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
        if parents is None and J in (128, 12):
            parents = np.arange(-1, J - 1)
            parents[0] = 0
        elif parents is None:
            parents = syn.random_parents(J, np.random.default_rng(J))
        for osc, rsc in ((0.3, 2.0), (30.0, 200.0)):
            rng = np.random.default_rng(J)
            F = 6001
            rot = rng.standard_normal((F, J, 4))
            rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
            root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
            off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
            off[0] = 0
            d = sk.to_root_dual_quat(rot, root, parents, off)
            d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
            e = np.abs(d - d_o)
            print(f"to_root_dq J={J:3d} offsets {osc:5.1f}: max err {e.max():.2e} = {e.max() / ulp_of(d_o):.2f} ulp of the largest component "
                  f"(real part {e[..., :4].max():.2e}, dual part {e[..., 4:].max():.2e})")


def o6d():
    F, J = 1 << 14, 52
    rng = np.random.default_rng(4)
    x = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, np.random.default_rng(4), 0.15)
    p_o, r_o, q_o = co.fk_from_ortho6d(x.astype(np.float64), root.astype(np.float64), off.astype(np.float64), syn.PARENTS_52, return_quat=True)
    for want_q in (False, True):
        out = sk.fk_from_ortho6d(x, root, off, syn.PARENTS_52, return_quat=want_q)
        line = f"fused o6d->fk J=52 return_quat={want_q}: pos {np.abs(out[0] - p_o).max():.2e} rotmats {np.abs(out[1] - r_o).max():.2e}"
        if want_q:
            line += f" quat (up to sign) {sre(out[2], q_o).max():.2e}; sign flips {int((np.abs(out[2] - q_o).max(-1) > 1e-5).sum())} of {F * J}"
        print(line)
    import pymotion_amd.rotations.ortho6d as o6

    m = o6.to_matrix(x)
    print(f"o6d.to_matrix: {np.abs(m - co.o6d_to_matrix(x.astype(np.float64))).max():.2e};  to_quat up to sign: "
          f"{sre(o6.to_quat(x), co.o6d_to_quat(x.astype(np.float64))).max():.2e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["o6d", "mirror", "dq", "ik"]
    for w in which:
        {"ik": ik, "mirror": mirror, "dq": dq, "o6d": o6d}[w]()
