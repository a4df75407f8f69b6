#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (this container only).

    python oracle/make_golden.py            # writes tests/golden/{elementwise,trig,skeleton,bvh,mirror,ik,time}.npz + synthetic22.bvh
    python oracle/make_golden.py --check    # also cross-checks oracle/ (C + NumPy) against the import

The reference (UPC-ViRVIG/pymotion v0.2.3, pure Python) lives at /root/reference and
never travels to the GPU box; what travels are the vectors written here: the exact fp32
inputs and the outputs the reference produced for them.  Per case three output sets:

    out64     reference NumPy path on the inputs up-cast to float64  (tight pin for the oracle)
    out_np    reference NumPy path on the fp32 inputs as given        (its mixed f32/f64 behaviour)
    out_t     reference torch-CPU twin on the fp32 inputs             (fp32)

Key layout inside an .npz:  "<case>|in|<name>", "<case>|out64|<name>", ...
Inputs that are integer (parents, euler order codes) are stored as int32 / uint8.
"""
import argparse
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import pymotion.ops.skeleton as sk  # noqa: E402
import pymotion.ops.skeleton_torch as skt  # noqa: E402
import pymotion.rotations.dual_quat as dq  # noqa: E402
import pymotion.rotations.dual_quat_torch as dqt  # noqa: E402
import pymotion.rotations.ortho6d as o6  # noqa: E402
import pymotion.rotations.ortho6d_torch as o6t  # noqa: E402
import pymotion.rotations.quat as qt  # noqa: E402
import pymotion.rotations.quat_torch as qtt  # noqa: E402

from pymotion_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Store:
    def __init__(self):
        self.d = {}

    def add(self, case, kind, **arrs):
        for k, v in arrs.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            self.d[f"{case}|{kind}|{k}"] = np.ascontiguousarray(v)

    def save(self, name):
        os.makedirs(OUT, exist_ok=True)
        path = os.path.join(OUT, name)
        np.savez_compressed(path, **self.d)
        print(f"{path}: {len(self.d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def up(a):
    return a.astype(np.float64) if a.dtype == np.float32 else a


def run3(store, case, ins, f_np, f_t, names, int_keys=()):
    """Run a reference function three ways and record everything."""
    store.add(case, "in", **ins)

    def call(f, conv):
        r = f(*[v if k in int_keys else conv(v) for k, v in ins.items()])
        return r if isinstance(r, tuple) else (r,)

    store.add(case, "out64", **dict(zip(names, call(f_np, up))))
    store.add(case, "out_np", **dict(zip(names, call(f_np, lambda v: v))))
    if f_t is not None:
        with torch.no_grad():
            store.add(case, "out_t", **dict(zip(names, call(f_t, T))))


# ---- input builders ------------------------------------------------------------------------------

def rand_quats(rng, n, unit=True):
    q = rng.standard_normal((n, 4)).astype(np.float32)
    if unit:
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
    return q


def branchy_matrices(rng, n):
    """Rotation matrices covering all four quat.from_matrix branches (quat.py:110-153), plus
    exact 180-degree turns and the identity."""
    q = rand_quats(rng, n)
    m = qt.to_matrix(q.astype(np.float64))
    specials = np.array(
        [
            np.eye(3),
            np.diag([1.0, -1.0, -1.0]),   # r22<0, r00>r11
            np.diag([-1.0, 1.0, -1.0]),   # r22<0, r00<=r11
            np.diag([-1.0, -1.0, 1.0]),   # r22>=0, r00<-r11
        ]
    )
    m = np.concatenate([specials, m], axis=0).astype(np.float32)
    r00, r11, r22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    branch = np.where(r22 < 0, np.where(r00 > r11, 0, 1), np.where(r00 < -r11, 2, 3))
    assert set(branch.tolist()) == {0, 1, 2, 3}, "all from_matrix branches must be hit"
    return m


def lit_chain():
    """The 3-joint chain of ops/tests/test_skeleton.py:237-245 / 326-331 (inputs only)."""
    offsets = np.array([[0, 0, 0], [0, 0, 1], [0, 0, 2]], dtype=np.float32)
    parents = np.array([0, 0, 1], dtype=np.int32)
    gpos = np.array([[0, 0, 0], [1, 1, 1]], dtype=np.float32)
    ident = np.tile(np.array([1, 0, 0, 0], dtype=np.float32), (2, 3, 1))

    def rx(a):
        return [[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]]

    def ry(a):
        return [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]

    def rz(a):
        return [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]

    mats = np.array(
        [[rx(np.pi / 2), ry(np.pi / 2), rz(np.pi / 2)], [ry(np.pi / 4), rz(np.pi / 4), rx(np.pi / 4)]]
    )
    rotq = qt.from_matrix(mats).astype(np.float32)
    return offsets, parents, gpos, ident, rotq


# ---- groups --------------------------------------------------------------------------------------------

def gen_elementwise():
    s = Store()
    rng = np.random.default_rng(1234)
    n = 64
    q = rand_quats(rng, n)
    qn = (rand_quats(rng, n, unit=False) * 3).astype(np.float32)  # non-unit
    qz = qn.copy()
    qz[::7] = 0  # zero quats
    v = rng.uniform(-2, 2, (n, 3)).astype(np.float32)

    run3(s, "normalize", {"q": qz}, qt.normalize, qtt.normalize, ["out"])
    run3(s, "length", {"q": qn}, qt.length, qtt.length, ["out"])
    run3(s, "to_matrix_unit", {"q": q}, qt.to_matrix, qtt.to_matrix, ["out"])
    run3(s, "to_matrix_nonunit", {"q": qn}, qt.to_matrix, qtt.to_matrix, ["out"])
    lit_q = np.array([[0.70710678, 0.70710678, 0, 0], [0.92387953, 0, 0.38268343, 0], [0, 0, 0, 1]], dtype=np.float32)
    run3(s, "to_matrix_lit", {"q": lit_q}, qt.to_matrix, qtt.to_matrix, ["out"])  # test_quat.py:205-233
    run3(s, "from_matrix", {"m": branchy_matrices(rng, n)}, qt.from_matrix, qtt.from_matrix, ["out"])
    run3(s, "mul", {"a": q, "b": rand_quats(rng, n)}, qt.mul, qtt.mul, ["out"])
    run3(s, "mul_nonunit", {"a": qn, "b": qz}, qt.mul, qtt.mul, ["out"])
    run3(s, "mul_vec", {"q": q, "v": v}, qt.mul_vec, qtt.mul_vec, ["out"])
    run3(s, "conjugate", {"q": qn}, qt.conjugate, qtt.conjugate, ["out"])
    run3(s, "inverse", {"q": q}, qt.inverse, qtt.inverse, ["out"])
    # broadcasting + multi-dim leading shape
    run3(s, "mul_bcast", {"a": q.reshape(4, 16, 4), "b": q[:16]}, qt.mul, qtt.mul, ["out"])

    run3(s, "dq_from_rt", {"q": q, "t": v}, dq.from_rotation_translation, dqt.from_rotation_translation, ["out"])
    d = dq.from_rotation_translation(q, v).astype(np.float32)
    run3(s, "dq_to_rt", {"dq": d}, dq.to_rotation_translation, dqt.to_rotation_translation, ["q", "t"])
    run3(s, "dq_from_t", {"t": v}, dq.from_translation, dqt.from_translation, ["out"])
    # dual_quat.normalize: one batch that is already unit (-> plain branch), one that is not
    # (-> the orthogonalising branch; the branch is decided for the WHOLE batch, dual_quat.py:106)
    d_scaled = (d * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    d_skew = d_scaled.copy()
    d_skew[:, 4:] += 0.1 * d_skew[:, :4]
    run3(s, "dq_normalize_scaled", {"dq": d_scaled}, dq.normalize, dqt.normalize, ["out"])
    run3(s, "dq_normalize_skew", {"dq": d_skew}, dq.normalize, dqt.normalize, ["out"])
    for nm, arr in (("unit", d), ("scaled", d_scaled), ("skew", d_skew), ("zero", np.zeros((4, 8), np.float32))):
        s.add(f"dq_is_unit_{nm}", "in", dq=arr)
        s.add(f"dq_is_unit_{nm}", "out64", out=np.array(bool(dq.is_unit(up(arr)))))
        s.add(f"dq_is_unit_{nm}", "out_np", out=np.array(bool(dq.is_unit(arr))))
        s.add(f"dq_is_unit_{nm}", "out_t", out=np.array(bool(dqt.is_unit(T(arr)))))

    x = rng.standard_normal((n, 3, 2)).astype(np.float32)
    x[1, :, 1] = x[1, :, 0] * 1.0001 + 1e-3  # near-parallel columns
    run3(s, "o6d_to_matrix", {"x": x}, o6.to_matrix, o6t.to_matrix, ["out"])
    run3(s, "o6d_to_quat", {"x": x}, o6.to_quat, o6t.to_quat, ["out"])
    run3(s, "o6d_from_quat", {"q": q}, o6.from_quat, o6t.from_quat, ["out"])
    run3(s, "o6d_from_matrix", {"m": branchy_matrices(rng, 8)}, o6.from_matrix, o6t.from_matrix, ["out"])
    # zero first column: NumPy -> NaN, torch (F.normalize eps=1e-12) -> finite zeros
    xz = x[:4].copy()
    xz[0, :, 0] = 0
    with np.errstate(all="ignore"):
        run3(s, "o6d_to_matrix_zero_col", {"x": xz}, o6.to_matrix, o6t.to_matrix, ["out"])
    s.save("elementwise.npz")


def gen_trig():
    s = Store()
    rng = np.random.default_rng(4321)
    n = 64
    axis = rng.standard_normal((n, 3)).astype(np.float32)
    axis /= np.linalg.norm(axis, axis=-1, keepdims=True)
    angle = rng.uniform(0, 2 * np.pi, (n, 1)).astype(np.float32)
    run3(s, "from_angle_axis", {"angle": angle, "axis": axis}, qt.from_angle_axis, qtt.from_angle_axis, ["out"])
    # test_quat.py:54-69 literals
    run3(
        s, "from_angle_axis_lit",
        {"angle": np.array([0, np.pi / 2, np.pi / 4, np.pi], dtype=np.float32)[:, None],
         "axis": np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)},
        qt.from_angle_axis, qtt.from_angle_axis, ["out"],
    )
    sa = (axis * angle).astype(np.float32)
    run3(s, "from_scaled_angle_axis", {"v": sa}, qt.from_scaled_angle_axis, qtt.from_scaled_angle_axis, ["out"])
    q = rand_quats(rng, n)
    q[0] = [1, 0, 0, 0]  # identity -> axis 0 (mask s>1e-8)
    q[1] = [-1, 0, 0, 0]
    run3(s, "to_angle_axis", {"q": q}, qt.to_angle_axis, qtt.to_angle_axis, ["angle", "axis"])
    run3(s, "to_scaled_angle_axis", {"q": q}, qt.to_scaled_angle_axis, qtt.to_scaled_angle_axis, ["out"])

    orders = ["xyz", "xzy", "yxz", "yzx", "zxy", "zyx"]
    e = rng.uniform(-np.pi, np.pi, (n, 3)).astype(np.float32)
    order_s = np.array([list(orders[i % 6]) for i in range(n)])
    order_c = np.vectorize(lambda c: "xyz".index(c), otypes=[np.uint8])(order_s)

    def fe_np(e_, o_):
        return qt.from_euler(e_, order_s)

    def fe_t(e_, o_):
        return qtt.from_euler(e_, order_s)

    run3(s, "from_euler", {"e": e, "order": order_c}, fe_np, fe_t, ["out"], int_keys=("order",))
    qe = qt.from_euler(e.astype(np.float64), order_s).astype(np.float32)

    def te_np(q_, o_):
        return qt.to_euler(q_, order_s)

    def te_t(q_, o_):
        return qtt.to_euler(q_, order_s)

    run3(s, "to_euler", {"q": qe, "order": order_c}, te_np, te_t, ["out"], int_keys=("order",))

    q0, q1 = rand_quats(rng, n), rand_quats(rng, n)
    t = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    for sh in (True, False):
        run3(
            s, f"slerp_shortest{int(sh)}", {"q0": q0, "q1": q1, "t": t},
            lambda a, b, c, sh=sh: qt.slerp(a, b, c, shortest=sh),
            lambda a, b, c, sh=sh: qtt.slerp(a, b, c, shortest=sh), ["out"],
        )
    run3(s, "slerp_scalar_t", {"q0": q0, "q1": q1}, lambda a, b: qt.slerp(a, b, 0.3), lambda a, b: qtt.slerp(a, b, 0.3), ["out"])
    s.save("trig.npz")


def skel_case(s, case, rot, gpos, off, parents, dq_ok=True):
    ins = {"rot": rot, "gpos": gpos, "off": off, "parents": parents}
    run3(s, f"fk_{case}", ins, sk.fk, lambda r, g, o, p: skt.fk(r, g, o, T(p)), ["pos", "rotmats"], int_keys=("parents",))
    if not dq_ok:
        return
    # dq path needs [F,J,4] (reference uses shape[1]) and static offsets with offsets[0]==0
    ins2 = {"rot": rot, "gpos": gpos, "parents": parents, "off": off}
    run3(s, f"to_root_dq_{case}", ins2, sk.to_root_dual_quat,
         lambda r, g, p, o: skt.to_root_dual_quat(r, g, T(p), o), ["dq"], int_keys=("parents",))
    d = sk.to_root_dual_quat(up(rot), up(gpos), parents, up(off)).astype(np.float32)
    run3(s, f"from_root_dq_{case}", {"dq": d, "parents": parents}, sk.from_root_dual_quat,
         lambda a, p: skt.from_root_dual_quat(a, T(p)), ["trans", "rot"], int_keys=("parents",))
    g = rand_quats(np.random.default_rng(7), rot.shape[0] * rot.shape[1]).reshape(rot.shape)
    run3(s, f"from_global_rot_{case}", {"gq": g, "parents": parents}, sk.from_global_rotations,
         lambda a, p: skt.from_global_rotations(a, T(p)), ["out"], int_keys=("parents",))


def gen_skeleton():
    s = Store()
    off3, par3, gpos3, ident, rotq = lit_chain()
    skel_case(s, "lit_identity", ident, gpos3, off3, par3)
    skel_case(s, "lit_rot", rotq, gpos3, off3, par3)
    # per-frame offsets [F,J,3] (test_skeleton.py:267) -- fk only
    skel_case(s, "lit_per_frame_off", rotq, gpos3, np.tile(off3, (2, 1, 1)) * np.array([1.0, 2.0], np.float32)[:, None, None], par3, dq_ok=False)
    # parents[0] = -1 is ignored by fk (skeleton.py:53)
    pm1 = par3.copy()
    pm1[0] = -1
    skel_case(s, "lit_parent0_minus1", rotq, gpos3, off3, pm1, dq_ok=False)

    rng = np.random.default_rng(2024)
    for name, parents, F in (
        ("rand_J22", syn.PARENTS_22, 64),
        ("rand_J52", syn.PARENTS_52, 24),
        ("rand_topo_J22", syn.random_parents(22, rng), 32),
        ("rand_topo_J7", syn.random_parents(7, rng), 33),
        ("chain_J12", np.maximum(np.arange(12) - 1, 0).astype(np.int32), 21),
        ("star_J9", np.zeros(9, dtype=np.int32), 20),
        ("single_J1", np.zeros(1, dtype=np.int32), 5),
    ):
        J = len(parents)
        rot = rng.standard_normal((F, J, 4)).astype(np.float32)
        gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
        off = syn.make_offsets(J, rng)
        # fk normalises internally -> raw gaussians; dq path does not -> unit quats
        run3(s, f"fk_{name}_raw", {"rot": rot, "gpos": gpos, "off": off, "parents": parents}, sk.fk,
             lambda r, g, o, p: skt.fk(r, g, o, T(p)), ["pos", "rotmats"], int_keys=("parents",))
        skel_case(s, name, rot / np.linalg.norm(rot, axis=-1, keepdims=True), gpos, off, parents)
    # zero quaternions -> identity rotation in fk (quat.py:423 eps placement)
    rot = rng.standard_normal((8, 22, 4)).astype(np.float32)
    rot[::3, ::5] = 0
    skel_case(s, "zero_quat_J22", rot, rng.uniform(-2, 2, (8, 3)).astype(np.float32), syn.make_offsets(22, rng), syn.PARENTS_22, dq_ok=False)
    # 6-D leading shape (test_skeleton.py:386-404) -- fk only (dq path is [F,J,4]-only in the reference)
    rot = rng.standard_normal((2, 3, 2, 2, 22, 4)).astype(np.float32)
    skel_case(s, "multidim_J22", rot, rng.uniform(-2, 2, (2, 3, 2, 2, 3)).astype(np.float32), syn.make_offsets(22, rng), syn.PARENTS_22, dq_ok=False)

    # config-4 composite: ortho6d.to_quat -> fk on the 52-joint tree
    x, gpos, off, par = syn.o6d_workload(16, seed=5)

    def comp_np(x_, g_, o_, p_):
        q_ = o6.to_quat(x_)
        return sk.fk(q_, g_, o_, p_) + (q_,)

    def comp_t(x_, g_, o_, p_):
        q_ = o6t.to_quat(x_)
        return skt.fk(q_, g_, o_, T(p_)) + (q_,)

    run3(s, "fk_from_o6d_J52", {"x": x, "gpos": gpos, "off": off, "parents": par}, comp_np, comp_t,
         ["pos", "rotmats", "quat"], int_keys=("parents",))
    s.save("skeleton.npz")


def gen_bvh():
    from pymotion.io.bvh import BVH

    s = Store()
    path = os.path.join(OUT, "synthetic22.bvh")
    syn.write_synthetic_bvh(path)
    b = BVH()
    b.load(path)
    d = b.data
    s.add("load", "out64", offsets=d["offsets"], end_sites=d["end_sites"], end_sites_parents=d["end_sites_parents"].astype(np.int32),
          parents=d["parents"].astype(np.int32), positions=d["positions"], rotations=d["rotations"],
          rot_order=np.vectorize(lambda c: "xyz".index(c), otypes=[np.uint8])(d["rot_order"]),
          frame_time=np.array(d["frame_time"]))
    s.add("load", "in", names=np.array([n.encode() for n in d["names"]]))
    rots, pos, parents, offsets, es, esp = b.get_data()
    s.add("get_data", "out64", rots=rots, pos=pos)
    p, r = sk.fk(rots, pos[:, 0, :], offsets, parents)  # README.md:73-78
    s.add("fk", "out64", pos=p, rotmats=r)
    b.set_data(rots, pos)
    s.add("set_data", "out64", rotations=b.data["rotations"])
    # unroll on its own: random unit quaternions with random sign flips, several axes; dual quats too
    rng = np.random.default_rng(11)
    q = rand_quats(rng, 300 * 5).reshape(300, 5, 4).astype(np.float32)
    base = np.cumsum(rng.normal(0, 0.05, (300, 5, 4)), axis=0) + rng.normal(0, 1, (1, 5, 4))
    base /= np.linalg.norm(base, axis=-1, keepdims=True)
    flips = rng.choice([-1.0, 1.0], (300, 5, 1))
    qs = (base * flips).astype(np.float32)  # a smooth path with random cover flips
    for nm, arr, ax in (("unroll_smooth_ax0", qs, 0), ("unroll_random_ax0", q, 0),
                        ("unroll_ax1", np.ascontiguousarray(qs.transpose(1, 0, 2)), 1),
                        ("unroll_4d_ax-3", np.ascontiguousarray(qs[:, None, :, :].repeat(2, axis=1).transpose(1, 0, 2, 3)), -3)):
        s.add(nm, "in", q=arr, axis=np.array(ax))
        s.add(nm, "out64", out=qt.unroll(arr.astype(np.float64).copy(), ax))
        s.add(nm, "out_t", out=qtt.unroll(T(arr.copy()), ax))
    t3 = rng.uniform(-1, 1, (300, 5, 3)).astype(np.float32)
    d8 = dq.from_rotation_translation(qs, t3).astype(np.float32)
    s.add("dq_unroll_ax0", "in", dq=d8, axis=np.array(0))
    s.add("dq_unroll_ax0", "out64", out=dq.unroll(d8.astype(np.float64).copy(), 0))
    s.save("bvh.npz")


def gen_mirror():
    """tests/golden/mirror.npz: reference mirror(mode in {'all','symmetry'}, axis in XYZ) on a 22-joint pose."""
    d = {}
    rng = np.random.default_rng(77)
    rot, root, off, par = syn.fk_workload(40, seed=77, normalized=True)
    end = rng.uniform(-0.1, 0.1, (5, 3)).astype(np.float32)
    mp = np.arange(22)
    mp[1:5], mp[5:9], mp[14:18], mp[18:22] = np.arange(5, 9), np.arange(1, 5), np.arange(18, 22), np.arange(14, 18)
    for ax in "XYZ":
        for mode in ("all", "symmetry"):
            r, g, o, e = sk.mirror(rot.astype(np.float64), root.astype(np.float64).copy(), par, off.astype(np.float64),
                                   end.astype(np.float64), mp if mode == "symmetry" else None, mode, ax)
            k = f"mirror_{mode}_{ax}"
            d[k + "|out64|rot"], d[k + "|out64|gt"], d[k + "|out64|off"], d[k + "|out64|end"] = r, g, o, e
    for k, v in (("rot", rot), ("root", root), ("off", off), ("parents", par), ("end", end), ("mapping", mp.astype(np.int32))):
        d["inputs|in|" + k] = v
    np.savez_compressed(os.path.join(OUT, "mirror.npz"), **d)


def gen_ik():
    """tests/golden/ik.npz: quat.from_to / from_to_axis (incl. the parallel / anti-parallel branches),
    from_root_positions on poses produced by fk, mirror(mode='positions')."""
    s = Store()
    rng = np.random.default_rng(31)
    n = 64
    v1 = rng.standard_normal((n, 3)).astype(np.float32)
    v2 = rng.standard_normal((n, 3)).astype(np.float32)
    v2[0] = v1[0] * 2.5                       # parallel
    v2[1] = -v1[1] * 0.5                      # anti-parallel, generic v1
    v1[2] = [3.0, 0.0, 0.0]; v2[2] = [-1.0, 0.0, 0.0]   # anti-parallel along x (other orthogonal helper)
    v1[3] = [0.0, 2.0, 0.0]; v2[3] = [0.0, 0.0, 5.0]    # 90 degrees
    ax = rng.standard_normal((n, 3)).astype(np.float32)
    ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
    run3(s, "from_to", {"v1": v1, "v2": v2}, qt.from_to, qtt.from_to, ["out"])
    run3(s, "from_to_nonorm", {"v1": v1 / np.linalg.norm(v1, axis=-1, keepdims=True), "v2": v2 / np.linalg.norm(v2, axis=-1, keepdims=True)},
         lambda a, b: qt.from_to(a, b, False), lambda a, b: qtt.from_to(a, b, False), ["out"])
    run3(s, "from_to_axis", {"v1": v1, "v2": v2, "axis": ax}, qt.from_to_axis, qtt.from_to_axis, ["out"])
    run3(s, "from_to_single", {"v1": v1[5], "v2": v2[5]}, qt.from_to, qtt.from_to, ["out"])
    for name, parents, F in (("J22", syn.PARENTS_22, 24), ("J52", syn.PARENTS_52, 8), ("topoJ9", syn.random_parents(9, rng), 16),
                             ("starJ6", np.zeros(6, dtype=np.int32), 12)):
        J = len(parents)
        rot = rng.standard_normal((F, J, 4))
        rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
        off = syn.make_offsets(J, rng).astype(np.float64)
        pos, _ = sk.fk(rot, np.zeros((F, 3)), off, parents)
        pos = np.ascontiguousarray(pos).astype(np.float32)
        run3(s, f"from_root_positions_{name}", {"pos": pos, "parents": parents, "off": off.astype(np.float32)},
             sk.from_root_positions, lambda p_, par_, o_: skt.from_root_positions(p_, T(par_), o_), ["rot"], int_keys=("parents",))
    rot, root, off, par = syn.fk_workload(12, seed=5, normalized=True)
    r, g, o, e = sk.mirror(rot.astype(np.float64), root.astype(np.float64).copy(), par, off.astype(np.float64), None, None, "positions", "X")
    s.add("mirror_positions_X", "in", rot=rot, root=root, off=off, parents=par)
    s.add("mirror_positions_X", "out64", rot=r, gt=g, off=o)
    s.save("ik.npz")


def gen_skeleton_extra():
    """tests/golden/skeleton_extra.npz: cases added after skeleton.npz was frozen.  fk with a NON-ZERO offsets[0]
    (ignored by the reference: positions[..., 0, :] is overwritten by global_pos, skeleton.py:49), shared and
    per-frame offsets, on both walk shapes (J <= 23 / J >= 24) and the single-joint skeleton."""
    s = Store()
    rng = np.random.default_rng(4049)
    for name, parents, F in (("J1", np.zeros(1, dtype=np.int32), 5), ("J2", np.zeros(2, dtype=np.int32), 7),
                             ("J22", syn.PARENTS_22, 33), ("J24", syn.random_parents(24, rng), 9), ("J52", syn.PARENTS_52, 6)):
        J = len(parents)
        rot = rng.standard_normal((F, J, 4)).astype(np.float32)
        gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
        off = rng.uniform(-0.3, 0.3, (J, 3)).astype(np.float32)  # row 0 deliberately not zero
        run3(s, f"fk_off0_{name}", {"rot": rot, "gpos": gpos, "off": off, "parents": parents}, sk.fk,
             lambda r, g, o, p: skt.fk(r, g, o, T(p)), ["pos", "rotmats"], int_keys=("parents",))
        offf = rng.uniform(-0.3, 0.3, (F, J, 3)).astype(np.float32)
        run3(s, f"fk_off0_pf_{name}", {"rot": rot, "gpos": gpos, "off": offf, "parents": parents}, sk.fk,
             lambda r, g, o, p: skt.fk(r, g, o, T(p)), ["pos", "rotmats"], int_keys=("parents",))
    s.save("skeleton_extra.npz")


def degenerate_o6d(rng, F, J):
    """6D inputs with the records Gram-Schmidt cannot handle gracefully, at fixed (frame, joint) places."""
    x = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
    a = rng.standard_normal(3).astype(np.float32)
    noise = rng.standard_normal(3).astype(np.float32)
    special = {
        "zero_matrix": np.zeros((3, 2), np.float32),
        "zero_first_col": np.stack([np.zeros(3, np.float32), a], axis=1),
        "zero_second_col": np.stack([a, np.zeros(3, np.float32)], axis=1),
        "parallel": np.stack([a, 2 * a], axis=1),
        "anti_parallel": np.stack([a, -0.5 * a], axis=1),
        "near_parallel": np.stack([a, 1.5 * a + 1e-4 * noise], axis=1).astype(np.float32),
        "tiny": (1e-20 * rng.standard_normal((3, 2))).astype(np.float32),
        "huge": (1e15 * rng.standard_normal((3, 2))).astype(np.float32),
    }
    where = {}
    for k, (name, rec) in enumerate(special.items()):
        f, j = k % F, (1 + 3 * k) % J if name != "zero_matrix" else 0   # one of them on the root joint
        x[f, j] = rec
        where[name] = (f, j)
    return x, where


def gen_degenerate():
    """tests/golden/degenerate.npz: the chain ortho6d.to_quat -> fk on zero / parallel / tiny / huge columns (round-1
    verdict: the fused kernel must equal that chain on EVERY input, with and without the quaternion output)."""
    s = Store()
    rng = np.random.default_rng(2718)
    for J, parents, F in ((22, syn.PARENTS_22, 9), (52, syn.PARENTS_52, 8)):
        x, where = degenerate_o6d(rng, F, J)
        gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
        off = syn.make_offsets(J, rng, 0.3 if J == 22 else 0.15)

        def comp_np(x_, g_, o_, p_):
            with np.errstate(all="ignore"):
                q_ = o6.to_quat(x_)
                return sk.fk(q_, g_, o_, p_) + (q_,)

        def comp_t(x_, g_, o_, p_):
            q_ = o6t.to_quat(x_)
            return skt.fk(q_, g_, o_, T(p_)) + (q_,)

        case = f"fk_from_o6d_degenerate_J{J}"
        ins = {"x": x, "gpos": gpos, "off": off, "parents": parents}
        run3(s, case, ins, comp_np, comp_t, ["pos", "rotmats", "quat"], int_keys=("parents",))
        # the torch twin's semantics (F.normalize eps = 1e-12: zero column -> zeros, ortho6d_torch.py:84-89) at full
        # precision: what a float32 implementation is judged against on records whose answer lives below fp32's digits
        r = comp_t(T(up(x)), T(up(gpos)), T(up(off)), parents)
        s.add(case, "out_t64", **dict(zip(["pos", "rotmats", "quat"], r)))
        s.add(case, "in", where=np.array([[f, j] for f, j in where.values()], dtype=np.int32))
        # the element-wise conversions on the same records
        run3(s, f"o6d_to_quat_degenerate_J{J}", {"x": x}, lambda x_: o6.to_quat(x_), o6t.to_quat, ["out"])
        run3(s, f"o6d_to_matrix_degenerate_J{J}", {"x": x}, lambda x_: o6.to_matrix(x_), o6t.to_matrix, ["out"])
        s.add(f"o6d_to_quat_degenerate_J{J}", "out_t64", out=o6t.to_quat(T(up(x))))
        s.add(f"o6d_to_matrix_degenerate_J{J}", "out_t64", out=o6t.to_matrix(T(up(x))))
    # unroll with RESETS (round-1 ADVICE): a neighbour dot product that is exactly 0 or NaN is "not < 0" whatever sign the
    # previous frame ended up with (quat.py:453-458), so the accumulated sign starts over -- zero-padded rows, exact
    # 90/180-degree steps ([1,0,0,0] -> [0,1,0,0]), NaN rows; placed inside, at the edges of and across the kernels'
    # 256-frame chunks and 64-frame sub-tiles.
    import pymotion.rotations.dual_quat as dq_ref
    import pymotion.rotations.dual_quat_torch as dqt_ref
    T_, S_ = 700, 7
    q = rng.standard_normal((T_, S_, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    for t in range(1, T_):  # a smooth path ...
        q[t] = 0.9 * q[t - 1] + 0.1 * q[t]
        q[t] /= np.linalg.norm(q[t], axis=-1, keepdims=True)
    q[rng.random((T_, S_)) < 0.3] *= -1          # ... with random sign flips
    ex, ey = np.array([1, 0, 0, 0], np.float32), np.array([0, 1, 0, 0], np.float32)
    q[5, 0] = 0; q[63, 0] = 0; q[64, 0] = 0; q[255, 0] = 0      # zero rows: sub-tile edge, chunk edge
    q[256, 1] = 0; q[257, 1] = 0; q[511, 1] = 0; q[512, 1] = 0; q[699, 1] = 0
    q[100, 2] = -ex; q[101, 2] = ey; q[102, 2] = -ex            # exactly orthogonal steps after a flipped frame
    q[255, 3] = -ex; q[256, 3] = ey                              # ... across a chunk boundary
    q[300, 4] = np.nan; q[301, 4, 2] = np.nan                    # NaN rows
    q[1, 5] = 0                                                  # reset right after the first frame
    # series 6 stays regular
    run3(s, "unroll_resets", {"q": q}, lambda q_: qt.unroll(q_.copy(), 0), lambda q_: qtt.unroll(q_.clone(), 0), ["out"])
    d8 = np.concatenate([q, rng.standard_normal((T_, S_, 4)).astype(np.float32)], axis=-1)
    run3(s, "dq_unroll_resets", {"dq": d8}, lambda d_: dq_ref.unroll(d_.copy(), 0), lambda d_: dqt_ref.unroll(d_.clone(), 0), ["out"])
    s.save("degenerate.npz")


def gen_time():
    """tests/golden/time.npz: ops/time.py interpolate_positions -- the literal of the reference's own test
    (ops/tests/test_time.py:12-66) and seeded clips with non-uniform times, exact hits and extrapolation on
    both sides.  The time axis is the second to last one (the only layout the reference's broadcast supports)."""
    import pymotion.ops.time as tm
    import pymotion.ops.time_torch as tmt

    s = Store()
    pos = np.array([[[0, 0, 0], [1, 1, 0], [2, 0, 0], [8, 0, 1], [20, 0, 0]],
                    [[1, 1, 1], [1, 1, 0], [2, 0, 0], [8, 0, 1], [20, 0, 0]]], dtype=np.float32)[np.newaxis, np.newaxis, ...]
    x = np.array([0, 2, 3, 4, 5], dtype=np.float32)
    new_x = np.array([0.5, 1.75, 2.25, 3.75, 4, 5, 6, 7, 8], dtype=np.float32)
    run3(s, "interp_lit", {"sample": new_x, "orig": x, "pos": pos}, lambda a, b, c: tm.interpolate_positions(a, b, c, 3),
         lambda a, b, c: tmt.interpolate_positions(a, b, c, 3), ["out"])
    rng = np.random.default_rng(77)
    for name, lead, Tn, Sn in (("clip", (22,), 40, 97), ("lead", (2, 3), 17, 33), ("two", (5,), 2, 9)):
        orig = np.cumsum(rng.uniform(0.01, 0.1, Tn)).astype(np.float32)
        sample = np.sort(rng.uniform(orig[0] - 0.2, orig[-1] + 0.2, Sn)).astype(np.float32)
        sample[1] = orig[0]; sample[2] = orig[-1]; sample[3] = orig[Tn // 2]   # exact hits (searchsorted side='left')
        sample = rng.permutation(sample)                                          # sample times need not be sorted
        p = rng.uniform(-2, 2, lead + (Tn, 3)).astype(np.float32)
        ax = len(lead)
        run3(s, f"interp_{name}", {"sample": sample, "orig": orig, "pos": p},
             lambda a, b, c, ax=ax: tm.interpolate_positions(a, b, c, ax), lambda a, b, c, ax=ax: tmt.interpolate_positions(a, b, c, ax), ["out"])
    s.save("time.npz")


# ---- optional cross-check of oracle/ against the import ------------------------------------------------

def check_oracle():
    from oracle import c_oracle as co
    from oracle import numpy_ref as nr
    import time

    rng = np.random.default_rng(99)
    worst = 0.0
    for J, parents in ((3, np.array([0, 0, 1], np.int32)), (22, syn.PARENTS_22), (52, syn.PARENTS_52)):
        F = 100_000 if J == 22 else 20_000
        rot = rng.standard_normal((F, J, 4))
        gpos = rng.uniform(-2, 2, (F, 3))
        off = syn.make_offsets(J, rng).astype(np.float64)
        p_ref, r_ref = sk.fk(rot, gpos, off, parents)
        p_c, r_c = co.fk(rot, gpos, off, parents)
        p_n, r_n = nr.fk(rot, gpos, off, parents)
        e = max(np.abs(p_ref - p_c).max(), np.abs(r_ref - r_c).max(), np.abs(p_ref - p_n).max(), np.abs(r_ref - r_n).max())
        rn = rot / np.linalg.norm(rot, axis=-1, keepdims=True)
        d_ref = sk.to_root_dual_quat(rn, gpos, parents, off)
        e = max(e, np.abs(d_ref - co.to_root_dual_quat(rn, gpos, parents, off)).max(),
                np.abs(d_ref - nr.to_root_dual_quat(rn, gpos, parents, off)).max())
        t_ref, q_ref = sk.from_root_dual_quat(d_ref, parents)
        t_c, q_c = co.from_root_dual_quat(d_ref, parents)
        t_n, q_n = nr.from_root_dual_quat(d_ref, parents)
        e = max(e, np.abs(t_ref - t_c).max(), np.abs(q_ref - q_c).max(), np.abs(t_ref - t_n).max(), np.abs(q_ref - q_n).max())
        print(f"J={J:2d} F={F}: max |oracle - reference| (f64) = {e:.3e}")
        worst = max(worst, e)
    assert worst < 1e-12, worst
    # timing equivalence of the NumPy baseline restatement (fp32 inputs, like the survey probe)
    for F in (1000, 100_000):
        rot, gpos, off, par = syn.fk_workload(F)
        best = {}
        for nm, f in (("reference", sk.fk), ("numpy_ref", nr.fk)):
            ts = []
            for _ in range(3 if F > 1000 else 20):
                t0 = time.perf_counter()
                f(rot, gpos, off, par)
                ts.append(time.perf_counter() - t0)
            best[nm] = min(ts)
        print(f"fk F={F}: reference {best['reference'] * 1e3:.1f} ms, numpy_ref {best['numpy_ref'] * 1e3:.1f} ms "
              f"(ratio {best['numpy_ref'] / best['reference']:.2f})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="also cross-check oracle/ against the imported reference")
    ap.add_argument("--only", default="", help="comma-separated subset of: elementwise,trig,skeleton,bvh,mirror,ik,time,skeleton_extra,degenerate")
    args = ap.parse_args()
    gens = {"elementwise": gen_elementwise, "trig": gen_trig, "skeleton": gen_skeleton, "bvh": gen_bvh, "mirror": gen_mirror,
            "ik": gen_ik, "time": gen_time, "skeleton_extra": gen_skeleton_extra, "degenerate": gen_degenerate}
    for name, gen in gens.items():
        if not args.only or name in args.only.split(","):
            gen()
    if args.check:
        check_oracle()

