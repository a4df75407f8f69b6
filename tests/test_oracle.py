"""CPU-only: pin oracle/ (C restatement + NumPy restatement) against the golden vectors that
oracle/make_golden.py produced by importing the reference.  f64 path must agree to 1e-12."""
import numpy as np
import pytest

from conftest import assert_close, golden, up64
from oracle import c_oracle as co
from oracle import numpy_ref as nr

TIGHT = 1e-12

EW = {
    "normalize": (lambda i: co.quat_normalize(i["q"]), ["out"]),
    "length": (lambda i: co.quat_length(i["q"]), ["out"]),
    "to_matrix_unit": (lambda i: co.quat_to_matrix(i["q"]), ["out"]),
    "to_matrix_nonunit": (lambda i: co.quat_to_matrix(i["q"]), ["out"]),
    "to_matrix_lit": (lambda i: co.quat_to_matrix(i["q"]), ["out"]),
    "from_matrix": (lambda i: co.quat_from_matrix(i["m"]), ["out"]),
    "mul": (lambda i: co.quat_mul(i["a"], i["b"]), ["out"]),
    "mul_nonunit": (lambda i: co.quat_mul(i["a"], i["b"]), ["out"]),
    "mul_bcast": (lambda i: co.quat_mul(i["a"], i["b"]), ["out"]),
    "mul_vec": (lambda i: co.quat_mul_vec(i["q"], i["v"]), ["out"]),
    "conjugate": (lambda i: co.quat_conjugate(i["q"]), ["out"]),
    "inverse": (lambda i: co.quat_conjugate(i["q"]), ["out"]),
    "dq_from_rt": (lambda i: co.dq_from_rt(i["q"], i["t"]), ["out"]),
    "dq_to_rt": (lambda i: co.dq_to_rt(i["dq"]), ["q", "t"]),
    "dq_from_t": (lambda i: co.dq_from_t(i["t"]), ["out"]),
    "o6d_to_matrix": (lambda i: co.o6d_to_matrix(i["x"]), ["out"]),
    "o6d_to_quat": (lambda i: co.o6d_to_quat(i["x"]), ["out"]),
    "o6d_from_quat": (lambda i: co.o6d_from_quat(i["q"]), ["out"]),
    "o6d_from_matrix": (lambda i: co.o6d_from_matrix(i["m"]), ["out"]),
    "o6d_to_matrix_zero_col": (lambda i: co.o6d_to_matrix(i["x"]), ["out"]),
}

TRIG = {
    "from_angle_axis": (lambda i: co.quat_from_angle_axis(i["angle"], i["axis"]), ["out"]),
    "from_angle_axis_lit": (lambda i: co.quat_from_angle_axis(i["angle"], i["axis"]), ["out"]),
    "from_scaled_angle_axis": (lambda i: co.quat_from_scaled_angle_axis(i["v"]), ["out"]),
    "to_angle_axis": (lambda i: co.quat_to_angle_axis(i["q"]), ["angle", "axis"]),
    "to_scaled_angle_axis": (lambda i: co.quat_to_scaled_angle_axis(i["q"]), ["out"]),
    "from_euler": (lambda i: co.quat_from_euler(i["e"], i["order"]), ["out"]),
    "to_euler": (lambda i: co.quat_to_euler(i["q"], i["order"]), ["out"]),
    "slerp_shortest1": (lambda i: co.quat_slerp(i["q0"], i["q1"], i["t"], True), ["out"]),
    "slerp_shortest0": (lambda i: co.quat_slerp(i["q0"], i["q1"], i["t"], False), ["out"]),
    "slerp_scalar_t": (lambda i: co.quat_slerp(i["q0"], i["q1"], 0.3, True), ["out"]),
}


def _run(table, g, case):
    fn, names = table[case]
    ins = up64(g.get(case, "in"))
    want = g.get(case, "out64")
    got = fn(ins)
    got = got if isinstance(got, tuple) else (got,)
    for n, a in zip(names, got):
        assert_close(a, want[n], TIGHT, f"{case}.{n}")


@pytest.mark.parametrize("case", sorted(EW))
def test_c_oracle_elementwise(case):
    _run(EW, golden("elementwise.npz"), case)


@pytest.mark.parametrize("case", sorted(TRIG))
def test_c_oracle_trig(case):
    g = golden("trig.npz")
    fn, names = TRIG[case]
    ins = up64(g.get(case, "in"))
    want = g.get(case, "out64")
    got = fn(ins)
    got = got if isinstance(got, tuple) else (got,)
    for n, a in zip(names, got):
        if case == "to_euler":  # angles live on a circle: compare modulo 2pi
            d = np.abs(a - want[n])
            d = np.minimum(d, 2 * np.pi - d)
            assert d.max() < 1e-9
        else:
            assert_close(a, want[n], 1e-10, f"{case}.{n}")


def test_all_golden_elementwise_cases_are_covered():
    g = golden("elementwise.npz")
    skipped = {c for c in g.names() if c.startswith("dq_normalize") or c.startswith("dq_is_unit")}
    assert set(g.names()) - skipped == set(EW)
    assert set(golden("trig.npz").names()) == set(TRIG)


def _skel_cases(prefix):
    return golden("skeleton.npz").names(prefix)


@pytest.mark.parametrize("case", _skel_cases("fk_") )
def test_oracles_fk(case):
    g = golden("skeleton.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")
    if case.startswith("fk_from_o6d"):
        pos, rm, q = co.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"], return_quat=True)
        assert_close(q, want["quat"], TIGHT, case)
    else:
        pos, rm = co.fk(i["rot"], i["gpos"], i["off"], i["parents"])
        p2, r2 = nr.fk(i["rot"], i["gpos"], i["off"], i["parents"])
        assert_close(p2, want["pos"], TIGHT, case + " numpy_ref pos")
        assert_close(r2, want["rotmats"], TIGHT, case + " numpy_ref rot")
    assert_close(pos, want["pos"], TIGHT, case + " pos")
    assert_close(rm, want["rotmats"], TIGHT, case + " rotmats")


@pytest.mark.parametrize("case", golden("skeleton_extra.npz").names("fk_off0_"))
def test_oracles_fk_ignore_a_nonzero_root_offset(case):
    """offsets[0] is overwritten by global_pos in the reference (skeleton.py:49), shared or per-frame"""
    g = golden("skeleton_extra.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")
    pos, rm = co.fk(i["rot"], i["gpos"], i["off"], i["parents"])
    assert_close(pos, want["pos"], TIGHT, case + " pos")
    assert_close(rm, want["rotmats"], TIGHT, case + " rotmats")
    assert_close(pos[:, 0], i["gpos"], 0, case + " root")
    p2, r2 = nr.fk(i["rot"], i["gpos"], i["off"], i["parents"])
    assert_close(p2, want["pos"], TIGHT, case + " numpy_ref pos")


@pytest.mark.parametrize("case", _skel_cases("to_root_dq_"))
def test_oracles_to_root_dq(case):
    g = golden("skeleton.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")
    assert_close(co.to_root_dual_quat(i["rot"], i["gpos"], i["parents"], i["off"]), want["dq"], TIGHT, case)
    assert_close(nr.to_root_dual_quat(i["rot"], i["gpos"], i["parents"], i["off"]), want["dq"], TIGHT, case)


@pytest.mark.parametrize("case", _skel_cases("from_root_dq_"))
def test_oracles_from_root_dq(case):
    g = golden("skeleton.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")
    for mod in (co, nr):
        t, q = mod.from_root_dual_quat(i["dq"], i["parents"])
        assert_close(t, want["trans"], TIGHT, case)
        assert_close(q, want["rot"], TIGHT, case)


@pytest.mark.parametrize("case", _skel_cases("from_global_rot_"))
def test_oracle_from_global_rotations(case):
    g = golden("skeleton.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")
    assert_close(co.from_global_rotations(i["gq"], i["parents"]), want["out"], TIGHT, case)


def test_oracle_f32_instantiation_close_to_f64():
    g = golden("skeleton.npz")
    i = g.get("fk_rand_J22_raw", "in")
    want = g.get("fk_rand_J22_raw", "out64")
    pos, rm = co.fk(i["rot"], i["gpos"], i["off"], i["parents"])  # float32 in -> *_f32 entry points
    assert pos.dtype == np.float32
    assert_close(pos, want["pos"], 1e-5, "f32 oracle pos")
    assert_close(rm, want["rotmats"], 1e-5, "f32 oracle rotmats")


def test_reference_literals_fk():
    """ops/tests/test_skeleton.py:253-266 and :326-331 -- the literal expectations themselves."""
    g = golden("skeleton.npz")
    out = g.get("fk_lit_identity", "out64")
    np.testing.assert_allclose(out["pos"], [[[0, 0, 0], [0, 0, 1], [0, 0, 3]], [[1, 1, 1], [1, 1, 2], [1, 1, 4]]], atol=1e-6)
    np.testing.assert_allclose(out["rotmats"], np.tile(np.eye(3), (2, 3, 1, 1)), atol=1e-6)
    out = g.get("fk_lit_rot", "out64")
    np.testing.assert_allclose(
        out["pos"], [[[0, 0, 0], [0, -1, 0], [2, -1, 0]], [[1, 1, 1], [1.707107, 1, 1.707107], [3.12132, 1, 3.12132]]], atol=1e-6
    )


# ---- degenerate 6D records (tests/golden/degenerate.npz): the oracle's chain ortho6d.to_quat -> fk against the reference --
_DEG_NAMES = ["zero_matrix", "zero_first_col", "zero_second_col", "parallel", "anti_parallel", "near_parallel", "tiny", "huge"]


@pytest.mark.parametrize("J", [22, 52])
@pytest.mark.parametrize("eps,kind", [(0.0, "out64"), (1e-12, "out_t64")])
def test_oracle_fused_chain_on_degenerate_ortho6d_records(J, eps, kind):
    """eps = 0: the NumPy reference (zero column -> NaN); eps = 1e-12: the torch twin's F.normalize floor, at float64.
    Exactly (anti-)parallel columns are rounding noise in the reference itself and are skipped (see test_gpu_degenerate.py)."""
    g = golden("degenerate.npz")
    case = f"fk_from_o6d_degenerate_J{J}"
    i, want = up64(g.get(case, "in")), g.get(case, "out64" if kind == "out64" else "out_t64")
    parents = g.get(case, "in")["parents"]
    with np.errstate(all="ignore"):
        pos, rm, q = co.fk_from_ortho6d(i["x"], i["gpos"], i["off"], parents, eps=eps, return_quat=True)
    rot_ok = np.ones(i["x"].shape[:2], bool)
    pos_ok = rot_ok.copy()
    q_ok = rot_ok.copy()
    for name, (f, j) in zip(_DEG_NAMES, g.get(case, "in")["where"]):
        if name in ("parallel", "anti_parallel"):
            d = {int(j)}
            for k in range(int(j) + 1, J):
                if parents[k] in d:
                    d.add(k)
            rot_ok[f, sorted(d)] = False
            pos_ok[f, sorted(d - {int(j)})] = False
            q_ok[f, j] = False
    for got, ref, ok, what in ((pos, want["pos"], pos_ok, "pos"), (rm, want["rotmats"], rot_ok, "rotmats"), (q, want["quat"], q_ok, "quat")):
        ref = np.asarray(ref, np.float64)
        sel = ok.reshape(ok.shape + (1,) * (got.ndim - 2)) & np.ones(got.shape, bool)
        assert (np.isnan(got[sel]) == np.isnan(ref[sel])).all(), what
        fin = sel & ~np.isnan(ref)
        # (the torch twin's to_matrix allocates default-dtype = fp32 results even for float64 inputs, quat_torch.py:318)
        tol = 1e-9 if kind == "out64" else 1e-6
        assert np.abs(got[fin] - ref[fin]).max() <= tol, (what, np.abs(got[fin] - ref[fin]).max())
