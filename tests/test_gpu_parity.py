"""GPU parity: the HIP path (through the public NumPy / torch front doors, i.e. through the C ABI)
against (1) the golden vectors the reference itself produced, (2) the CPU oracle on seeded
random inputs at awkward sizes.  Tolerance: 1e-5 max abs error, fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import assert_close, golden
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

ATOL = 1e-5

import pymotion_amd.ops.skeleton as sk  # noqa: E402
import pymotion_amd.rotations.dual_quat as dq  # noqa: E402
import pymotion_amd.rotations.ortho6d as o6  # noqa: E402
import pymotion_amd.rotations.quat as quat  # noqa: E402


def _torch_mods():
    import torch

    import pymotion_amd.ops.skeleton_torch as skt
    import pymotion_amd.rotations.dual_quat_torch as dqt
    import pymotion_amd.rotations.ortho6d_torch as o6t
    import pymotion_amd.rotations.quat_torch as quatt

    return torch, skt, dqt, o6t, quatt


# case -> (numpy-callable(ins), torch-callable(mods, ins_as_tensors), output names)
EW = {
    "normalize": (lambda i: quat.normalize(i["q"]), lambda m, i: m[4].normalize(i["q"]), ["out"]),
    "length": (lambda i: quat.length(i["q"]), lambda m, i: m[4].length(i["q"]), ["out"]),
    "to_matrix_unit": (lambda i: quat.to_matrix(i["q"]), lambda m, i: m[4].to_matrix(i["q"]), ["out"]),
    "to_matrix_nonunit": (lambda i: quat.to_matrix(i["q"]), lambda m, i: m[4].to_matrix(i["q"]), ["out"]),
    "to_matrix_lit": (lambda i: quat.to_matrix(i["q"]), lambda m, i: m[4].to_matrix(i["q"]), ["out"]),
    "from_matrix": (lambda i: quat.from_matrix(i["m"]), lambda m, i: m[4].from_matrix(i["m"]), ["out"]),
    "mul": (lambda i: quat.mul(i["a"], i["b"]), lambda m, i: m[4].mul(i["a"], i["b"]), ["out"]),
    "mul_nonunit": (lambda i: quat.mul(i["a"], i["b"]), lambda m, i: m[4].mul(i["a"], i["b"]), ["out"]),
    "mul_bcast": (lambda i: quat.mul(i["a"], i["b"]), lambda m, i: m[4].mul(i["a"], i["b"]), ["out"]),
    "mul_vec": (lambda i: quat.mul_vec(i["q"], i["v"]), lambda m, i: m[4].mul_vec(i["q"], i["v"]), ["out"]),
    "conjugate": (lambda i: quat.conjugate(i["q"]), lambda m, i: m[4].conjugate(i["q"]), ["out"]),
    "inverse": (lambda i: quat.inverse(i["q"]), lambda m, i: m[4].inverse(i["q"]), ["out"]),
    "dq_from_rt": (lambda i: dq.from_rotation_translation(i["q"], i["t"]), lambda m, i: m[2].from_rotation_translation(i["q"], i["t"]), ["out"]),
    "dq_to_rt": (lambda i: dq.to_rotation_translation(i["dq"]), lambda m, i: m[2].to_rotation_translation(i["dq"]), ["q", "t"]),
    "dq_from_t": (lambda i: dq.from_translation(i["t"]), lambda m, i: m[2].from_translation(i["t"]), ["out"]),
    "o6d_to_matrix": (lambda i: o6.to_matrix(i["x"]), lambda m, i: m[3].to_matrix(i["x"]), ["out"]),
    "o6d_to_quat": (lambda i: o6.to_quat(i["x"]), lambda m, i: m[3].to_quat(i["x"]), ["out"]),
    "o6d_from_quat": (lambda i: o6.from_quat(i["q"]), lambda m, i: m[3].from_quat(i["q"]), ["out"]),
    "o6d_from_matrix": (lambda i: o6.from_matrix(i["m"]), lambda m, i: m[3].from_matrix(i["m"]), ["out"]),
}

TRIG = {
    "from_angle_axis": (lambda i: quat.from_angle_axis(i["angle"], i["axis"]), lambda m, i: m[4].from_angle_axis(i["angle"], i["axis"]), ["out"]),
    "from_angle_axis_lit": (lambda i: quat.from_angle_axis(i["angle"], i["axis"]), lambda m, i: m[4].from_angle_axis(i["angle"], i["axis"]), ["out"]),
    "from_scaled_angle_axis": (lambda i: quat.from_scaled_angle_axis(i["v"]), lambda m, i: m[4].from_scaled_angle_axis(i["v"]), ["out"]),
    "to_angle_axis": (lambda i: quat.to_angle_axis(i["q"]), lambda m, i: m[4].to_angle_axis(i["q"]), ["angle", "axis"]),
    "to_scaled_angle_axis": (lambda i: quat.to_scaled_angle_axis(i["q"]), lambda m, i: m[4].to_scaled_angle_axis(i["q"]), ["out"]),
    "slerp_shortest1": (lambda i: quat.slerp(i["q0"], i["q1"], i["t"], True), lambda m, i: m[4].slerp(i["q0"], i["q1"], i["t"], True), ["out"]),
    "slerp_shortest0": (lambda i: quat.slerp(i["q0"], i["q1"], i["t"], False), lambda m, i: m[4].slerp(i["q0"], i["q1"], i["t"], False), ["out"]),
    "slerp_scalar_t": (lambda i: quat.slerp(i["q0"], i["q1"], 0.3), lambda m, i: m[4].slerp(i["q0"], i["q1"], 0.3), ["out"]),
}

# (round 1 allowed 2e-3 here: sqrt(1 - w^2) near |w| = 1 in fp32, like the reference's own fp32 twin -- its tests use
# low_atol = 1e-3, test_quat.py:29.  The difference is formed in float64 now and the 1e-5 bar holds.)
LOOSE = {}


def _tup(x):
    return x if isinstance(x, tuple) else (x,)


# golden element 1 of the ortho6d cases has near-parallel columns (make_golden.py): Gram-Schmidt
# amplifies fp32 input rounding by ~1e3 there, for the reference's own fp32 twin as well
ILL = {"o6d_to_matrix": [1], "o6d_to_quat": [1]}


def _check(case, names, got, g, atol=ATOL):
    want = g.get(case, "out64")
    for n, a in zip(names, _tup(got)):
        a, w = np.asarray(a, dtype=np.float64), want[n]
        if case in ILL:
            keep = np.ones(len(w), bool)
            keep[ILL[case]] = False
            ref32 = g.get(case, "out_t")[n]  # the reference's fp32 twin on the same inputs
            for k in ILL[case]:
                assert np.abs(a[k] - w[k]).max() <= max(1e-3, 4 * np.abs(ref32[k] - w[k]).max())
            a, w = a[keep], w[keep]
        assert_close(a, w, atol, f"{case}.{n}")


@pytest.mark.parametrize("case", sorted(EW))
def test_elementwise_numpy_vs_reference_golden(case):
    g = golden("elementwise.npz")
    fn, _, names = EW[case]
    _check(case, names, fn(g.get(case, "in")), g)


@pytest.mark.parametrize("case", sorted(EW))
def test_elementwise_torch_vs_reference_golden(case):
    m = _torch_mods()
    torch = m[0]
    g = golden("elementwise.npz")
    _, fn, names = EW[case]
    ins = {k: torch.from_numpy(v).cuda() for k, v in g.get(case, "in").items()}
    got = _tup(fn(m, ins))
    assert all(t.is_cuda for t in got), "result must stay on the input's device"
    _check(case, names, tuple(t.cpu().numpy() for t in got), g)
    # CPU tensors: copied over and back, result on the CPU
    got_cpu = _tup(fn(m, {k: v.cpu() for k, v in ins.items()}))
    assert all(not t.is_cuda for t in got_cpu)
    _check(case, names, tuple(t.numpy() for t in got_cpu), g)


@pytest.mark.parametrize("case", sorted(TRIG))
def test_trig_vs_reference_golden(case):
    g = golden("trig.npz")
    fn, fn_t, names = TRIG[case]
    atol = LOOSE.get(case, ATOL)
    _check(case, names, fn(g.get(case, "in")), g, atol)
    m = _torch_mods()
    ins = {k: m[0].from_numpy(v).cuda() for k, v in g.get(case, "in").items()}
    _check(case, names, tuple(t.cpu().numpy() for t in _tup(fn_t(m, ins))), g, atol)


def test_sin_cos_of_the_trig_ops_over_a_wide_angle_range():
    """from_angle_axis uses the library's own sin/cos (one double-precision FMA for the range reduction +
    minimax polynomials): libm-grade over every range a caller can produce, libm itself beyond 1e8 rad."""
    rng = np.random.default_rng(12)
    ax = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (4096, 1))
    for scale, tol in ((3.2, 2e-7), (100.0, 2e-7), (2e4, 2e-7), (5e7, 2e-7), (3e9, 2e-7)):
        ang = (rng.uniform(-scale, scale, (4096, 1))).astype(np.float32)
        got = quat.from_angle_axis(ang, ax)
        h = ang.astype(np.float64)[:, 0] / 2
        assert_close(got[:, 0], np.cos(h), tol, f"cos up to {scale}")
        assert_close(got[:, 3], np.sin(h), tol, f"sin up to {scale}")
    exact = quat.from_angle_axis(np.array([[0.0], [np.pi], [-np.pi], [2 * np.pi]], np.float32), ax[:4])
    assert_close(exact[:, 0], [1.0, np.cos(np.float32(np.pi) / 2), np.cos(np.float32(np.pi) / 2), -1.0], 2e-7)
    bad = quat.from_angle_axis(np.array([[np.inf], [np.nan]], np.float32), ax[:2])
    assert np.isnan(bad[:, 0]).all()


def _order_strings(codes):
    return np.array(list("xyz"))[codes]


def test_euler_vs_reference_golden():
    g = golden("trig.npz")
    i = g.get("from_euler", "in")
    order = _order_strings(i["order"])
    _check("from_euler", ["out"], quat.from_euler(i["e"], order), g)
    i = g.get("to_euler", "in")
    got = quat.to_euler(i["q"], _order_strings(i["order"]))
    want = g.get("to_euler", "out64")["out"]
    d = np.abs(got - want)
    d = np.minimum(d, 2 * np.pi - d)  # angles live on a circle
    assert d.max() < 1e-5  # (round 1: 5e-4 -- the atan2 operands are differences of components, formed in float64 now)
    # a single order triple for the whole batch takes the per-call path
    e = i["q"][:7]
    o1 = np.tile(np.array(["z", "x", "y"]), (7, 1))
    a = quat.to_euler(e, o1)
    b = co.quat_to_euler(e.astype(np.float64), o1)
    d = np.abs(a - b)
    assert np.minimum(d, 2 * np.pi - d).max() < 1e-5


@pytest.mark.parametrize("F,J", [(1000, 22), (257, 52), (5000, 7), (3, 130)])
def test_euler_with_one_order_per_joint_tiled_over_the_frames(F, J):
    """the layout of every BVH clip: the order array repeats with period J and travels as a [J, 3] table; results must
    equal the order-per-element path and the oracle, whatever the tile boundaries do to e % J"""
    rng = np.random.default_rng(F + J)
    per_joint = np.array(list("xyz"))[rng.permuted(np.tile(np.arange(3), (J, 1)), axis=1)]
    order = np.tile(per_joint, (F, 1, 1))
    e = rng.uniform(-3, 3, (F, J, 3)).astype(np.float32)
    q = quat.from_euler(e, order)
    assert_close(q, co.quat_from_euler(e.astype(np.float64), order), ATOL, "from_euler, per-joint table")
    odd = order.copy()
    odd[0, 0] = odd[0, 0][::-1]  # breaks the period -> order-per-element path; every other element must agree bit for bit
    q1 = quat.from_euler(e, odd)
    np.testing.assert_array_equal(q1.reshape(-1, 4)[1:], q.reshape(-1, 4)[1:])
    a = quat.to_euler(q.astype(np.float32), order)
    b = quat.to_euler(q.astype(np.float32), odd)
    np.testing.assert_array_equal(a.reshape(-1, 3)[1:], b.reshape(-1, 3)[1:])
    back = co.quat_from_euler(a, order)
    assert np.minimum(np.abs(back - q).max(-1), np.abs(back + q).max(-1)).max() < 2e-5


def test_from_euler_every_order_triple_and_both_angle_ranges():
    """from_euler takes a closed form for three distinct axes and the general Hamilton products for a repeated axis (a branch per
    wave), and its sin / cos an fp32 reduction below 4096 rad and the float64 one above: all 27 triples, alone and mixed inside
    one wave, on small and large angles, against the oracle (quat.py:43-82 is defined for any triple)"""
    rng = np.random.default_rng(27)
    axes = np.array(list("xyz"))
    triples = np.array([[a, b, c] for a in range(3) for b in range(3) for c in range(3)])
    n = 1000
    for scale in (3.2, 3000.0, 5000.0, 2.0e5):
        e = rng.uniform(-scale, scale, (n, 3)).astype(np.float32)
        e[::7, 1] = 0.0
        for t in triples:                                   # one order for the whole batch
            order = np.tile(axes[t], (n, 1))
            got = quat.from_euler(e, order)
            assert_close(got, co.quat_from_euler(e.astype(np.float64), order), 2e-6 if scale < 1e4 else ATOL, f"order {''.join(axes[t])} scale {scale}")
        order = axes[triples[rng.integers(0, 27, n)]]      # an order per element: distinct and repeated axes in the same wave
        assert_close(quat.from_euler(e, order), co.quat_from_euler(e.astype(np.float64), order), 2e-6 if scale < 1e4 else ATOL, f"mixed orders, scale {scale}")


def test_to_euler_on_quadrant_boundaries_and_identity():
    """the library's own atan2 must keep np.arctan2's conventions where they matter: axis-aligned rotations
    (operands exactly 0), the identity (atan2(0, 0) family) and every quadrant"""
    ang = np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, 0.3, -2.8, 1.0e-7, -1.0e-7], np.float64)
    grid = np.stack(np.meshgrid(ang, ang, ang, indexing="ij"), -1).reshape(-1, 3)
    for order in (["z", "x", "y"], ["x", "y", "z"], ["y", "z", "x"]):
        o = np.tile(np.array(order), (len(grid), 1))
        q = co.quat_from_euler(grid, o)
        got = quat.to_euler(q.astype(np.float32), o)
        want = co.quat_to_euler(q.astype(np.float32).astype(np.float64), o)
        d = np.abs(got - want)
        d = np.minimum(d, 2 * np.pi - d)
        # gimbal-locked poses (middle angle +-pi/2) split the remaining rotation between the outer angles
        # arbitrarily: judge those through the rotation they encode
        lock = np.isclose(np.abs(np.sin(grid[:, 1])), 1.0, atol=1e-6)
        assert d[~lock].max() < 2e-5, (order, d[~lock].max())
        back = co.quat_from_euler(got.astype(np.float64), o)
        err = np.minimum(np.abs(back - q).max(-1), np.abs(back + q).max(-1))
        assert err.max() < 2e-5, (order, err.max())
        assert np.isfinite(got).all() and (got >= 0).all() and (got <= 2 * np.pi + 1e-6).all()


def test_o6d_zero_column_matches_each_front_door():
    g = golden("elementwise.npz")
    x = g.get("o6d_to_matrix_zero_col", "in")["x"]
    got = o6.to_matrix(x)
    want = g.get("o6d_to_matrix_zero_col", "out_np")["out"]
    assert np.isnan(got[0]).any() and (np.isnan(got) == np.isnan(want)).all()  # NumPy reference: NaN
    assert_close(got[2:], g.get("o6d_to_matrix_zero_col", "out64")["out"][2:], ATOL)  # [1] is the ill-conditioned one
    m = _torch_mods()
    got_t = m[3].to_matrix(m[0].from_numpy(x).cuda()).cpu().numpy()
    want_t = g.get("o6d_to_matrix_zero_col", "out_t")["out"]
    assert np.isfinite(got_t).all()  # torch twin: F.normalize eps -> zeros, no NaN
    assert_close(got_t[[0, 2, 3]], want_t[[0, 2, 3]], ATOL)


def test_output_dtypes_follow_the_reference():
    g = golden("elementwise.npz")
    q32 = g.get("mul", "in")["a"]
    assert quat.to_matrix(q32).dtype == np.float64  # quat.py:306
    assert quat.mul(q32, q32).dtype == np.float32
    assert quat.mul(q32.astype(np.float64), q32).dtype == np.float64
    assert quat.normalize(q32).dtype == np.float32
    assert dq.from_rotation_translation(q32, q32[:, :3]).dtype == np.float64  # dual_quat.py:32
    m = _torch_mods()
    torch = m[0]
    t = torch.from_numpy(q32).cuda()
    assert m[4].to_matrix(t.double()).dtype == torch.get_default_dtype()  # quat_torch.py:318
    assert m[4].mul(t.double(), t).dtype == torch.float64
    assert m[2].from_rotation_translation(t, t[:, :3]).dtype == torch.float32


@pytest.mark.parametrize("n", [1, 63, 64, 255, 256, 257, 1000, 100_003])
def test_elementwise_vs_oracle_sizes(n):
    rng = np.random.default_rng(n)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q2 = rng.standard_normal((n, 4)).astype(np.float32)
    v = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    x = rng.standard_normal((n, 3, 2)).astype(np.float32)
    qn = q / np.linalg.norm(q, axis=-1, keepdims=True)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    assert_close(quat.to_matrix(q), co.quat_to_matrix(f64(q)), 2e-5, "to_matrix (entries up to ~40)")
    assert_close(quat.to_matrix(qn), co.quat_to_matrix(f64(qn)), ATOL, "to_matrix unit")
    assert_close(quat.normalize(q), co.quat_normalize(f64(q)), ATOL, "normalize")
    assert_close(quat.mul(qn, q2), co.quat_mul(f64(qn), f64(q2)), ATOL, "mul")
    assert_close(quat.mul_vec(qn, v), co.quat_mul_vec(f64(qn), f64(v)), ATOL, "mul_vec")
    m = co.quat_to_matrix(f64(qn)).astype(np.float32)
    assert_close(quat.from_matrix(m), co.quat_from_matrix(f64(m)), ATOL, "from_matrix")
    assert_close(o6.to_matrix(x), co.o6d_to_matrix(f64(x)), 5e-5, "o6d.to_matrix (ill-conditioned rows allowed)")
    d = dq.from_rotation_translation(qn, v)
    assert_close(d, co.dq_from_rt(f64(qn), f64(v)), ATOL, "dq.from_rt")
    r, t = dq.to_rotation_translation(d.astype(np.float32))
    assert_close(r, qn, ATOL, "dq round trip q")
    assert_close(t, v, ATOL, "dq round trip t")


# ---- skeleton ops ---------------------------------------------------------------------------------

def _skel(prefix):
    return golden("skeleton.npz").names(prefix)


@pytest.mark.parametrize("case", [c for c in _skel("fk_") if not c.startswith("fk_from_o6d")])
def test_fk_numpy_and_torch_vs_reference_golden(case):
    g = golden("skeleton.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")
    pos, rm = sk.fk(i["rot"], i["gpos"], i["off"], i["parents"])
    assert pos.dtype == np.float64 and pos.flags.c_contiguous  # skeleton.py:44 (f64), contiguous by design
    assert_close(pos, want["pos"], ATOL, case + " pos")
    assert_close(rm, want["rotmats"], ATOL, case + " rotmats")
    torch, skt = _torch_mods()[:2]
    tp, tr = skt.fk(*[torch.from_numpy(i[k]).cuda() for k in ("rot", "gpos", "off")], torch.from_numpy(i["parents"]))
    assert tp.dtype == torch.float32 and tp.is_cuda
    assert_close(tp.cpu().numpy(), want["pos"], ATOL, case + " torch pos")
    assert_close(tr.cpu().numpy(), want["rotmats"], ATOL, case + " torch rotmats")


@pytest.mark.parametrize("case", golden("skeleton_extra.npz").names("fk_off0_"))
def test_fk_ignores_a_nonzero_root_offset_like_the_reference(case):
    """regression (found by tests/test_gpu_fuzz.py): the three-lane walk used to add offsets[0] to the root"""
    g = golden("skeleton_extra.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")
    pos, rm = sk.fk(i["rot"], i["gpos"], i["off"], i["parents"])
    assert_close(pos, want["pos"], ATOL, case + " pos")
    assert_close(rm, want["rotmats"], ATOL, case + " rotmats")
    assert_close(pos[:, 0], i["gpos"], 0, case + " root position is global_pos, bit for bit")
    torch, skt = _torch_mods()[:2]
    tp, tr = skt.fk(*[torch.from_numpy(i[k]).cuda() for k in ("rot", "gpos", "off")], torch.from_numpy(i["parents"]))
    assert_close(tp.cpu().numpy(), want["pos"], ATOL, case + " torch pos")


def test_fk_from_ortho6d_vs_reference_golden():
    g = golden("skeleton.npz")
    case = "fk_from_o6d_J52"
    i, want = g.get(case, "in"), g.get(case, "out64")
    pos, rm, q = sk.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"], return_quat=True)
    assert_close(pos, want["pos"], ATOL, "pos")
    assert_close(rm, want["rotmats"], ATOL, "rotmats")
    assert_close(q, want["quat"], ATOL, "quat")
    # without the quaternion output the kernel takes the Gram-Schmidt matrix as the local rotation directly
    # (no matrix -> quaternion -> matrix trip): same transforms up to fp32 rounding, same parity vs the reference
    pos2, rm2 = sk.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"])
    assert_close(pos2, want["pos"], ATOL, "pos (no quat output)")
    assert_close(rm2, want["rotmats"], ATOL, "rotmats (no quat output)")
    # and the fused kernel equals the two-launch chain on the GPU
    p3, r3 = sk.fk(o6.to_quat(i["x"]), i["gpos"], i["off"], i["parents"])
    assert_close(p3, pos, 1e-6, "fused vs chained")


@pytest.mark.parametrize("case", _skel("to_root_dq_"))
def test_to_root_dq_vs_reference_golden(case):
    g = golden("skeleton.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")
    d = sk.to_root_dual_quat(i["rot"], i["gpos"], i["parents"], i["off"])
    assert d.dtype == np.float64
    assert_close(d, want["dq"], ATOL, case)
    torch, skt = _torch_mods()[:2]
    dt = skt.to_root_dual_quat(torch.from_numpy(i["rot"]).cuda(), torch.from_numpy(i["gpos"]).cuda(),
                               torch.from_numpy(i["parents"]), torch.from_numpy(i["off"]).cuda())
    assert_close(dt.cpu().numpy(), want["dq"], ATOL, case + " torch")


@pytest.mark.parametrize("case", _skel("from_root_dq_"))
def test_from_root_dq_vs_reference_golden(case):
    g = golden("skeleton.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")
    t, q = sk.from_root_dual_quat(i["dq"], i["parents"])  # (translations, rotations): skeleton.py:204
    assert_close(t, want["trans"], ATOL, case + " trans")
    assert_close(q, want["rot"], ATOL, case + " rot")


@pytest.mark.parametrize("case", _skel("from_global_rot_"))
def test_from_global_rotations_vs_reference_golden(case):
    g = golden("skeleton.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")
    assert_close(sk.from_global_rotations(i["gq"], i["parents"]), want["out"], ATOL, case)


def test_reference_style_dq_round_trip():
    """Reads like ops/tests/test_skeleton.py:test_dq: encode, decode, recover the pose."""
    g = golden("skeleton.npz")
    i = g.get("to_root_dq_rand_J22", "in")
    d = sk.to_root_dual_quat(i["rot"], i["gpos"], i["parents"], i["off"])
    t, q = sk.from_root_dual_quat(d, i["parents"])
    assert_close(q, i["rot"], ATOL, "rotations")
    assert_close(t[:, 1:, :], np.tile(i["off"][1:], (t.shape[0], 1, 1)), ATOL, "offsets")
    assert_close(t[:, 0, :], i["gpos"], ATOL, "global position")
    # multi-dim leading shape: joint axis is -2 (fix of the reference's shape[1])
    rot6 = np.tile(i["rot"][:4].reshape(2, 2, 22, 4), (3, 1, 1, 1, 1))
    gp6 = np.tile(i["gpos"][:4].reshape(2, 2, 3), (3, 1, 1, 1))
    d6 = sk.to_root_dual_quat(rot6, gp6, i["parents"], i["off"])
    assert d6.shape == (3, 2, 2, 22, 8)
    assert_close(d6[1].reshape(4, 22, 8), d[:4], 1e-7, "leading dims are just batch")


@pytest.mark.parametrize("F,J", [(1, 1), (1, 22), (19, 22), (20, 22), (21, 22), (1000, 22), (100_003, 22), (777, 52),
                                 (50, 3), (333, 128), (7, 300), (9, 512),
                                 # joint counts that are multiples of 8 / 16: the fk image is padded per frame there
                                 (77, 8), (130, 16), (61, 24), (2049, 24), (45, 32), (33, 48), (21, 64), (19, 80)])
def test_fk_and_dq_vs_oracle_sizes(F, J):
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(F * 1000 + J)
    parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if parents is None:
        parents = syn.random_parents(J, rng)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.3 if J <= 52 else 0.05)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    rot_z = rot.copy()
    rot_z[::5, ::3] = 0  # zero quaternions -> identity local rotation (quat.py:423), on every kernel shape
    pz, rz = sk.fk(rot_z, gpos, off, parents)
    pz_o, rz_o = co.fk(f64(rot_z), f64(gpos), f64(off), parents)
    assert_close(pz, pz_o, ATOL, "fk pos with zero quats")
    assert_close(rz, rz_o, ATOL, "fk rotmats with zero quats")
    pos, rm = sk.fk(rot, gpos, off, parents)
    p_o, r_o = co.fk(f64(rot), f64(gpos), f64(off), parents)
    assert_close(pos, p_o, ATOL, "fk pos")
    assert_close(rm, r_o, ATOL, "fk rotmats")
    rn = rot / np.linalg.norm(rot, axis=-1, keepdims=True)
    d = sk.to_root_dual_quat(rn, gpos, parents, off)
    d_o = co.to_root_dual_quat(f64(rn), f64(gpos), parents, f64(off))
    assert_close(d, d_o, ATOL, "to_root_dq")
    t, q = sk.from_root_dual_quat(d_o.astype(np.float32), parents)
    t_o, q_o = co.from_root_dual_quat(f64(d_o.astype(np.float32)), parents)
    assert_close(t, t_o, ATOL, "from_root_dq trans")
    assert_close(q, q_o, ATOL, "from_root_dq rot")


@pytest.mark.parametrize("J", [22, 52])
def test_fk_quaternions_of_small_and_large_norm(J):
    """fk normalises with q / (|q| + 1e-8) (skeleton.py:45): the eps is invisible in fp32 for |q| >= ~0.2 and a visible
    shrink of the matrix below (L - I scales by (1 - 1e-8 / |q|)^2: 4e-6 at |q| = 0.01).  The kernels' residual-scaled
    conversion (fk.hip: PREC_RESID) has to reproduce that on both sides of the point where |q| + 1e-8f == |q| in fp32."""
    from pymotion_amd import synthetic as syn

    parents = syn.PARENTS_22 if J == 22 else syn.PARENTS_52
    rng = np.random.default_rng(1000 + J)
    F = 3001
    rot = rng.standard_normal((F, J, 4))
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    # norms log-uniform in [1e-3, 1e3], with the band [0.01, 0.2] (eps enters the fp32 reciprocal) over-represented
    norms = np.where(rng.random((F, J, 1)) < 0.5, 10.0 ** rng.uniform(-2, np.log10(0.2), (F, J, 1)), 10.0 ** rng.uniform(-3, 3, (F, J, 1)))
    rot = (rot * norms).astype(np.float32)
    gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.3 if J == 22 else 0.15)
    pos, rm = sk.fk(rot, gpos, off, parents)
    p_o, r_o = co.fk(rot.astype(np.float64), gpos.astype(np.float64), off.astype(np.float64), parents)
    assert_close(rm, r_o, 3e-6, "rotmats")
    assert_close(pos, p_o, 4e-6, "pos")


@pytest.mark.parametrize("nt", ["0", "1", "2", "3", "7"])
@pytest.mark.parametrize("F,J", [(4 * 7 + 1, 52), (4 * 6, 28), (3, 64), (4 * 13 + 2, 33), (4 * 5 + 3, 65), (9, 128), (6, 96)])
def test_fk_pipelined_tiles_ragged_groups(monkeypatch, nt, F, J):
    """28 <= J <= 128 runs fk_pipe_kernel (4 records per lane up to 64 joints, 8 beyond): `nt` tiles per workgroup (0 = the
    one-tile kernel).  Odd tile counts, a partial last tile and a partial last group must all come out identical.
    Production picks nt from the batch size; the tuning build of the library (-DPM_TUNING) takes it from PM_FK_NT."""
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    monkeypatch.setenv("PM_FK_NT", nt)
    monkeypatch.setenv("PM_FK_WIDE", "0")  # (random trees of more than 92 joints take fkwide.hip's kernel otherwise: tests/test_gpu_wide.py)
    with _lib.variant("tuning"):
        _fk_pipelined_body(F, J)


def _fk_pipelined_body(F, J):
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(F * 100 + J)
    parents = syn.PARENTS_52 if J == 52 else syn.random_parents(J, rng)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.15)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    pos, rm = sk.fk(rot, gpos, off, parents)
    p_o, r_o = co.fk(f64(rot), f64(gpos), f64(off), parents)
    assert_close(pos, p_o, ATOL, "fk pos")
    assert_close(rm, r_o, ATOL, "fk rotmats")
    x = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
    q_o = co.o6d_to_quat(f64(x))
    p_o, r_o = co.fk(q_o, f64(gpos), f64(off), parents)
    for want_q in (False, True):
        out = sk.fk_from_ortho6d(x, gpos, off, parents, return_quat=want_q)
        assert_close(out[0], p_o, 2e-5, "fused pos")
        assert_close(out[1], r_o, 2e-5, "fused rotmats")
        if want_q:
            assert_close(out[2], q_o, ATOL, "fused quats")
    # per-frame offsets ride in the same pipelined kernel (their records are prefetched like the rotations)
    offf = rng.uniform(-0.2, 0.2, (F, J, 3)).astype(np.float32)
    pos, rm = sk.fk(rot, gpos, offf, parents)
    p_o, r_o = co.fk(f64(rot), f64(gpos), f64(offf), parents)
    assert_close(pos, p_o, ATOL, "fk pos, per-frame offsets")
    assert_close(rm, r_o, ATOL, "fk rotmats, per-frame offsets")
    out = sk.fk_from_ortho6d(x, gpos, offf, parents, return_quat=True)
    p_o, r_o = co.fk(q_o, f64(gpos), f64(offf), parents)
    assert_close(out[0], p_o, 2e-5, "fused pos, per-frame offsets")
    assert_close(out[1], r_o, 2e-5, "fused rotmats, per-frame offsets")


def test_fk_per_frame_offsets_and_unaligned_views():
    from pymotion_amd import synthetic as syn

    torch, skt = _torch_mods()[:2]
    rot, gpos, off, parents = syn.fk_workload(501, seed=3)
    offs = np.tile(off, (501, 1, 1)) * np.linspace(0.5, 1.5, 501, dtype=np.float32)[:, None, None]
    pos, rm = sk.fk(rot, gpos, offs, parents)
    p_o, r_o = co.fk(rot.astype(np.float64), gpos.astype(np.float64), offs.astype(np.float64), parents)
    assert_close(pos, p_o, ATOL, "per-frame offsets pos")
    # a torch view whose data_ptr is only 4-byte aligned takes the scalar path of the kernels
    big = torch.from_numpy(np.concatenate([np.zeros(1, np.float32), rot.ravel()])).cuda()
    rot_view = big[1:].view(501, 22, 4)
    assert rot_view.data_ptr() % 16 != 0
    tp, tr = skt.fk(rot_view, torch.from_numpy(gpos).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(parents))
    p1, r1 = co.fk(rot.astype(np.float64), gpos.astype(np.float64), off.astype(np.float64), parents)
    assert_close(tp.cpu().numpy(), p1, ATOL, "unaligned pos")
    assert_close(tr.cpu().numpy(), r1, ATOL, "unaligned rotmats")


def test_device_resident_parents_and_offsets_are_rechecked_when_they_change():
    """parents / offsets living on the GPU cost a host sync to inspect; the verdict is remembered per tensor
    VERSION, so an in-place edit must be seen"""
    torch, skt = _torch_mods()[:2]
    from pymotion_amd import synthetic as syn

    rot, root, off, par = syn.fk_workload(33, seed=4, normalized=True)
    tr, tg, to = (torch.from_numpy(a).cuda() for a in (rot, root, off))
    tp = torch.from_numpy(par.astype(np.int64)).cuda()
    for _ in range(3):
        pos, _ = skt.fk(tr, tg, to, tp)
        assert_close(pos.cpu().numpy(), co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), par)[0], ATOL)
        skt.to_root_dual_quat(tr, tg, tp, to)
    tp[5] = 2  # re-parent joint 5 in place (was 0)
    par2 = par.copy()
    par2[5] = 2
    pos, _ = skt.fk(tr, tg, to, tp)
    assert_close(pos.cpu().numpy(), co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), par2)[0], ATOL)
    to[0, 1] = 0.5  # offsets[0] != 0 now: to_root_dual_quat must assert like the reference (skeleton_torch.py:242)
    with pytest.raises(AssertionError):
        skt.to_root_dual_quat(tr, tg, tp, to)
    tp[3] = 7  # not topological any more
    with pytest.raises(ValueError):
        skt.fk(tr, tg, to, tp)


def test_a_new_parents_tensor_at_a_recycled_address_is_a_new_skeleton():
    """Regression (round-1 verdict, ADVICE): torch's caching allocator hands the freed block of one `parents` tensor to
    the next tensor of the same size, with version 0 again -- an (address, version) key then serves the PREVIOUS
    skeleton's topology.  The memo is keyed on the tensor object; every new tensor is re-read like the reference does
    (skeleton_torch.py:56)."""
    torch, skt = _torch_mods()[:2]
    from pymotion_amd import synthetic as syn

    J, F = 22, 65
    rng = np.random.default_rng(5)
    rot, root, off, _ = syn.fk_workload(F, seed=6, normalized=True)
    tr, tg, to = (torch.from_numpy(a).cuda() for a in (rot, root, off))
    chain = np.maximum(np.arange(J) - 1, 0).astype(np.int64)
    topologies = [syn.PARENTS_22.astype(np.int64), chain] + [syn.random_parents(J, rng).astype(np.int64) for _ in range(6)]
    reused, prev_ptr = 0, None
    for k, par in enumerate(topologies * 3):
        tp = torch.from_numpy(par).cuda()          # fresh tensor, version 0
        reused += int(tp.data_ptr() == prev_ptr)
        pos, rm = skt.fk(tr, tg, to, tp)
        p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), par.astype(np.int32))
        assert_close(pos.cpu().numpy(), p_o, ATOL, f"topology {k}")
        assert_close(rm.cpu().numpy(), r_o, ATOL, f"topology {k}")
        dq = skt.to_root_dual_quat(tr, tg, tp, to)
        assert_close(dq.cpu().numpy(), co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64),
                                                           par.astype(np.int32), off.astype(np.float64)), ATOL, f"dq {k}")
        prev_ptr = tp.data_ptr()
        del tp                                      # the block goes back to the allocator ...
    assert reused > 0, "the allocator never recycled the address: the regression was not exercised"
    # same hazard for the root-offset assertion (skeleton_torch.py:242): a NEW offsets tensor with offsets[0] != 0 that
    # lands on the address of one already checked must still trip it
    tp = torch.from_numpy(topologies[0]).cuda()
    for _ in range(4):
        good = torch.from_numpy(off).cuda()
        skt.to_root_dual_quat(tr, tg, tp, good)
        ptr = good.data_ptr()
        del good
        bad_np = off.copy()
        bad_np[0, 2] = 0.25
        bad = torch.from_numpy(bad_np).cuda()
        if bad.data_ptr() == ptr:
            reused += 100
        with pytest.raises(AssertionError):
            skt.to_root_dual_quat(tr, tg, tp, bad)
        del bad
    assert reused >= 100, "offsets address was never recycled"


def test_numpy_door_large_inputs_take_the_threaded_staging_path():
    """>= 4 MB arrays that need a cast or a gather (float64, strided, Fortran order) are converted by several
    threads into reused staging memory; the results must be bit-identical to the plain fp32 contiguous call"""
    from pymotion_amd import synthetic as syn

    rot, root, off, par = syn.fk_workload(60_000, seed=9)
    p0, r0 = sk.fk(rot, root, off, par)
    wide = np.zeros(rot.shape[:-1] + (8,), np.float64)
    wide[..., ::2] = rot
    for variant in (rot.astype(np.float64), wide[..., ::2], np.asfortranarray(rot)):
        assert not (variant.dtype == np.float32 and variant.flags.c_contiguous)
        p1, r1 = sk.fk(variant, root.astype(np.float64), off, par)
        np.testing.assert_array_equal(p1, p0)
        np.testing.assert_array_equal(r1, r0)
    for _ in range(3):  # staging buffers are recycled between calls: results stay independent arrays
        p2, r2 = sk.fk(rot, root, off, par)
        assert p2 is not p0 and not np.shares_memory(p2, p0)
        np.testing.assert_array_equal(p2, p0)


def test_bad_topology_raises_value_error():
    rot = np.zeros((2, 3, 4), np.float32)
    with pytest.raises(ValueError, match="topological"):
        sk.fk(rot, np.zeros((2, 3), np.float32), np.zeros((3, 3), np.float32), np.array([0, 2, 1]))


def test_empty_batch():
    pos, rm = sk.fk(np.zeros((0, 22, 4), np.float32), np.zeros((0, 3), np.float32), np.zeros((22, 3), np.float32),
                    np.maximum(np.arange(22) - 1, 0))
    assert pos.shape == (0, 22, 3) and rm.shape == (0, 22, 3, 3)
    assert quat.mul(np.zeros((0, 4)), np.zeros((0, 4))).shape == (0, 4)


def test_dual_quat_normalize_and_is_unit_vs_reference_golden():
    g = golden("elementwise.npz")
    m = _torch_mods()
    torch, dqt = m[0], m[2]
    for case in ("dq_normalize_scaled", "dq_normalize_skew"):
        x = g.get(case, "in")["dq"]
        want = g.get(case, "out64")["out"]
        assert_close(dq.normalize(x), want, ATOL, case)
        assert_close(dqt.normalize(torch.from_numpy(x).cuda()).cpu().numpy(), want, ATOL, case + " torch")
    for nm in ("unit", "scaled", "skew", "zero"):
        case = f"dq_is_unit_{nm}"
        x = g.get(case, "in")["dq"]
        want = bool(g.get(case, "out_np")["out"])
        assert dq.is_unit(x) is want, case
        assert dqt.is_unit(torch.from_numpy(x).cuda()) is want, case
    # a single bad element flips the verdict for the whole batch (global .all(), dual_quat.py:132-136)
    x = g.get("dq_is_unit_unit", "in")["dq"].copy()
    assert dq.is_unit(x)
    x[17, 4:] += 0.5 * x[17, :4]
    assert not dq.is_unit(x)


def _mirror_tie_elements(rot, off, parents, mapping):
    """[F, J] mask of mirrored local rotations whose SIGN is a coin toss.  The sign is decided by from_matrix's four-way
    branch (quat.py:110-156) on the world rotations of the joint and of its parent; in terms of the world quaternion g the
    predicates are xx + yy > ww + zz, |x| > |y|, |w| < |z| (mirror.hip), so an element is a tie when two of those
    quantities agree to within the fp32 chain's own error."""
    _, rm_w = co.fk(np.asarray(rot, np.float64), np.zeros((rot.shape[0], 3)), np.asarray(off, np.float64), parents)
    sq = co.quat_from_matrix(rm_w) ** 2
    neg = sq[..., 1] + sq[..., 2] > sq[..., 0] + sq[..., 3]
    margin = np.minimum(np.abs(sq[..., 1] + sq[..., 2] - sq[..., 0] - sq[..., 3]),
                        np.where(neg, np.abs(sq[..., 1] - sq[..., 2]), np.abs(sq[..., 0] - sq[..., 3])))
    tie = margin < 4e-6
    mp = np.arange(len(parents)) if mapping is None else np.asarray(mapping)
    par = np.asarray(parents).copy()
    par[0] = 0
    return tie[:, mp] | tie[:, mp[par]]


@pytest.mark.parametrize("mode", ["all", "symmetry"])
@pytest.mark.parametrize("axis", ["X", "Y", "Z"])
def test_mirror_vs_reference_golden(mode, axis):
    g = golden("mirror.npz")
    i = g.get("inputs", "in")
    want = g.get(f"mirror_{mode}_{axis}", "out64")
    root0 = i["root"].copy()
    r, gt, off, end = sk.mirror(i["rot"], i["root"], i["parents"], i["off"], i["end"],
                                i["mapping"] if mode == "symmetry" else None, mode, axis)
    np.testing.assert_array_equal(i["root"], root0)  # not modified in place (the reference's 'symmetry' does)
    # quat.from_matrix picks one of four branches: a quaternion and its negative are the same rotation
    err = np.minimum(np.abs(r - want["rot"]).max(-1), np.abs(r + want["rot"]).max(-1)).max()
    assert err <= ATOL, err
    # ... and the SAME sign wherever the reference's branch is not a tie
    tie_el = _mirror_tie_elements(i["rot"], i["off"], i["parents"], i["mapping"] if mode == "symmetry" else None)
    same = np.abs(r - want["rot"]).max(-1) <= ATOL
    assert same[~tie_el].all(), (int((~same[~tie_el]).sum()), "sign flips off the branch ties")
    assert tie_el.mean() < 0.02
    assert_close(gt, want["gt"], 1e-7, "translation")
    assert_close(off, want["off"], 1e-7, "offsets")
    assert_close(end, want["end"], 1e-7, "end sites")
    torch, skt = _torch_mods()[:2]
    rt, *_ = skt.mirror(torch.from_numpy(i["rot"]).cuda(), torch.from_numpy(i["root"]).cuda(), torch.from_numpy(i["parents"]),
                        torch.from_numpy(i["off"]).cuda(), None, i["mapping"] if mode == "symmetry" else None, mode, axis)
    rt = rt.cpu().numpy()
    assert np.minimum(np.abs(rt - want["rot"]).max(-1), np.abs(rt + want["rot"]).max(-1)).max() <= ATOL
    # mirroring twice is the identity (up to the double cover)
    r2, gt2, off2, _ = sk.mirror(r, gt, i["parents"], off, None, i["mapping"] if mode == "symmetry" else None, mode, axis)
    assert np.minimum(np.abs(r2 - i["rot"]).max(-1), np.abs(r2 + i["rot"]).max(-1)).max() <= ATOL
    assert_close(gt2, i["root"], 1e-7, "translation twice")


def test_mirror_argument_errors():
    rot = np.zeros((2, 3, 4), np.float32)
    args = (rot, np.zeros((2, 3), np.float32), np.array([0, 0, 1]), np.zeros((3, 3), np.float32))
    with pytest.raises(ValueError):
        sk.mirror(*args, mode="symmetry")  # joints_mapping required
    with pytest.raises(ValueError):
        sk.mirror(*args, mode="nope")
    with pytest.raises(ValueError):
        sk.mirror(*args, axis="W")


@pytest.mark.usefixtures("lane_per_frame_at_test_sizes")
@pytest.mark.parametrize("J,kind", [(52, "body"), (31, "random"), (130, "random"), (40, "random"), (41, "chain"), (64, "chain"), (65, "random"),
                                    (96, "random"), (96, "chain"), (250, "random"), (251, "random"), (300, "chain"), (66, "chain"), (71, "chain"), (128, "chain"), (512, "chain")])
def test_mirror_big_skeletons_vs_oracle_composition(J, kind):
    """Bigger skeletons (8 / 4 frames per wave, odd joint counts, several 64-joint windows of the walk; from 40 joints on two or
    four list-scheduled chains per frame, up to 250 joints): check against the reference's chain rebuilt from oracle pieces
    (fk -> from_matrix -> negate -> from_global_rotations)."""
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(J)
    parents = {"body": syn.PARENTS_52, "random": syn.random_parents(J, rng)}.get(kind)
    if parents is None:  # a chain with two branch points
        parents = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
        parents[J // 2] = 0
        parents[3 * J // 4] = J // 4
    F = 257
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-1, 1, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.1)
    got, gt, o2, _ = sk.mirror(rot, root, parents, off, None, None, "all", "Y")
    from pymotion_amd import _lib
    # mode 'all', long (from 66 joints, from 52 when the row is whole 64-byte pieces), few open branch points
    assert ("mirror_deep_kernel" in _lib.last_kernel_name()) == (kind in ("chain", "body") and (J >= 66 or (J >= 52 and J % 4 == 0))), _lib.last_kernel_name()
    _, rm = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), parents)
    g = co.quat_from_matrix(rm)
    g[..., 1] *= -1  # axis Y -> components (1, 3)  (skeleton.py:313-315)
    g[..., 3] *= -1
    want = co.from_global_rotations(g, parents)
    err = np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max()
    assert err <= ATOL, err
    tie_el = _mirror_tie_elements(rot, off, parents, None)
    same = np.abs(got - want).max(-1) <= ATOL
    assert same[~tie_el].all() and tie_el.mean() < 0.01, (int((~same[~tie_el]).sum()), float(tie_el.mean()))
    assert_close(gt, root * np.array([1, -1, 1], np.float32), 0, "translation")
    assert_close(o2, off * np.array([1, -1, 1], np.float32), 0, "offsets")


@pytest.mark.parametrize("walk", ["dispatch", "scheduled"])
@pytest.mark.parametrize("J", [52, 97])
def test_mirror_symmetry_mapping_on_the_scheduled_walk(J, walk, monkeypatch):
    """mode='symmetry' permutes the joints' world rotations before they are made local again (skeleton.py:322-331); on skeletons
    that take the multi-chain walk, against the same composition of oracle pieces (up to the sign of each quaternion).  `dispatch`: what the production
    library picks (since round 6 the step-list walk, mirror_wide_kernel); `scheduled`: the chain scheduler's kernel (the tuning build with PM_MIRROR_WIDE=0)"""
    import contextlib

    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    if walk == "scheduled":
        monkeypatch.setenv("PM_MIRROR_WIDE", "0")
    ctx = _lib.variant("tuning") if walk == "scheduled" else contextlib.nullcontext()

    rng = np.random.default_rng(7 * J)
    parents = syn.PARENTS_52 if J == 52 else syn.random_parents(J, rng)
    mapping = np.arange(J)
    pairs = rng.permutation(np.arange(1, J))[: 2 * ((J - 1) // 3)].reshape(-1, 2)
    mapping[pairs[:, 0]], mapping[pairs[:, 1]] = pairs[:, 1], pairs[:, 0]  # an involution, root fixed
    F = 130
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-1, 1, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.1)
    with ctx:
        got, *_ = sk.mirror(rot, root, parents, off, None, mapping, "symmetry", "X")
        assert ("mirror_wide_kernel" in _lib.last_kernel_name()) == (walk == "dispatch"), _lib.last_kernel_name()
    _, rm = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), parents)
    g = co.quat_from_matrix(rm)[:, mapping]
    g[..., 2] *= -1  # axis X -> components (2, 3)  (skeleton.py:310-312)
    g[..., 3] *= -1
    want = co.from_global_rotations(g, parents)
    err = np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max()
    assert err <= ATOL, err


def test_fk_is_hip_graph_capturable_and_stream_ordered():
    """The C ABI does no allocation / synchronisation of its own: a call on torch's capturing stream lands in
    a HIP graph and replays with new inputs in the same buffers."""
    import ctypes as C

    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    torch = _torch_mods()[0]
    rot, root, off, parents = syn.fk_workload(2000, seed=21)
    d_rot, d_root, d_off = (torch.from_numpy(x).cuda() for x in (rot, root, off))
    pos = torch.zeros((2000, 22, 3), device="cuda")
    rm = torch.zeros((2000, 22, 3, 3), device="cuda")
    pp = parents.ctypes.data_as(C.c_void_p)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def launch():
        _lib.call("pm_fk_f32", p(d_rot), p(d_root), p(d_off), 0, pp, 2000, 22, p(pos), p(rm),
                  C.c_void_p(torch.cuda.current_stream().cuda_stream))

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()  # warm-up outside capture (module load)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    rot2, root2, _, _ = syn.fk_workload(2000, seed=22)
    d_rot.copy_(torch.from_numpy(rot2))
    d_root.copy_(torch.from_numpy(root2))
    pos.zero_()
    g.replay()
    torch.cuda.synchronize()
    p_o, r_o = co.fk(rot2.astype(np.float64), root2.astype(np.float64), off.astype(np.float64), parents)
    assert_close(pos.cpu().numpy(), p_o, ATOL, "graph replay pos")
    assert_close(rm.cpu().numpy(), r_o, ATOL, "graph replay rotmats")


def test_unroll_one_pass_is_hip_graph_capturable():
    """the look-back scan resets its ticket / status words with a memset node ahead of the kernel: captured once, the pair
    replays correctly any number of times on new data"""
    import ctypes as C

    from pymotion_amd import _lib

    torch = _torch_mods()[0]
    T, S = 20_000, 22
    rng = np.random.default_rng(8)

    def clip(seed):
        r = np.random.default_rng(seed)
        base = np.cumsum(r.normal(0, 0.08, (T, S, 4)), axis=0) + r.normal(0, 1, (1, S, 4))
        return (base * r.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)

    q = torch.from_numpy(clip(1)).cuda()
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)), dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def launch():
        _lib.call("pm_quat_unroll_f32", p(q), T, S, p(out), p(ws), C.c_void_p(torch.cuda.current_stream().cuda_stream))

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for seed in (2, 3, 4):
        host = clip(seed)
        q.copy_(torch.from_numpy(host))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), co.quat_unroll(host.astype(np.float64), 0).astype(np.float32))
    del rng


def test_maximum_joint_count_everywhere():
    """PM_MAX_JOINTS = 512: every skeleton kernel still fits its LDS tile; one more joint is rejected."""
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(512)
    J, F = 512, 6
    parents = syn.random_parents(J, rng)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-1, 1, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.02)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    g = co.quat_from_matrix(co.fk(f64(rot), np.zeros((F, 3)), f64(off), parents)[1])
    assert_close(sk.from_global_rotations(g.astype(np.float32), parents), co.from_global_rotations(f64(g.astype(np.float32)), parents), ATOL)
    got, *_ = sk.mirror(rot, root, parents, off, None, None, "all", "X")
    g[..., 2] *= -1
    g[..., 3] *= -1
    want = co.from_global_rotations(g, parents)
    assert np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max() <= 5e-5  # depth up to ~20 levels of fp32 products
    pos, _ = sk.fk(rot, np.zeros_like(root), off, parents)
    pos_or = co.fk(f64(rot), np.zeros((F, 3)), f64(off), parents)[0]
    assert_close(pos, pos_or, ATOL)
    # feed both sides the SAME (oracle) positions so that the comparison does not depend on the fk kernel
    pos32 = pos_or.astype(np.float32)
    r_ik = sk.from_root_positions(pos32, parents, off)
    r_or = co.from_root_positions(f64(pos32), parents, f64(off))
    d = np.minimum(np.abs(r_ik - r_or).max(-1), np.abs(r_ik + r_or).max(-1))
    # per record, tests/test_ik.py's bar beyond 128 joints: 2e-5 + what 64 ulps of the fp32 positions do to the reference's own float64
    # answer (a handful of the 3072 random bones sit next to from_to's anti-parallel case, where one ulp of input alone moves the
    # reference by ~1e-3; round 4 judged this call by quantiles with a 0.05 cap)
    from test_ik import _record_bar

    bar, sens = _record_bar(pos32, parents, off, r_or, k=64, draws=12)
    assert (d <= bar).all(), (float(d.max()), float(((d - 2e-5) / np.maximum(sens, 1e-12)).max()), int((d > bar).sum()))
    assert np.median(d) <= 1e-5, float(np.median(d))
    with pytest.raises(ValueError):
        sk.fk(np.zeros((2, 513, 4), np.float32), np.zeros((2, 3), np.float32), np.zeros((513, 3), np.float32), np.maximum(np.arange(513) - 1, 0))


def test_numpy_float64_output_opt_out():
    from pymotion_amd import config
    from pymotion_amd import synthetic as syn

    rot, root, off, par = syn.fk_workload(100, seed=2)
    assert sk.fk(rot, root, off, par)[0].dtype == np.float64
    config.numpy_float64_outputs = False
    try:
        p32, r32 = sk.fk(rot, root, off, par)
        assert p32.dtype == np.float32 and r32.dtype == np.float32
        assert quat.to_matrix(rot[0]).dtype == np.float32
    finally:
        config.numpy_float64_outputs = True
    assert_close(p32, sk.fk(rot, root, off, par)[0], 0, "same values, only the dtype differs")


def _tree(kind, J, rng):
    """topologies that stress the chain scheduler of to_root_dual_quat: pure chain, star, binary heap (breadth-first
    order), two long chains hanging off a deep trunk, random"""
    p = np.zeros(J, dtype=np.int32)
    if kind == "chain":
        p[1:] = np.arange(J - 1)
    elif kind == "star":
        p[:] = 0
    elif kind == "star_off_1":
        p[1:] = 1
        p[1] = 0
    elif kind == "heap":
        p[1:] = (np.arange(1, J) - 1) // 2
    elif kind == "two_arms":
        t = J // 3
        p[1:t] = np.arange(t - 1)
        p[t] = t - 1
        p[t + 1:2 * t] = np.arange(t, 2 * t - 1)
        p[2 * t] = t - 1
        p[2 * t + 1:] = np.arange(2 * t, J - 1)
    else:
        for i in range(1, J):
            p[i] = rng.integers(0, i)
    return p


@pytest.mark.parametrize("walk", ["dispatch", "scheduled"])
@pytest.mark.parametrize("kind", ["chain", "star", "star_off_1", "heap", "two_arms", "random", "random2"])
@pytest.mark.parametrize("J", [28, 31, 52, 64, 65, 100, 129, 250, 251])
def test_to_root_dq_chain_scheduler_on_many_topologies(kind, J, walk, monkeypatch):
    """28 <= J <= 250 walks several chains per frame (dq.hip: schedule_chains) when the tree is wide enough; every shape of
    tree -- including the ones that must fall back to one chain -- has to give the oracle's dual quaternions, and the
    decode has to bring the inputs back.  `dispatch`: what the production library picks (since round 6 the step-list kernel of dqwide.hip wherever its list
    holds the tree); `scheduled`: the chain scheduler's kernels (the tuning build with PM_DQ_WIDE=0: what unaligned arrays get)"""
    import contextlib

    from pymotion_amd import _lib

    if walk == "scheduled":
        monkeypatch.setenv("PM_DQ_WIDE", "0")
    ctx = _lib.variant("tuning") if walk == "scheduled" else contextlib.nullcontext()
    rng = np.random.default_rng(hash((kind, J)) % (1 << 31))
    parents = _tree(kind, J, rng)
    F = 37
    rot = rng.standard_normal((F, J, 4))
    rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
    gpos = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = rng.uniform(-0.3, 0.3, (J, 3)).astype(np.float32)
    off[0] = 0
    with ctx:
        d = sk.to_root_dual_quat(rot, gpos, parents, off)
        assert walk == "dispatch" or "wide_kernel" not in _lib.last_kernel_name()
    d_o = co.to_root_dual_quat(rot.astype(np.float64), gpos.astype(np.float64), parents, off.astype(np.float64))
    depth = 1
    dd = np.zeros(J, int)
    for i in range(1, J):
        dd[i] = dd[parents[i]] + 1
    depth = dd.max()
    assert_close(d, d_o, max(ATOL, 3e-7 * depth * depth), f"{kind} J={J} depth={depth}")  # (a 250-joint chain accumulates 250 fp32 products)
    t, q = sk.from_root_dual_quat(d, parents)
    assert_close(q, rot, max(ATOL, 3e-7 * depth * depth), "round trip rot")


@pytest.mark.parametrize("J,osc", [(22, 0.3), (22, 30.0), (52, 0.15), (52, 30.0), (100, 0.2)])
def test_every_skeleton_kernel_on_views_that_are_only_4_byte_aligned(J, osc):
    """base pointers that are not 16-byte aligned take the scalar-access instantiation of each kernel (VEC = false) --
    the several-chains dual-quaternion walk, the two-chain IK, both arithmetic levels of fk and the fused ortho6d source
    included -- with the aligned call's results (bit for bit where the arithmetic is a plain chain)"""
    torch, skt = _torch_mods()[:2]
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(J + int(osc))
    parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if parents is None:
        parents = syn.random_parents(J, rng)
    F = 203
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32) * (100 if osc > 1 else 1)
    off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
    off[0] = 0
    x6 = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
    par_t = torch.from_numpy(np.asarray(parents))

    def shifted(a):  # the same values at an address that is 4 bytes past a 16-byte boundary
        big = torch.empty(a.size + 1, dtype=torch.float32, device="cuda")
        v = big[1:].view(a.shape)
        v.copy_(torch.from_numpy(a))
        assert v.data_ptr() % 16 == 4
        return v

    al = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    for mk in (al, shifted):
        r, g, o, x = mk(rot), mk(root), mk(off), mk(x6)
        res = list(skt.fk(r, g, o, par_t)) + list(skt.fk_from_ortho6d(x, g, o, par_t, return_quat=True)) + list(skt.fk_from_ortho6d(x, g, o, par_t))
        dq = skt.to_root_dual_quat(r, g, par_t, o)
        res += [dq] + list(skt.from_root_dual_quat(mk(dq.cpu().numpy()), par_t))
        res += [skt.from_global_rotations(r, par_t), skt.from_root_positions(mk(res[0].cpu().numpy()), par_t, o)]
        res += [skt.mirror(r, g, par_t, o)[0]]
        if mk is al:
            want = [t.cpu().numpy() for t in res]
        else:
            for k, (t, w) in enumerate(zip(res, want)):
                if k >= 11:  # from_root_positions amplifies the last-ulp differences between two instantiations (where the
                    # compiler contracts a multiply-add) ~100x; mirror's from_matrix sign sits on branch boundaries
                    d = np.minimum(np.abs(t.cpu().numpy() - w).max(-1), np.abs(t.cpu().numpy() + w).max(-1))
                    assert np.median(d) <= 1e-6 and (d > 5e-4).mean() <= 2e-3, (k, np.median(d), d.max())
                    continue
                if 7 <= k <= 9 and (J >= 40 or (osc > 1 and J >= 20)):  # (below 40 joints: big bones, by the front door's scale hint)
                    # to_root_dual_quat: the aligned call walks one lane per frame in float64 (deep.hip), the 4-byte-aligned view stays on
                    # the tile kernel -- both within 2 ulp of the largest component of the float64 oracle, not bit-identical
                    # (8, 9: the decode of either result)
                    tn = t.cpu().numpy()
                    assert np.abs(tn - w).max() <= max(2e-6, 3 * 2.0 ** (np.floor(np.log2(np.abs(w).max())) - 23)), (k, np.abs(tn - w).max())
                    continue
                np.testing.assert_array_equal(t.cpu().numpy(), w, err_msg=f"result {k}")


def test_numpy_door_pipeline_from_two_threads_at_once():
    """the chunked pipeline shares its streams / events per device: concurrent big calls from different Python threads are
    serialised, small ones run the plain path beside them -- every result equals the single-threaded one"""
    import threading

    from pymotion_amd import synthetic as syn

    rot, root, off, par = syn.fk_workload(150_000, seed=31)
    want = sk.fk(rot, root, off, par)
    small = sk.fk(rot[:500], root[:500], off, par)
    out, errs = {}, []

    def big(k):
        try:
            out[k] = sk.fk(rot, root, off, par)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    def little(k):
        try:
            for _ in range(20):
                p, r = sk.fk(rot[:500], root[:500], off, par)
                np.testing.assert_array_equal(p, small[0])
            out[k] = True
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    ths = [threading.Thread(target=big, args=(0,)), threading.Thread(target=big, args=(1,)), threading.Thread(target=little, args=(2,))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for k in (0, 1):
        np.testing.assert_array_equal(out[k][0], want[0])
        np.testing.assert_array_equal(out[k][1], want[1])
