"""GPU: the fused ortho6d -> fk kernel equals the reference's chain ortho6d.to_quat -> fk on EVERY input (round-1 verdict,
"weak" item 2) -- zero, (anti-)parallel, nearly parallel, tiny and huge 6D columns, with and without the quaternion
output, through both doors, on both walk shapes.  Goldens: tests/golden/degenerate.npz, written by oracle/make_golden.py
from the imported reference:

    out64    NumPy reference on the inputs up-cast to float64  -> what the NumPy door (Gram-Schmidt without eps:
             rotations/ortho6d.py:83-85, zero column = NaN) is judged against
    out_t64  torch twin on float64 tensors (F.normalize eps = 1e-12: zero column = zeros, ortho6d_torch.py:84-89)
             -> what the torch door is judged against

Exactly (anti-)parallel columns have NO stable answer in the reference itself: what is left of the second column after the
projection is rounding noise (float64 NumPy returns a noise-directed rotation, fp32 NumPy NaN, fp32 torch the zero-column
result -- see the golden).  Those two records are only required to come out as NaN or as a proper rotation."""
import numpy as np
import pytest

from conftest import golden
from pymotion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
NAMES = ["zero_matrix", "zero_first_col", "zero_second_col", "parallel", "anti_parallel", "near_parallel", "tiny", "huge"]
UNSTABLE = ("parallel", "anti_parallel")


def _descendants(parents, j):
    out = {j}
    for k in range(j + 1, len(parents)):
        if parents[k] in out:
            out.add(k)
    return sorted(out)


def _masks(i, shape_fj):
    """rot_ok[f, j] / pos_ok[f, j]: where the reference's answer is stable"""
    rot_ok = np.ones(shape_fj, bool)
    pos_ok = np.ones(shape_fj, bool)
    for name, (f, j) in zip(NAMES, i["where"]):
        if name in UNSTABLE:
            d = _descendants(i["parents"], int(j))
            rot_ok[f, d] = False
            pos_ok[f, [k for k in d if k != j]] = False   # a joint's position depends on its PARENT's rotation
    return rot_ok, pos_ok


def _close(got, want, ok, atol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    sel = ok.reshape(ok.shape + (1,) * (got.ndim - ok.ndim)) & np.ones(got.shape, bool)
    assert (np.isnan(got[sel]) == np.isnan(want[sel])).all(), f"{what}: NaN pattern differs"
    fin = sel & ~np.isnan(want)
    assert np.abs(got[fin] - want[fin]).max() <= atol, f"{what}: {np.abs(got[fin] - want[fin]).max():.2e}"


def _proper_or_nan(rm, f, j):
    m = np.asarray(rm[f, j], np.float64)
    if np.isnan(m).any():
        return True
    return np.abs(m @ m.T - np.eye(3)).max() < 1e-5 and abs(np.linalg.det(m) - 1) < 1e-5


@pytest.mark.parametrize("J", [22, 52])
@pytest.mark.parametrize("want_q", [False, True])
def test_fused_ortho6d_fk_on_degenerate_records_numpy_door(J, want_q):
    import pymotion_amd.ops.skeleton as sk

    g = golden("degenerate.npz")
    case = f"fk_from_o6d_degenerate_J{J}"
    i, want = g.get(case, "in"), g.get(case, "out64")
    rot_ok, pos_ok = _masks(i, i["x"].shape[:2])
    out = sk.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"], return_quat=want_q)
    _close(out[0], want["pos"], pos_ok, 1e-5, "pos")
    _close(out[1], want["rotmats"], rot_ok, 1e-5, "rotmats")
    if want_q:
        q_ok = np.ones(rot_ok.shape, bool)
        for name, (f, j) in zip(NAMES, i["where"]):
            q_ok[f, j] = name not in UNSTABLE
        _close(out[2], want["quat"], q_ok, 1e-5, "quat")
    for name, (f, j) in zip(NAMES, i["where"]):
        if name in UNSTABLE:
            assert _proper_or_nan(out[1], f, j), name


@pytest.mark.parametrize("J", [22, 52])
@pytest.mark.parametrize("want_q", [False, True])
def test_fused_ortho6d_fk_on_degenerate_records_torch_door(J, want_q):
    import torch

    import pymotion_amd.ops.skeleton_torch as skt

    g = golden("degenerate.npz")
    case = f"fk_from_o6d_degenerate_J{J}"
    i, want = g.get(case, "in"), g.get(case, "out_t64")
    rot_ok, pos_ok = _masks(i, i["x"].shape[:2])
    out = skt.fk_from_ortho6d(torch.from_numpy(i["x"]).cuda(), torch.from_numpy(i["gpos"]).cuda(), torch.from_numpy(i["off"]).cuda(),
                              torch.from_numpy(i["parents"]), return_quat=want_q)
    out = [o.cpu().numpy() for o in out]
    assert not np.isnan(out[1][rot_ok]).any()           # the torch twin never produces NaN from zero columns
    _close(out[0], want["pos"], pos_ok, 1e-5, "pos")
    _close(out[1], want["rotmats"], rot_ok, 1e-5, "rotmats")
    if want_q:
        q_ok = np.ones(rot_ok.shape, bool)
        for name, (f, j) in zip(NAMES, i["where"]):
            q_ok[f, j] = name not in UNSTABLE
        _close(out[2], want["quat"], q_ok, 1e-5, "quat")
    for name, (f, j) in zip(NAMES, i["where"]):
        if name in UNSTABLE:
            assert _proper_or_nan(out[1], f, j), name


@pytest.mark.parametrize("J", [22, 52])
def test_with_and_without_the_quaternion_output_agree_everywhere(J):
    """the shortcut variant (no quaternion materialised) must be the same function as the full one, degenerate records included"""
    import pymotion_amd.ops.skeleton as sk

    i = golden("degenerate.npz").get(f"fk_from_o6d_degenerate_J{J}", "in")
    a = sk.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"], return_quat=False)
    b = sk.fk_from_ortho6d(i["x"], i["gpos"], i["off"], i["parents"], return_quat=True)
    for x, y, what in ((a[0], b[0], "pos"), (a[1], b[1], "rotmats")):
        assert (np.isnan(x) == np.isnan(y)).all(), what
        assert np.nanmax(np.abs(x - y)) <= 2e-6, what


@pytest.mark.parametrize("J", [22, 52])
def test_elementwise_ortho6d_conversions_on_degenerate_records(J):
    import torch

    import pymotion_amd.rotations.ortho6d as o6
    import pymotion_amd.rotations.ortho6d_torch as o6t

    g = golden("degenerate.npz")
    for fn_np, fn_t, case in ((o6.to_quat, o6t.to_quat, f"o6d_to_quat_degenerate_J{J}"), (o6.to_matrix, o6t.to_matrix, f"o6d_to_matrix_degenerate_J{J}")):
        i = g.get(case, "in")
        where = g.get(f"fk_from_o6d_degenerate_J{J}", "in")["where"]
        ok = np.ones(i["x"].shape[:2], bool)
        for name, (f, j) in zip(NAMES, where):
            ok[f, j] = name not in UNSTABLE
        _close(fn_np(i["x"]), g.get(case, "out64")["out"], ok, 1e-5, case + " numpy")
        _close(fn_t(torch.from_numpy(i["x"]).cuda()).cpu().numpy(), g.get(case, "out_t64")["out"], ok, 1e-5, case + " torch")


def test_random_batch_with_sprinkled_degenerate_records_at_config4_size():
    """2^18 x 52 random 6D with ~0.1 % zero / parallel records: everything else must be untouched by the rare float64 branch
    (same values as a batch without them), and the fused kernel still equals the GPU's own two-launch chain"""
    import torch

    import pymotion_amd.ops.skeleton_torch as skt
    import pymotion_amd.rotations.ortho6d_torch as o6t

    F = 1 << 16
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    x = torch.randn((F, 52, 3, 2), generator=g, device="cuda")
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off = torch.from_numpy(syn.make_offsets(52, np.random.default_rng(4), 0.15)).cuda()
    par = torch.from_numpy(syn.PARENTS_52)
    p0, r0 = skt.fk_from_ortho6d(x, root, off, par)
    x2 = x.clone()
    hit = torch.rand((F, 52), generator=g, device="cuda") < 1e-3
    x2[hit] = 0.0
    p1, r1 = skt.fk_from_ortho6d(x2, root, off, par)
    clean = ~hit.any(dim=1)                               # frames without any zeroed record: bit-identical
    assert bool(torch.equal(p0[clean], p1[clean])) and bool(torch.equal(r0[clean], r1[clean]))
    assert bool(torch.isfinite(r1).all())                 # torch door: zero columns never give NaN
    p2, r2 = skt.fk(o6t.to_quat(x2), root, off, par)      # the two-launch chain on the GPU
    # every frame, no conditioning allowance (round 2: 5e-5 on the 97 % best-conditioned frames; Gram-Schmidt no longer
    # loses accuracy near (anti-)parallel columns -- common.hpp: o6d2m)
    assert float((r1 - r2).abs().max()) < 1e-5
    assert float((p1 - p2).abs().max()) < 1e-5
