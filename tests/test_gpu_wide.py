"""GPU: fk on long skeletons whose tree is wide (fkwide.hip: a wave per frame, its lanes over the joints of a host-made step list).

Beyond 128 joints a tree the streamed walk declines (more cross-chunk branch points than register sets: random / bushy trees) used to
fall to the four-frame tile kernels (34 % of the HBM spec at J = 129, 8.6 % at 512).  Checked here on the PRODUCTION library: which kernel
ran, parity with the float64 C oracle on metre and centimetre data (per-frame PREC_DYN: float64 rotations + fixed-point chain), every
alignment of a frame's rows (J mod 4, F odd), single frames, the root position bit for bit, NaN / Inf where the reference has them,
outputs that are views into bigger buffers (nothing written outside), trees too deep for the step list falling back, and the host
scheduler's invariants (CPU).  Later in round 5 the same walk took fk's other sources -- per-frame offsets, the fused ortho6d source with and
without its quaternions -- beyond 128 joints and where it measured faster below: oracle parity, degenerate and NaN records, NaN / Inf offsets,
every source at 510 / 511 / 512 joints (the two-launch path where nothing else fits)."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn


def _ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def _batch(F, J, seed, osc, rsc):
    rng = np.random.default_rng(seed)
    rot = rng.standard_normal((F, J, 4))
    rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True) * rng.uniform(0.5, 2.0, (F, J, 1))).astype(np.float32)  # fk normalises (skeleton.py:45)
    root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
    off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
    off[0] = 0
    return rot, root, off


def bushy(J, seed=None):
    return syn.random_parents(J, np.random.default_rng(J if seed is None else seed))


def broom(J, handle):
    """a chain of `handle` joints with everything else hanging off its last joint: depth handle + 1, one very wide level"""
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    p[handle:] = handle - 1
    return p


def comb(J):
    """a spine of J / 4 joints, each with a three-joint tooth: parents far back in the table"""
    n = J // 4
    p = [0] + list(range(0, n - 1))
    for s in range(n):
        p += [s, len(p), len(p) + 1]
    while len(p) < J:
        p.append(len(p) - 1)
    return np.asarray(p[:J], dtype=np.int32)


WIDE_CASES = [(129, "bushy"), (130, "bushy"), (131, "bushy"), (160, "bushy"), (200, "bushy"), (255, "bushy"), (256, "bushy"), (300, "bushy"),
              (400, "bushy"), (511, "bushy"), (512, "bushy"), (300, "broom"), (512, "broom")]


def _parents(kind, J):
    return {"bushy": bushy, "broom": lambda j: broom(j, 10), "comb": comb}[kind](J)


@pytest.mark.gpu
@pytest.mark.parametrize("J,kind", WIDE_CASES)
def test_fk_wide_walk_against_the_oracle(J, kind):
    import pymotion_amd.ops.skeleton as sk

    parents = _parents(kind, J)
    assert (parents[1:] < np.arange(1, J)).all()
    depth = int(syn.depth_of(parents).max())
    for F, osc, rsc in ((1, 0.1, 2.0), (3, 10.0, 200.0), (16, 0.1, 2.0), (77, 10.0, 200.0), (333, 0.1, 2.0), (1001, 10.0, 2.0)):
        rot, root, off = _batch(F, J, 9000 * J + F, osc, rsc)
        pos, rm = sk.fk(rot, root, off, parents)
        name = _lib.last_kernel_name()
        assert "fk_wide_kernel" in name, (name, J, kind)
        p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)
        # rotations: an fp32 chain of `depth` 3 x 3 products; positions: those errors times the bones, or 2 ulp of the largest coordinate
        assert np.abs(rm - r_o).max() <= max(2e-6, 2.5e-7 * depth), (F, np.abs(rm - r_o).max())
        bar = max(1e-5, 3 * _ulp_of(p_o)) if osc < 1 else max(1e-5, 2 * _ulp_of(p_o), 4e-7 * depth * osc * 3)
        assert np.abs(pos - p_o).max() <= bar, (F, osc, np.abs(pos - p_o).max() / _ulp_of(p_o), "ulp")
        np.testing.assert_array_equal(pos[:, 0].astype(np.float32), root)  # the root is the caller's value (skeleton.py:49)


@pytest.mark.gpu
@pytest.mark.parametrize("J,kind", [(96, "bushy"), (100, "bushy"), (104, "bushy"), (112, "bushy"), (120, "bushy"), (128, "bushy"), (129, "bushy"), (131, "bushy"), (160, "comb"), (192, "bushy"), (200, "bushy"), (256, "bushy"), (300, "broom"), (384, "bushy"),
                                    (400, "bushy"), (512, "bushy")])
def test_fk_wide_walk_other_sources_against_the_oracle(J, kind):
    """per-frame offsets, the fused ortho6d source with and without its quaternions, both together: the same walk (same kernel template),
    metre and centimetre data, single frames and odd batches"""
    import pymotion_amd.ops.skeleton as sk

    parents = _parents(kind, J)
    depth = int(syn.depth_of(parents).max())
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    for F, osc, rsc in ((1, 0.1, 2.0), (5, 10.0, 200.0), (77, 0.1, 2.0), (602, 10.0, 2.0)):
        rot, root, off = _batch(F, J, 77 * J + F, osc, rsc)
        rng = np.random.default_rng(J + F)
        x6 = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
        offs = (off[None] * np.linspace(0.7, 1.3, F, dtype=np.float32)[:, None, None]).astype(np.float32)
        rot_bar = max(2e-6, 2.5e-7 * depth)

        def pos_bar(p_o):
            return max(1e-5, 3 * _ulp_of(p_o)) if osc < 1 else max(1e-5, 2 * _ulp_of(p_o), 4e-7 * depth * osc * 3)

        # up to 128 joints the pipelined tiles keep these sources except where their image is padded (multiples of sixteen joints, from 101 on)
        # and, from 93 joints on, for the ortho6d source with per-frame offsets AND quaternions (fk.hip: wide_mid)
        def wide_expected(all_three):
            return J > 128 or (J > 100 and J % 16 == 0) or (all_three and J > 92)

        pos, rm = sk.fk(rot, root, offs, parents)
        assert ("fk_wide_kernel" in _lib.last_kernel_name() and "true>" in _lib.last_kernel_name()) == wide_expected(False), _lib.last_kernel_name()
        p_o, r_o = co.fk(f64(rot), f64(root), f64(offs), parents)
        assert np.abs(rm - r_o).max() <= rot_bar and np.abs(pos - p_o).max() <= pos_bar(p_o), (F, osc, np.abs(pos - p_o).max() / _ulp_of(p_o))
        np.testing.assert_array_equal(pos[:, 0].astype(np.float32), root)
        q_o = co.o6d_to_quat(f64(x6))
        for o_in in (off, offs):
            p_o, r_o = co.fk(q_o, f64(root), f64(o_in), parents)
            for want_q in (False, True):
                out = sk.fk_from_ortho6d(x6, root, o_in, parents, return_quat=want_q)
                assert ("fk_wide_kernel" in _lib.last_kernel_name()) == wide_expected(want_q and o_in is offs), (_lib.last_kernel_name(), want_q)
                # (the conversion's own fp32 error, 1e-6 a record, rides the chain)
                assert np.abs(out[1] - r_o).max() <= max(1e-5, 4 * rot_bar), (F, np.abs(out[1] - r_o).max())
                assert np.abs(out[0] - p_o).max() <= max(2e-5, 4 * pos_bar(p_o)), (F, osc, np.abs(out[0] - p_o).max())
                np.testing.assert_array_equal(out[0][:, 0].astype(np.float32), root)
                if want_q:
                    assert np.minimum(np.abs(out[2] - q_o).max(-1), np.abs(out[2] + q_o).max(-1)).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("J", [131, 256, 500])
def test_fk_wide_walk_with_degenerate_ortho6d_records(J):
    """zero / parallel records sprinkled over a batch: the frames without any are bit-identical to the batch without them, the fused kernel
    equals the GPU's own two-launch chain (ortho6d.to_quat, then fk) on every frame, and so do its quaternions"""
    import torch

    import pymotion_amd.ops.skeleton_torch as skt
    import pymotion_amd.rotations.ortho6d_torch as o6t

    F = 3001
    parents = bushy(J)
    g = torch.Generator(device="cuda")
    g.manual_seed(J)
    x = torch.randn((F, J, 3, 2), generator=g, device="cuda")
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(4), 0.15)).cuda()
    par = torch.from_numpy(parents)
    p0, r0, q0 = skt.fk_from_ortho6d(x, root, off, par, return_quat=True)
    assert "fk_wide_kernel" in _lib.last_kernel_name()
    x2 = x.clone()
    hit = torch.rand((F, J), generator=g, device="cuda") < 1e-3
    x2[hit] = 0.0
    par_hit = torch.rand((F, J), generator=g, device="cuda") < 1e-3
    x2[..., 1] = torch.where(par_hit[..., None], -2.0 * x2[..., 0], x2[..., 1])  # exactly anti-parallel columns
    hit = hit | par_hit
    p1, r1, q1 = skt.fk_from_ortho6d(x2, root, off, par, return_quat=True)
    clean = ~hit.any(dim=1)
    assert int(clean.sum()) > 100 and int((~clean).sum()) > 100
    assert bool(torch.equal(p0[clean], p1[clean])) and bool(torch.equal(r0[clean], r1[clean])) and bool(torch.equal(q0[clean], q1[clean]))
    assert bool(torch.isfinite(r1).all()) and bool(torch.isfinite(q1).all())
    pa, ra = skt.fk_from_ortho6d(x2, root, off, par)  # without the quaternion output: the same function
    assert bool(torch.equal(pa, p1)) and bool(torch.equal(ra, r1))
    q2 = o6t.to_quat(x2)
    p2, r2 = skt.fk(q2, root, off, par)
    keep = ~par_hit  # exactly (anti-)parallel columns have no stable answer in the reference itself (tests/test_gpu_degenerate.py)
    assert float((torch.minimum((q1 - q2).abs().amax(-1), (q1 + q2).abs().amax(-1)))[keep].max()) < 1e-5
    ok = ~par_hit.any(dim=1)
    assert float((r1 - r2)[ok].abs().max()) < 2e-5 and float((p1 - p2)[ok].abs().max()) < 2e-5


@pytest.mark.gpu
def test_fk_wide_walk_does_not_depend_on_a_frames_place_in_the_batch():
    """frames are independent: a frame's result must not depend on its place in the batch (XCD tile order, the alignment of its rows in
    HBM and in the LDS image).  (The wide walk against the tile kernels on the same arrays, bit for bit on metre data -- same local
    rotations, same products in the same order: tools/fk_wide_sweep.py, profiles/r05_fk_wide_sweep.txt.)"""
    import pymotion_amd.ops.skeleton as sk

    J, F = 200, 500
    parents = bushy(J, 5)
    rot, root, off = _batch(F, J, 4242, 0.1, 2.0)
    pos, rm = sk.fk(rot, root, off, parents)
    assert "fk_wide_kernel" in _lib.last_kernel_name()
    perm = np.random.default_rng(1).permutation(F)
    pos2, rm2 = sk.fk(rot[perm], root[perm], off, parents)
    np.testing.assert_array_equal(pos2, pos[perm])
    np.testing.assert_array_equal(rm2, rm[perm])


@pytest.mark.gpu
@pytest.mark.parametrize("J,shift_pos,shift_rot", [(129, 4, 0), (200, 0, 12), (131, 20, 8), (511, 28, 28)])
def test_fk_wide_walk_with_outputs_inside_bigger_buffers(J, shift_pos, shift_rot):
    """raw ABI: `pos` / `rotmats` 16-byte aligned views into bigger buffers: same bits, nothing written outside the arrays (the first and
    last floats of a frame's rows are written one by one)"""
    import torch

    F = 77
    parents = bushy(J)
    rot, root, off = _batch(F, J, 31 * J, 10.0, 200.0)
    dev = torch.device("cuda:0")
    rot_d, root_d, off_d = (torch.from_numpy(x).to(dev) for x in (rot, root, off))
    pp = parents.astype(np.int32).ctypes.data_as(C.c_void_p)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pos0 = torch.empty(F * J * 3, device=dev)
    rm0 = torch.empty(F * J * 9, device=dev)
    _lib.call("pm_fk_f32", P(rot_d), P(root_d), P(off_d), 0, pp, F, J, P(pos0), P(rm0), None)
    assert "fk_wide_kernel" in _lib.last_kernel_name()
    pad = 64
    pos_b = torch.full((F * J * 3 + 2 * pad,), -7.0, device=dev)
    rm_b = torch.full((F * J * 9 + 2 * pad,), -7.0, device=dev)
    pos1 = pos_b[pad + shift_pos: pad + shift_pos + F * J * 3]
    rm1 = rm_b[pad + shift_rot: pad + shift_rot + F * J * 9]
    assert pos1.data_ptr() % 16 == 0 and rm1.data_ptr() % 16 == 0
    _lib.call("pm_fk_f32", P(rot_d), P(root_d), P(off_d), 0, pp, F, J, P(pos1), P(rm1), None)
    assert "fk_wide_kernel" in _lib.last_kernel_name()
    torch.cuda.synchronize()
    assert torch.equal(pos1.view(torch.int32), pos0.view(torch.int32))
    assert torch.equal(rm1.view(torch.int32), rm0.view(torch.int32))
    for buf, lo, n in ((pos_b, pad + shift_pos, F * J * 3), (rm_b, pad + shift_rot, F * J * 9)):
        assert bool((buf[:lo] == -7.0).all()) and bool((buf[lo + n:] == -7.0).all())
    p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)
    assert np.abs(pos0.cpu().numpy().reshape(F, J, 3) - p_o).max() <= max(1e-5, 2 * _ulp_of(p_o), 4e-7 * 30 * 30)


@pytest.mark.gpu
def test_fk_wide_walk_keeps_nan_and_inf_where_the_reference_has_them():
    """a NaN quaternion poisons its joint and everything below it, a NaN root coordinate its row of every position of the frame, on metre
    data and on data that takes the fixed-point chain (a frame with a non-finite input takes the float walk, which propagates them)"""
    import pymotion_amd.ops.skeleton as sk

    J = 200
    parents = bushy(J, 11)
    for osc, rsc in ((0.1, 2.0), (10.0, 200.0)):
        F = 100
        rot, root, off = _batch(F, J, 99, osc, rsc)
        rot[7, 150, 1] = np.nan
        rot[40, 1, 0] = np.nan       # near the root: much of the skeleton
        rot[41, 0, 3] = np.inf
        root[60, 2] = np.nan
        root[61, 0] = np.inf
        with np.errstate(all="ignore"):
            p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)
        pos, rm = sk.fk(rot, root, off, parents)
        assert "fk_wide_kernel" in _lib.last_kernel_name()
        assert (np.isnan(rm) == np.isnan(r_o)).all()
        assert (np.isnan(pos) == np.isnan(p_o)).all()
        fin = np.isfinite(p_o)
        assert np.abs(pos[fin] - p_o[fin]).max() <= max(1e-5, 3 * _ulp_of(p_o[fin]))


@pytest.mark.gpu
@pytest.mark.parametrize("J", [200, 512])
def test_fk_wide_walk_other_sources_keep_nan_and_inf(J):
    """the same on per-frame offsets (a NaN / Inf offset: its joint's position and everything below, and -- the reference's 4 x 4 products --
    row r of the rotations below it) and on the ortho6d source (a NaN record), metre and centimetre data"""
    import pymotion_amd.ops.skeleton as sk

    parents = bushy(J, 11)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    for osc, rsc in ((0.1, 2.0), (10.0, 200.0)):
        F = 90
        rot, root, off = _batch(F, J, 98, osc, rsc)
        offs = (off[None] * np.linspace(0.7, 1.3, F, dtype=np.float32)[:, None, None]).astype(np.float32)
        offs[5, 3, 1] = np.nan
        offs[6, J - 1, 0] = np.nan     # a leaf: nothing below
        offs[7, 2, 2] = np.inf
        offs[8, 0, 0] = np.nan         # offsets[0] is ignored (skeleton.py:49)
        rot[20, 17, 2] = np.nan
        root[30, 1] = np.inf
        with np.errstate(all="ignore"):
            p_o, r_o = co.fk(f64(rot), f64(root), f64(offs), parents)
        pos, rm = sk.fk(rot, root, offs, parents)
        assert "fk_wide_kernel" in _lib.last_kernel_name()
        assert (np.isnan(rm) == np.isnan(r_o)).all() and (np.isnan(pos) == np.isnan(p_o)).all()
        assert (np.isinf(pos) == np.isinf(p_o)).all()
        fin = np.isfinite(p_o)
        assert np.abs(pos[fin] - p_o[fin]).max() <= max(1e-5, 3 * _ulp_of(p_o[fin]))
        assert np.isfinite(pos[8]).all()
        x6 = np.random.default_rng(J).standard_normal((F, J, 3, 2)).astype(np.float32)
        x6[11, 9, 1, 0] = np.nan
        with np.errstate(all="ignore"):
            q_o = co.o6d_to_quat(f64(x6))
            p_o, r_o = co.fk(q_o, f64(root), f64(offs), parents)
        pos, rm, q = sk.fk_from_ortho6d(x6, root, offs, parents, return_quat=True)
        assert "fk_wide_kernel" in _lib.last_kernel_name()
        assert (np.isnan(rm) == np.isnan(r_o)).all() and (np.isnan(pos) == np.isnan(p_o)).all() and (np.isnan(q) == np.isnan(q_o)).all()
        fin = np.isfinite(p_o)
        assert np.abs(pos[fin] - p_o[fin]).max() <= max(2e-5, 3 * _ulp_of(p_o[fin]))


@pytest.mark.gpu
def test_fk_trees_too_deep_for_the_step_list_fall_back():
    """a 300-joint comb (parents far back in the table: the streamed walk declines it; depth 78: more than the step list holds) still
    gets the right answer from the tile kernels, whatever the source"""
    import pymotion_amd.ops.skeleton as sk

    J, F = 300, 50
    parents = comb(J)
    rot, root, off = _batch(F, J, 5, 0.1, 2.0)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    pos, rm = sk.fk(rot, root, off, parents)
    name = _lib.last_kernel_name()
    assert "fk_wide_kernel" not in name, name
    p_o, r_o = co.fk(f64(rot), f64(root), f64(off), parents)
    assert np.abs(pos - p_o).max() <= 1e-5 and np.abs(rm - r_o).max() <= 2e-5
    offs = (off[None] * np.linspace(0.8, 1.2, F, dtype=np.float32)[:, None, None]).astype(np.float32)
    pos, rm = sk.fk(rot, root, offs, parents)
    assert "fk_wide_kernel" not in _lib.last_kernel_name()
    p_o, r_o = co.fk(f64(rot), f64(root), f64(offs), parents)
    assert np.abs(pos - p_o).max() <= 1e-5 and np.abs(rm - r_o).max() <= 2e-5
    x6 = np.random.default_rng(3).standard_normal((F, J, 3, 2)).astype(np.float32)
    pos, rm, q = sk.fk_from_ortho6d(x6, root, offs, parents, return_quat=True)
    assert "fk_wide_kernel" not in _lib.last_kernel_name()
    q_o = co.o6d_to_quat(f64(x6))
    p_o, r_o = co.fk(q_o, f64(root), f64(offs), parents)
    assert np.abs(pos - p_o).max() <= 2e-5 and np.abs(rm - r_o).max() <= 4e-5


@pytest.mark.gpu
@pytest.mark.parametrize("J", [510, 511, 512])
@pytest.mark.parametrize("kind", ["chain", "bushy"])
def test_fk_every_source_at_the_maximum_joint_count(J, kind):
    """PM_MAX_JOINTS and the two counts below it, every source and output of fk, on a chain (the step list declines it) and a random tree:
    the ortho6d source with per-frame offsets AND the quaternion output needs 164 KB for a four-frame tile at 511 / 512 joints -- a chain
    there runs as the reference's own two steps (ortho6d.to_quat, then fk) instead of being refused"""
    import pymotion_amd.ops.skeleton as sk

    F = 37
    parents = np.maximum(np.arange(J) - 1, 0).astype(np.int32) if kind == "chain" else bushy(J)
    rot, root, off = _batch(F, J, J, 0.02, 2.0)
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    offs = (off[None] * np.linspace(0.8, 1.2, F, dtype=np.float32)[:, None, None]).astype(np.float32)
    x6 = np.random.default_rng(J).standard_normal((F, J, 3, 2)).astype(np.float32)
    q_o = co.o6d_to_quat(f64(x6))
    depth = int(syn.depth_of(parents).max())
    rot_bar, pos_bar = max(2e-5, 5e-7 * depth), max(2e-5, 1e-6 * depth)
    for o_in in (off, offs):
        pos, rm = sk.fk(rot, root, o_in, parents)
        p_o, r_o = co.fk(f64(rot), f64(root), f64(o_in), parents)
        assert np.abs(pos - p_o).max() <= pos_bar and np.abs(rm - r_o).max() <= rot_bar
        p_o, r_o = co.fk(q_o, f64(root), f64(o_in), parents)
        for want_q in (False, True):
            out = sk.fk_from_ortho6d(x6, root, o_in, parents, return_quat=want_q)
            assert np.abs(out[0] - p_o).max() <= 2 * pos_bar and np.abs(out[1] - r_o).max() <= 2 * rot_bar, (np.abs(out[0] - p_o).max(), np.abs(out[1] - r_o).max())
            if want_q:
                assert np.minimum(np.abs(out[2] - q_o).max(-1), np.abs(out[2] + q_o).max(-1)).max() <= 1e-5


def test_fk_wide_plan_schedules_every_joint_once_after_its_parent():
    """CPU: the host's step list (pm_fk_wide_plan_debug) -- every joint but the root exactly once, a step after its parent's, idle quads on the idle slot, bounded by ceil(J / 16) + depth steps; trees that need more than 48 steps are declined"""
    h = _lib.lib()
    buf = (C.c_uint32 * (50 * 16))()
    for J, parents in [(129, bushy(129)), (512, bushy(512)), (512, broom(512, 30)), (200, bushy(200, 3)), (17, bushy(17)), (1, np.zeros(1, np.int32)),
                       (300, comb(300)), (512, np.maximum(np.arange(512) - 1, 0).astype(np.int32))]:
        n = h.pm_fk_wide_plan_debug(parents.astype(np.int32).ctypes.data_as(C.c_void_p), J, buf)
        depth = int(syn.depth_of(parents).max())
        if n == -4:  # PM_EUNSUPPORTED: declined: only when the list-scheduling bound itself exceeds the 48 steps the kernel's list holds
            assert (J + 14) // 16 + depth > 48, (J, depth)
            continue
        assert depth <= n <= min(48, (J + 14) // 16 + depth), (J, n, depth)
        words = np.frombuffer(buf, dtype=np.uint32)[: (n + 2) * 16].reshape(n + 2, 16)
        step_of = {0: -1}  # the root takes no step (its slot holds R_0 = L_0 as parked)
        for s in range(n):
            for w in words[s]:
                j, p = int(w & 0xFFFF), int(w >> 16)
                if j == J + 1:
                    assert p == J
                    continue
                assert 0 < j < J and j not in step_of
                step_of[j] = s
                assert p == parents[j] and step_of[p] < s
        assert len(step_of) == J
        assert (words[n:] == ((J + 1) | (J << 16))).all()


@pytest.mark.gpu
@pytest.mark.parametrize("J,chains", [(104, 8), (112, 8), (128, 8), (160, 8), (192, 16), (200, 16), (250, 16)])
def test_to_root_dual_quat_on_long_wide_trees(J, chains, monkeypatch):
    """random trees beyond 100 joints (deep.hip's lane-per-frame kernels decline them: too many open branch points) walk eight / sixteen
    chains per frame in the scheduled kernel (two / one frame a wave, many tiles per workgroup): which instance ran, parity with the float64
    oracle on metre and centimetre data (the precise step), single frames, partial tiles and groups, the round trip back.  (Since round 6 the dispatch asks the
    step-list kernel of dqwide.hip first -- tests/test_gpu_dqwide.py --; these instances are what a call gets whose arrays are not 16-byte aligned, and run
    here on the tuning build with PM_DQ_WIDE=0.)"""
    import pymotion_amd.ops.skeleton as sk

    monkeypatch.setenv("PM_DQ_WIDE", "0")
    parents = bushy(J)
    for F, osc, rsc in ((1, 0.3, 2.0), (2, 30.0, 200.0), (3, 0.3, 2.0), (65, 30.0, 200.0), (1000, 0.3, 2.0), (20_001, 30.0, 200.0), (70_001, 0.3, 2.0)):
        rng = np.random.default_rng(17 * J + F)
        rot = rng.standard_normal((F, J, 4))
        rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
        root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
        off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
        off[0] = 0
        with _lib.variant("tuning"):
            d = sk.to_root_dual_quat(rot, root, parents, off)
            name = _lib.last_kernel_name()
        assert "to_root_dq_sched_kernel<%d," % chains in name, (name, J, F)
        d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
        assert np.abs(d - d_o).max() <= max(1e-5, 3 * _ulp_of(d_o)), (F, osc, np.abs(d - d_o).max() / _ulp_of(d_o))
        t, q = sk.from_root_dual_quat(d, parents)
        assert np.abs(q - rot).max() <= 4e-6
        assert np.abs(t[:, 1:] - off[1:]).max() <= 4e-6 * max(1.0, np.abs(d_o).max())

