"""GPU: fk's four-frame pipelined kernel walking four JOINTS of a frame at a time (fk.hip: tree_walk_w4 -- sixteen quads = four frames x four
slots of a host-made step list; humanoids and other wide trees of 24...100 joints, SMPL-H and BASELINE config 4 among them).

Checked: parity with the float64 C oracle for the quaternion source, per-frame offsets, the fused ortho6d source with and without the
quaternion output, on metre and centimetre data (the fixed-point chain rides the same step), every tile remainder (F = 1 ... 4 k + 3), several
tiles per workgroup, the padded image of joint counts that are multiples of sixteen; bit equality with the one-joint-at-a-time walk it replaces
(tuning build, PM_FK_W4 = 0 / 1); which trees qualify (a chain does not); the step list's invariants at width four on the CPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn


def _ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def humanoid(J):
    """spine, head, legs, arms, three-joint fingers off the wrists for as many joints as are left (tests/test_gpu_deep.py)"""
    p = [0]

    def chain(start, n):
        for i in range(n):
            p.append(start if i == 0 else len(p) - 1)
        return len(p) - 1

    se = chain(0, 6)
    chain(se, 3)
    chain(0, 5)
    chain(0, 5)
    lw = chain(se, 4)
    rw = chain(se, 4)
    side = 0
    while len(p) + 3 <= J:
        chain(lw if side == 0 else rw, 3)
        side ^= 1
    while len(p) < J:
        p.append(len(p) - 1)
    return np.asarray(p[:J], dtype=np.int32)


def _tree(kind, J):
    if kind == "smplh":
        return syn.PARENTS_52
    if kind == "humanoid":
        return humanoid(J)
    if kind == "chain":
        p = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
        p[J // 2] = 0
        return p
    return syn.random_parents(J, np.random.default_rng(J))


def _data(F, J, seed, osc, rsc):
    rng = np.random.default_rng(seed)
    rot = (rng.standard_normal((F, J, 4)) * rng.uniform(0.5, 2.0, (F, J, 1))).astype(np.float32)  # fk normalises (skeleton.py:45)
    root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
    off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
    off[0] = 0
    x6 = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
    return rot, root, off, x6


CASES = [(52, "smplh"), (24, "random"), (30, "random"), (40, "humanoid"), (48, "random"), (48, "humanoid"), (64, "humanoid"), (64, "random"), (80, "humanoid"),
         (92, "random"), (100, "humanoid")]


@pytest.mark.gpu
@pytest.mark.parametrize("J,kind", CASES)
def test_fk_four_joints_a_step_against_the_oracle(J, kind):
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    depth = int(syn.depth_of(parents).max())
    f64 = lambda a: a.astype(np.float64)  # noqa: E731
    for F, osc, rsc in ((1, 0.15, 2.0), (3, 20.0, 150.0), (4, 0.15, 2.0), (5, 0.15, 2.0), (17, 20.0, 150.0), (1001, 0.15, 2.0), (70_002, 20.0, 150.0)):
        rot, root, off, x6 = _data(F, J, 100 * J + F, osc, rsc)
        pos_bar = max(1e-5, 3 * _ulp_of(root)) if osc < 1 else max(1e-5, 4e-7 * depth * osc * 3)
        pos, rm = sk.fk(rot, root, off, parents)
        assert "fk_pipe_kernel" in _lib.last_kernel_name(), (_lib.last_kernel_name(), J, kind)
        p_o, r_o = co.fk(f64(rot), f64(root), f64(off), parents)
        assert np.abs(rm - r_o).max() <= max(2e-6, 2.5e-7 * depth)
        assert np.abs(pos - p_o).max() <= max(pos_bar, 2 * _ulp_of(p_o)), (F, osc, np.abs(pos - p_o).max() / _ulp_of(p_o))
        np.testing.assert_array_equal(pos[:, 0].astype(np.float32), root)  # the root is the caller's value (skeleton.py:49)
        if F > 20_000 and J != 52:
            continue  # (the variants below at clip sizes and on SMPL-H's big batch)
        offs = (off[None] * np.linspace(0.7, 1.3, F, dtype=np.float32)[:, None, None]).astype(np.float32)  # per-frame offsets
        pos, rm = sk.fk(rot, root, offs, parents)
        assert "fk_pipe_kernel" in _lib.last_kernel_name()
        p_o, r_o = co.fk(f64(rot), f64(root), f64(offs), parents)
        assert np.abs(rm - r_o).max() <= max(2e-6, 2.5e-7 * depth) and np.abs(pos - p_o).max() <= max(pos_bar, 2 * _ulp_of(p_o))
        q_o = co.o6d_to_quat(f64(x6))
        p_o, r_o = co.fk(q_o, f64(root), f64(off), parents)
        for want_q in (False, True):
            out = sk.fk_from_ortho6d(x6, root, off, parents, return_quat=want_q)
            assert "fk_pipe_kernel" in _lib.last_kernel_name() or J < 30  # (the ortho6d source takes this walk from 30 joints on)
            assert np.abs(out[1] - r_o).max() <= 1e-5 and np.abs(out[0] - p_o).max() <= max(pos_bar, 2e-5, 2 * _ulp_of(p_o))
            if want_q:
                assert np.minimum(np.abs(out[2] - q_o).max(-1), np.abs(out[2] + q_o).max(-1)).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("J,kind", [(52, "smplh"), (48, "random"), (64, "humanoid"), (80, "random"), (96, "humanoid")])
@pytest.mark.parametrize("nt", ["1", "2", "3"])
def test_fk_four_joints_a_step_equals_the_twelve_lane_walk_to_the_bit(J, kind, nt, monkeypatch):
    """same local rotations, same products in the same order: PM_FK_W4 = 1 and 0 on the tuning build agree bit for bit -- quaternion source,
    per-frame offsets, fused ortho6d with quaternions, centimetre data (the fixed-point chain), partial last tiles and groups"""
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    monkeypatch.setenv("PM_FK_NT", nt)
    monkeypatch.setenv("PM_FK_FPW", "4")
    monkeypatch.setenv("PM_FK_WIDE", "0")
    monkeypatch.setenv("PM_FK_STREAM", "0")
    with _lib.variant("tuning"):
        for F, osc, rsc in ((4 * 9 + 1, 0.15, 2.0), (4 * 6 + 3, 20.0, 150.0), (2, 0.15, 2.0)):
            rot, root, off, x6 = _data(F, J, 7 * J + F, osc, rsc)
            offs = (off[None] * np.linspace(0.7, 1.3, F, dtype=np.float32)[:, None, None]).astype(np.float32)
            res = {}
            for w4 in ("0", "1"):
                monkeypatch.setenv("PM_FK_W4", w4)
                res[w4] = [*sk.fk(rot, root, off, parents), *sk.fk(rot, root, offs, parents), *sk.fk_from_ortho6d(x6, root, off, parents, return_quat=True)]
                assert "fk_pipe_kernel" in _lib.last_kernel_name()
            for a, b in zip(res["0"], res["1"]):
                np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_fk_chains_keep_the_one_joint_at_a_time_walk():
    """a 48-joint tree of two chains needs 24 steps of four: more than 0.32 J -- the dispatch leaves it where it was, and both give the oracle's answer"""
    import pymotion_amd.ops.skeleton as sk

    J, F = 48, 333
    parents = _tree("chain", J)
    rot, root, off, _ = _data(F, J, 5, 0.15, 2.0)
    buf = (C.c_uint32 * (50 * 16))()
    n16 = _lib.lib().pm_fk_wide_plan_debug(parents.ctypes.data_as(C.c_void_p), J, buf)
    assert n16 >= 23  # (depth 24: a step per level whatever the width)
    pos, rm = sk.fk(rot, root, off, parents)
    p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)
    assert np.abs(pos - p_o).max() <= 1e-5 and np.abs(rm - r_o).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("J", [8, 9, 10, 11, 14, 16, 18, 21, 22, 23, 26, 27, 29])
def test_fk_sixteen_frame_walk_gives_the_same_bits_whichever_frames_share_a_half_wave(J, monkeypatch):
    """round 6: quad q of the sixteen-frame tile walks frame fmap[q] (fk.hip: q4_frame_map -- the split of the frames over the two half-waves with
    the fewest LDS bank conflicts, per joint count).  Which quad walks a frame changes nothing but the banks: every one of the five candidate
    splits (PM_FK_FMAP on the tuning build) and the production library's own pick give the block split's results bit for bit, on metre and
    centimetre data (the fixed-point chain), full and partial tiles -- and the oracle's within the usual bar"""
    import pymotion_amd.ops.skeleton as sk

    # (from 24 joints on a wide tree takes the four-frame kernel: chains keep the sixteen-frame tile up to 29)
    parents = syn.PARENTS_22 if J == 22 else (_tree("chain", J) if J >= 24 else syn.random_parents(J, np.random.default_rng(J)))
    for F, osc, rsc in ((16 * 7 + 5, 0.15, 2.0), (16 * 3 + 15, 25.0, 150.0), (3, 0.15, 2.0)):
        rot, root, off, _ = _data(F, J, 11 * J + F, osc, rsc)
        got = {}
        with _lib.variant("tuning"):
            for split in ("0", "1", "2", "3", "4"):
                monkeypatch.setenv("PM_FK_FMAP", split)
                got[split] = sk.fk(rot, root, off, parents)
                assert "fk_kernel<16" in _lib.last_kernel_name(), _lib.last_kernel_name()
            monkeypatch.delenv("PM_FK_FMAP")
        got["prod"] = sk.fk(rot, root, off, parents)
        assert "fk_kernel<16" in _lib.last_kernel_name(), _lib.last_kernel_name()
        for k in ("1", "2", "3", "4", "prod"):
            np.testing.assert_array_equal(got[k][0], got["0"][0])
            np.testing.assert_array_equal(got[k][1], got["0"][1])
        p_o, r_o = co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)
        assert np.abs(got["prod"][1] - r_o).max() <= 1e-5 and np.abs(got["prod"][0] - p_o).max() <= max(1e-5, 2 * _ulp_of(p_o))
