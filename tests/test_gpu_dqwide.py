"""to_root_dual_quat's step-list kernel (pymotion_amd/csrc/dqwide.hip; reference: pymotion/ops/skeleton.py:207-244): 1 / 2 / 4 / 8 frames a wave and
16 / 8 / 4 / 2 joints of a frame a step, the steps from a host-made list held in registers.  Against the float64 oracle at every tile shape (single frames,
partial tiles, several tiles a workgroup, partial last groups), metre-scale bones (the fp32 step) and centimetre-scale ones (the precise step: float64
quaternion chain re-read as head + 8-bit residual, fixed-point translations); bit for bit against the scheduled walk of dq.hip on metre-scale data (same
products in the same order); NaN / Inf where the reference has them -- a non-finite ROOT quaternion included, which this kernel copies like the reference does
(skeleton.py:223: `rotations.copy()`, the root is never multiplied) while the walks of dq.hip run it through identity (x) q_0 --; and which instance the
production dispatch picks.  The forced instances run on the -DPM_TUNING compilation (PM_DQ_WIDE = frames a wave): same kernels as libpmhip.so."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from test_gpu_deep import chain_like, humanoid_with_hands

pytestmark = pytest.mark.gpu


def _ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def _tree(kind, J):
    if kind == "body":
        return np.asarray(syn.PARENTS_22, dtype=np.int32)
    if kind == "smplh":
        return np.asarray(syn.PARENTS_52, dtype=np.int32)
    if kind == "chain":
        return chain_like(J)
    if kind == "humanoid":
        return humanoid_with_hands(J)
    if kind == "star":
        return np.zeros(J, dtype=np.int32)
    return syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)


def _batch(F, J, seed, osc, rsc):
    rng = np.random.default_rng(seed)
    rot = rng.standard_normal((F, J, 4))
    rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
    root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
    off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
    off[0] = 0
    return rot, root, off


def _oracle(rot, root, parents, off):
    with np.errstate(all="ignore"):
        return co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))


SHAPES = [(16, "bushy"), (17, "star"), (22, "body"), (31, "chain"), (52, "smplh"), (64, "humanoid"), (65, "bushy"), (100, "bushy"), (128, "bushy"),
          (129, "humanoid"), (250, "bushy"), (300, "humanoid"), (512, "bushy")]


@pytest.mark.parametrize("fpw", [1, 2, 4, 8])
@pytest.mark.parametrize("J,kind", SHAPES)
def test_step_list_kernel_against_the_oracle(J, kind, fpw, monkeypatch):
    import pymotion_amd.ops.skeleton as sk

    if fpw * J > 512:
        pytest.skip("a tile is at most eight batches of 64 records")
    parents = _tree(kind, J)
    depth = int(syn.depth_of(parents).max())
    monkeypatch.setenv("PM_DQ_WIDE", str(fpw))
    with _lib.variant("tuning"):
        # (frames, tiles a workgroup, offsets scale, root scale): single frames, partial tiles, a partial last group of three tiles, centimetre-scale data
        for F, nt, osc, rsc in ((1, 1, 0.3, 2.0), (fpw, 1, 30.0, 200.0), (fpw + 1, 2, 0.3, 2.0), (65, 1, 30.0, 200.0), (7 * fpw + 3, 3, 0.3, 2.0),
                                (1000, 3, 30.0, 200.0), (2049, 4, 0.3, 2.0)):
            monkeypatch.setenv("PM_DQW_NT", str(nt))
            rot, root, off = _batch(F, J, 100 * J + F, osc, rsc)
            d = sk.to_root_dual_quat(rot, root, parents, off)
            name = _lib.last_kernel_name()
            assert "to_root_dq_wide_kernel<%d," % fpw in name, name
            d_o = _oracle(rot, root, parents, off)
            err = np.abs(d - d_o).max()
            assert err <= max(1e-5, 3 * _ulp_of(d_o)), (F, nt, osc, err / _ulp_of(d_o), "ulp")
            assert np.abs(d[..., :4] - d_o[..., :4]).max() <= max(2e-6, 2.5e-7 * depth)
            if osc >= 1.0:  # the precise step: the quaternions are a float64 chain rounded ONCE (2^-25 below 1) -- a parent re-read without its residual reads 5-6e-8
                assert np.abs(d[..., :4] - d_o[..., :4]).max() <= 3.2e-8, np.abs(d[..., :4] - d_o[..., :4]).max()
            t, q = sk.from_root_dual_quat(d, parents)
            assert np.abs(q - rot).max() <= 4e-6
            assert np.abs(t[:, 1:] - off[1:]).max() <= 4e-6 * max(1.0, np.abs(d_o).max())


@pytest.mark.parametrize("J,kind", [(22, "body"), (52, "smplh"), (100, "bushy"), (200, "bushy"), (64, "chain")])
def test_step_list_kernel_gives_the_scheduled_walks_bits_on_metre_scale_data(J, kind, monkeypatch):
    """same products in the same order (dq_step_math), the same phase C: the fp32 step's results are the scheduled walk's to the bit"""
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    rot, root, off = _batch(3001, J, J, 0.3, 2.0)
    with _lib.variant("tuning"):
        monkeypatch.setenv("PM_DQ_WIDE", "0")
        monkeypatch.setenv("PM_DQ_DEEP", "0")
        ref = sk.to_root_dual_quat(rot, root, parents, off)
        assert "to_root_dq_sched_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
        for fpw in (1, 2, 4, 8):
            if fpw * J > 512:
                continue
            monkeypatch.setenv("PM_DQ_WIDE", str(fpw))
            d = sk.to_root_dual_quat(rot, root, parents, off)
            if "wide_kernel" not in _lib.last_kernel_name():
                continue
            np.testing.assert_array_equal(d.view(np.int32), ref.view(np.int32), err_msg=f"fpw {fpw}")


@pytest.mark.parametrize("fpw", [1, 2, 4, 8])
def test_step_list_kernel_keeps_nan_and_inf_where_the_reference_has_them(fpw, monkeypatch):
    import pymotion_amd.ops.skeleton as sk

    J = 52
    parents = _tree("smplh", J)
    dep = syn.depth_of(parents)
    F = 400
    for osc, rsc in ((0.3, 2.0), (30.0, 200.0)):
        rot, root, off = _batch(F, J, 7, osc, rsc)
        inner = [j for j in range(J) if dep[j] >= 2]
        rot[70, inner[3], 2] = np.nan          # a joint below the root's children: its subtree goes NaN
        rot[71, inner[10], 0] = np.inf
        root[150, 1] = np.nan                  # a root position: the root's dual part only (its children stay local, skeleton.py:236-237)
        root[151, 2] = np.inf
        rot[230, 0] = (np.inf, 0.0, 0.0, 0.0)  # the ROOT's quaternion: copied (skeleton.py:223), the root's record only
        rot[231, 0, 3] = np.nan
        d_o = _oracle(rot, root, parents, off)
        monkeypatch.setenv("PM_DQ_WIDE", str(fpw))
        with _lib.variant("tuning"):
            d = sk.to_root_dual_quat(rot, root, parents, off)
            assert "to_root_dq_wide_kernel<%d," % fpw in _lib.last_kernel_name(), _lib.last_kernel_name()
        assert (np.isnan(d) == np.isnan(d_o)).all(), np.argwhere(np.isnan(d) != np.isnan(d_o))[:5]
        assert (np.isinf(d) == np.isinf(d_o)).all()
        np.testing.assert_array_equal(np.sign(d[np.isinf(d_o)]), np.sign(d_o[np.isinf(d_o)]))
        fin = np.isfinite(d_o)
        assert np.abs(d[fin] - d_o[fin]).max() <= max(1e-5, 3 * _ulp_of(d_o[fin]))
        clean = np.ones(F, bool)
        clean[[70, 71, 150, 151, 230, 231]] = False
        assert np.isfinite(d[clean]).all()     # a frame's NaN stays in its frame, whatever shares its wave


PICKS = [  # (J, tree, offsets scale, frames a wave the production dispatch picks -- None: another kernel)
    (12, "bushy", 0.3, None),       # below kDqWideMinJ: sixteen frames a wave on the one-chain kernel
    (16, "bushy", 0.3, 8), (22, "body", 0.3, 8), (22, "body", 30.0, 8), (33, "bushy", 0.3, 4), (52, "smplh", 0.15, 4), (52, "smplh", 30.0, 4), (64, "chain", 0.3, 8),   # (a narrow tree: two thirds of the quad-steps busy only at two joints a step)
   
    (100, "bushy", 0.3, 4), (128, "bushy", 30.0, 4), (129, "humanoid", 0.3, 2), (200, "bushy", 0.3, 1), (512, "bushy", 30.0, 1),
    (96, "chain", 0.3, 4),          # 48 levels: the list holds them at four joints a step, and over a third of the quad-steps are busy
    (28, "chain", 30.0, 8),         # centimetre-scale bones (the front door's hint says so) on a skeleton deep enough for the float64 bone rotation, but too few
                                    # joint-frames for the lane-per-frame kernels: they decline, this one takes it
    (130, "chain", 0.3, None),      # 65 levels: more steps than the list holds at any width
]


@pytest.mark.parametrize("J,kind,osc,fpw", PICKS)
def test_production_dispatch_of_the_step_list_kernel(J, kind, osc, fpw):
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    rot, root, off = _batch(257, J, 5 * J, osc, 2.0 if osc < 1 else 200.0)
    assert _lib.lib() is _lib._handles.get("prod")
    d = sk.to_root_dual_quat(rot, root, parents, off)
    name = _lib.last_kernel_name()
    if fpw is None:
        assert "wide_kernel" not in name, name
    else:
        assert "to_root_dq_wide_kernel<%d," % fpw in name, name
    d_o = _oracle(rot, root, parents, off)
    depth = int(syn.depth_of(parents).max())
    assert np.abs(d - d_o).max() <= max(1e-5, 3 * _ulp_of(d_o) * max(1.0, depth / 64.0))


def test_raw_abi_without_a_scale_hint():
    """pm_to_root_dq_f32 cannot see the bones' scale and is tuned for metre-scale data: the step-list kernel first, whatever the depth (a caller who knows its bones are
    big says so through pm_to_root_dq_hint_f32 and keeps the lane-per-frame kernels on the calls they take)"""
    import torch

    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for J, kind, want in ((22, "body", "to_root_dq_wide_kernel<8,"), (52, "smplh", "to_root_dq_wide_kernel<4,"), (200, "bushy", "to_root_dq_wide_kernel<1,"),
                          (64, "chain", "to_root_dq_wide_kernel<8,")):
        parents = _tree(kind, J)
        rot, root, off = _batch(300, J, J, 0.3, 2.0)
        tr, tp, to = (torch.from_numpy(x).cuda() for x in (rot, root, off))
        out = torch.empty((300, J, 8), device="cuda")
        _lib.call("pm_to_root_dq_f32", P(tr), P(tp), parents.ctypes.data_as(C.c_void_p), P(to), 300, J, P(out), None)
        assert want in _lib.last_kernel_name(), _lib.last_kernel_name()
        assert np.abs(out.cpu().numpy() - _oracle(rot, root, parents, off)).max() <= 1e-5
