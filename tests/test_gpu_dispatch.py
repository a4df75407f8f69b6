"""The PRODUCTION library's choice between the tile kernels and the long-skeleton kernels (one lane per frame; fk's streamed walk) by the
joint-frames of the call (common.hpp: lane_per_frame_pays): a clip of real length stays on the tile kernels, a batch big enough to fill
the chip takes the long-skeleton kernel.  Every kernel is tied to the float64 ORACLE here, on libpmhip.so itself: three 256-frame
slices (first, middle and last tiles) of the big production call against oracle/c_oracle.py at the bars of the test-size suites
(test_gpu_deep.py, test_ik.py, test_gpu_parity.py -- those run the same kernels on the -DPM_TUNING compilation with the threshold at 0).
`test_every_kernel_name_of_the_production_library_is_tied_to_the_oracle` lists the kernel templates the skeleton dispatchers can
return and checks that the oracle-tied calls of this file and of the listed suites saw each of them."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib

pytestmark = pytest.mark.gpu
SEEN = set()     # kernel template names the oracle-tied calls of this module dispatched to (production library)


def _ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def _slices(F, n=256):
    return [slice(0, n), slice((F // 2) - n // 2, (F // 2) + n // 2), slice(F - n, F)]


def _note():
    import re

    name = _lib.last_kernel_name()
    SEEN.add(re.search(r"pm::(\w+_kernel)", name).group(1))
    return name


def _f64(*a):
    return [x.astype(np.float64) for x in a]


def _chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    p[J // 2] = 0
    p[3 * J // 4] = J // 4
    return p


def _batch(F, J, seed):
    rng = np.random.default_rng(seed)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = rng.uniform(-0.3, 0.3, (J, 3)).astype(np.float32)
    off[0] = 0
    return rot, root, off


def test_to_root_dual_quat_by_joint_frames():
    """centimetre-scale bones on a skeleton deep enough for the float64 bone rotation (the front door's hint says so): the lane-per-frame kernel when
    the call fills the chip, the step-list kernel (dqwide.hip) on a clip of real length"""
    import pymotion_amd.ops.skeleton as sk

    J, par = 64, _chain_like(64)
    rot, root, off = _batch(40_000, J, 1)       # 2.56 M joint-frames: above the 2.4 M of deep.hip's kernels
    root, off = root * 100.0, off * 100.0
    big = sk.to_root_dual_quat(rot, root, par, off)
    assert "to_root_dq_deep_kernel" in _note(), _lib.last_kernel_name()
    small = sk.to_root_dual_quat(rot[:4096], root[:4096], par, off)
    assert "to_root_dq_wide_kernel" in _note(), _lib.last_kernel_name()
    d_o = co.to_root_dual_quat(*_f64(rot[:4096], root[:4096]), par, off.astype(np.float64))
    assert np.abs(big[:4096] - small).max() <= 3 * _ulp_of(d_o)
    assert np.abs(small - d_o).max() <= max(1e-5, 3 * _ulp_of(d_o)), np.abs(small - d_o).max() / _ulp_of(d_o)
    for sl in _slices(len(rot)):
        d_o = co.to_root_dual_quat(*_f64(rot[sl], root[sl]), par, off.astype(np.float64))
        # float64 state (deep.hip): the oracle's value rounded once -- test_gpu_deep.py's bar
        assert np.abs(big[sl] - d_o).max() <= _ulp_of(d_o), (sl, np.abs(big[sl] - d_o).max() / _ulp_of(d_o))
        assert np.abs(big[sl][..., :4] - d_o[..., :4]).max() <= 6.1e-8


def test_to_root_dual_quat_step_list_kernel_against_the_oracle():
    """dqwide.hip on the production library: a wide 256-joint tree (a frame a wave, sixteen joints a step) and SMPL-H (eight / four frames a wave), metre-scale
    bones (the fp32 step) and centimetre-scale ones (the precise step: 2 ulp of the largest dual component, tests/test_gpu_large_magnitude.py's bar),
    first / middle / last tiles of a call big enough for several tiles a workgroup"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import synthetic as syn

    for J, par, F in ((256, syn.random_parents(256, np.random.default_rng(256)), 140_000), (52, np.asarray(syn.PARENTS_52, dtype=np.int32), 140_001)):
        rot, root, off = _batch(F, J, 31 + J)
        for scale in (1.0, 100.0):
            d = sk.to_root_dual_quat(rot, root * scale, par, off * scale)
            # (centimetre-scale bones by the front door's hint on a tree the lane-per-frame kernels take -- SMPL-H -- go to them first where a call has their
            # 2.4 M joint-frames; the NumPy door's pipeline chunks of this call do not)
            assert "to_root_dq_wide_kernel" in _note(), _lib.last_kernel_name()
            for sl in _slices(F):
                d_o = co.to_root_dual_quat(*_f64(rot[sl], root[sl] * np.float32(scale)), par, (off * np.float32(scale)).astype(np.float64))
                assert np.abs(d[sl] - d_o).max() <= max(1e-5, 2 * _ulp_of(d_o)), (J, scale, sl, np.abs(d[sl] - d_o).max() / _ulp_of(d_o))
                assert np.abs(d[sl][..., :4] - d_o[..., :4]).max() <= (2e-6 if scale == 1.0 else 1.2e-7)


def test_to_root_dual_quat_ring_kernel_by_joint_frames():
    """a joint count that is not a multiple of eight: the line-aligned ring kernel, on the production library"""
    import pymotion_amd.ops.skeleton as sk

    J, par = 130, _chain_like(130)
    rot, root, off = _batch(19_000, J, 11)      # 2.47 M joint-frames
    big = sk.to_root_dual_quat(rot, root, par, off)
    assert "to_root_dq_ring_kernel" in _note(), _lib.last_kernel_name()
    for sl in _slices(len(rot)):
        d_o = co.to_root_dual_quat(*_f64(rot[sl], root[sl]), par, off.astype(np.float64))
        assert np.abs(big[sl] - d_o).max() <= _ulp_of(d_o), (sl, np.abs(big[sl] - d_o).max() / _ulp_of(d_o))
        assert np.abs(big[sl][..., :4] - d_o[..., :4]).max() <= 6.1e-8


def _mirror_want(rot, off, par, sl):
    # the reference's chain rebuilt from oracle pieces (fk -> from_matrix -> negate -> from_global_rotations; axis 0 = X: components 2, 3,
    # skeleton.py:310-312)
    _, rm = co.fk(rot[sl].astype(np.float64), np.zeros((len(rot[sl]), 3)), off.astype(np.float64), par)
    g = co.quat_from_matrix(rm)
    g[..., 2] *= -1
    g[..., 3] *= -1
    return co.from_global_rotations(g, par)


def test_mirror_by_joint_frames():
    """a chain-like skeleton deeper than the step list of mirror_wide_kernel holds: the lane-per-frame kernel when the call fills the chip, the tile kernels
    on a clip of real length"""
    J, par = 130, _chain_like(130)
    rot, root, off = _batch(50_000, J, 2)       # 6.5 M joint-frames
    import torch

    dev = torch.device("cuda:0")
    tr = torch.from_numpy(rot).to(dev)
    out = torch.empty((rot.shape[0], J, 4), device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.call("pm_mirror_rotations_f32", P(tr), par.ctypes.data_as(C.c_void_p), None, 0, rot.shape[0], J, P(out), None)
    assert "mirror_deep_kernel" in _note(), _lib.last_kernel_name()
    out_s = torch.empty((4096, J, 4), device=dev)
    _lib.call("pm_mirror_rotations_f32", P(tr), par.ctypes.data_as(C.c_void_p), None, 0, 4096, J, P(out_s), None)
    assert "mirror_kernel<" in _note(), _lib.last_kernel_name()
    a, b = out[:4096].cpu().numpy(), out_s.cpu().numpy()
    assert np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max() <= 4e-6
    # up to the sign of each quaternion: test_gpu_parity.py's bar for the big skeletons
    big = out.cpu().numpy()
    for got, sl in [(big[sl], sl) for sl in _slices(rot.shape[0])] + [(b, slice(0, 4096, 16))]:
        if got.shape[0] != 256:
            got = got[::16]
        want = _mirror_want(rot, off, par, sl)
        err = np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max()
        assert err <= 1e-5 * max(1.0, 65 / 32), (sl, err)


def test_mirror_step_list_kernel_against_the_oracle():
    """mirror_wide_kernel on the production library: SMPL-H and a random 96-joint tree (four frames a wave; one and two tiles a workgroup) and a wide 300-joint
    tree (one frame a wave, sixteen joints a step), first / middle / last tiles of a production-size call, up to the sign of each quaternion"""
    import torch

    from pymotion_amd import synthetic as syn

    dev = torch.device("cuda:0")
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for J, par, F, fpw in ((52, np.asarray(syn.PARENTS_52, dtype=np.int32), 70_001, 4), (96, syn.random_parents(96, np.random.default_rng(96)).astype(np.int32), 140_000, 4),
                           (300, syn.random_parents(300, np.random.default_rng(300)).astype(np.int32), 70_000, 1)):
        rot, root, off = _batch(F, J, 40 + J)
        tr = torch.from_numpy(rot).to(dev)
        out = torch.full((F, J, 4), float("nan"), device=dev)
        _lib.call("pm_mirror_rotations_f32", P(tr), par.ctypes.data_as(C.c_void_p), None, 0, F, J, P(out), None)
        assert "mirror_wide_kernel<%d," % fpw in _note(), _lib.last_kernel_name()
        got = out.cpu().numpy()
        for sl in _slices(F):
            want = _mirror_want(rot, off, par, sl)
            err = np.minimum(np.abs(got[sl] - want).max(-1), np.abs(got[sl] + want).max(-1)).max()
            assert err <= 1e-5, (J, sl, err)


def test_from_root_positions_by_joint_frames():
    import pymotion_amd.ops.skeleton as sk
    from oracle import c_oracle as co
    from pymotion_amd import synthetic as syn

    F = 60_000                                   # x 52 = 3.1 M joint-frames
    rot, root, off, par = syn.fk_workload(F, parents=syn.PARENTS_52, seed=3, normalized=True, offset_scale=0.15)
    pos, _ = sk.fk(rot, np.zeros_like(root), off, par)
    pos = pos.astype(np.float32)
    big = sk.from_root_positions(pos, par, off)
    assert "from_root_positions_order_kernel" in _note(), _lib.last_kernel_name()
    small = sk.from_root_positions(pos[:2048], par, off)
    assert "from_root_positions_kernel<" in _note(), _lib.last_kernel_name()
    err = np.minimum(np.abs(big[:2048] - small).max(-1), np.abs(big[:2048] + small).max(-1))
    assert np.median(err) <= 1e-6 and np.quantile(err, 0.999) <= 2e-5, (float(np.median(err)), float(np.quantile(err, 0.999)))
    # per record against the oracle, test_ik.py's bar: 2e-5 + 8 x how far one ulp of the inputs moves the reference's own answer
    from test_ik import _reference_sensitivity

    for got, sl in [(big[sl], sl) for sl in _slices(F)] + [(small[:256], slice(0, 256))]:
        ref = co.from_root_positions(pos[sl].astype(np.float64), par, off.astype(np.float64))
        e2 = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
        sens = _reference_sensitivity(pos[sl], par, off, ref, draws=12)
        assert (e2 <= 2e-5 + 8.0 * sens).all(), (sl, float(e2.max()), float(((e2 - 2e-5) / np.maximum(sens, 1e-12)).max()))
        assert np.median(e2) <= 1e-6 and np.quantile(e2, 0.999) <= 2e-5


def test_fk_long_skeleton_by_joint_frames():
    import pymotion_amd.ops.skeleton as sk

    J, par = 128, _chain_like(128)
    rot, root, off = _batch(8192, J, 4)          # 1.05 M joint-frames
    pos, rm = sk.fk(rot, root, off, par)
    assert "fk_stream_kernel" in _note(), _lib.last_kernel_name()
    pos_s, rm_s = sk.fk(rot[:512], root[:512], off, par)
    assert "fk_stream_kernel" not in _note(), _lib.last_kernel_name()
    assert np.abs(pos[:512] - pos_s).max() <= 1e-5 and np.abs(rm[:512] - rm_s).max() <= 1e-5
    depth = 64  # of _chain_like(128)
    for (p, r), sl in [((pos[sl], rm[sl]), sl) for sl in _slices(len(rot))] + [((pos_s[:256], rm_s[:256]), slice(0, 256))]:
        p_o, r_o = co.fk(*_f64(rot[sl], root[sl], off), par)
        # test_gpu_deep.py's bars: rotations an fp32 chain of `depth` products; positions 1e-5 or 3 ulp of the largest coordinate
        assert np.abs(r - r_o).max() <= max(2e-6, 2.5e-7 * depth), (sl, np.abs(r - r_o).max())
        assert np.abs(p - p_o).max() <= max(1e-5, 3 * _ulp_of(p_o)), (sl, np.abs(p - p_o).max())


def test_tile_kernels_against_the_oracle():
    """the kernels a clip of real length takes, on the production library, against the oracle (their full suites: test_gpu_parity.py)"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import synthetic as syn

    for par, F, want in ((syn.PARENTS_22, 4099, "fk_kernel"), (syn.PARENTS_52, 1031, "fk_pipe_kernel")):
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=F)
        pos, rm = sk.fk(rot, root, off, par)
        assert want in _note(), _lib.last_kernel_name()
        p_o, r_o = co.fk(*_f64(rot, root, off), par)
        assert np.abs(pos - p_o).max() <= 1e-5 and np.abs(rm - r_o).max() <= 1e-5
        rn = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
        d = sk.to_root_dual_quat(rn, root, par, off)
        assert "to_root_dq_wide_kernel" in _note(), _lib.last_kernel_name()
        d_o = co.to_root_dual_quat(*_f64(rn, root), par, off.astype(np.float64))
        assert np.abs(d - d_o).max() <= 1e-5
        t, q = sk.from_root_dual_quat(d_o.astype(np.float32), par)
        assert "gather_parent_kernel" in _note(), _lib.last_kernel_name()
        t_o, q_o = co.from_root_dual_quat(d_o.astype(np.float32).astype(np.float64), par)
        assert np.abs(t - t_o).max() <= 1e-5 and np.abs(q - q_o).max() <= 1e-5


def test_to_root_dual_quat_one_chain_tile_kernel_against_the_oracle():
    """a skeleton that is one chain: nothing to schedule onto a second chain -- to_root_dq_kernel"""
    import pymotion_amd.ops.skeleton as sk

    J = 12
    par = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    rot, root, off = _batch(3001, J, 21)
    d = sk.to_root_dual_quat(rot, root, par, off)
    assert "to_root_dq_kernel" in _note(), _lib.last_kernel_name()
    d_o = co.to_root_dual_quat(*_f64(rot, root), par, off.astype(np.float64))
    assert np.abs(d - d_o).max() <= 1e-5


def test_to_root_dual_quat_scheduled_walk_against_the_oracle():
    """a long, narrow tree (more steps than the step list of dqwide.hip holds) on a clip of real length (too few joint-frames for the lane-per-frame kernels):
    the scheduled walk of dq.hip, to_root_dq_sched_kernel"""
    import pymotion_amd.ops.skeleton as sk

    J, par = 130, _chain_like(130)
    rot, root, off = _batch(3001, J, 23)
    d = sk.to_root_dual_quat(rot, root, par, off)
    assert "to_root_dq_sched_kernel" in _note(), _lib.last_kernel_name()
    d_o = co.to_root_dual_quat(*_f64(rot, root), par, off.astype(np.float64))
    assert np.abs(d - d_o).max() <= 1e-5


def test_fk_wide_walk_on_bushy_trees():
    """beyond 128 joints a tree the streamed walk declines takes a wave per frame with its lanes over the joints (fkwide.hip), at any
    batch size; tied to the oracle here on slices of a production-size call (the full suite: tests/test_gpu_wide.py)"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import synthetic as syn

    J = 256
    par = syn.random_parents(J, np.random.default_rng(J))
    depth = int(syn.depth_of(par).max())
    rot, root, off = _batch(6_000, J, 21)
    pos, rm = sk.fk(rot, root, off, par)
    assert "fk_wide_kernel" in _note(), _lib.last_kernel_name()
    for sl in _slices(len(rot)):
        p_o, r_o = co.fk(*_f64(rot[sl], root[sl], off), par)
        assert np.abs(rm[sl] - r_o).max() <= max(2e-6, 2.5e-7 * depth)
        assert np.abs(pos[sl] - p_o).max() <= max(1e-5, 3 * _ulp_of(p_o))


ALL_SKELETON_KERNELS = {"fk_wide_kernel", "to_root_dq_kernel", "to_root_dq_sched_kernel", "to_root_dq_wide_kernel", "to_root_dq_deep_kernel", "to_root_dq_ring_kernel", "gather_parent_kernel",
                        "fk_kernel", "fk_pipe_kernel", "fk_stream_kernel", "mirror_kernel", "mirror_deep_kernel", "mirror_wide_kernel", "from_root_positions_kernel",
                        "from_root_positions_order_kernel"}


def test_every_kernel_name_of_the_production_library_is_tied_to_the_oracle(request):
    """every kernel template `pm_last_kernel_name()` can return from libpmhip.so (the `set_kernel_name` sites of csrc/*.hip; asserted against
    the sources in tests/test_abi.py) was dispatched to by an oracle-tied call of this module -- on the production library"""
    ran = {i.name for i in request.session.items if i.module is request.module}
    if len(ran) < 12:
        pytest.skip("needs the whole module (the other tests collect the kernel names)")
    assert _lib.lib() is _lib._handles.get("prod"), "this module must run on the production library"
    assert SEEN == ALL_SKELETON_KERNELS, (sorted(ALL_SKELETON_KERNELS - SEEN), sorted(SEEN - ALL_SKELETON_KERNELS))
