"""The PRODUCTION library's choice between the tile kernels and the long-skeleton kernels (one lane per frame; fk's streamed walk) by the
joint-frames of the call (common.hpp: lane_per_frame_pays): a clip of real length stays on the tile kernels, a batch big enough to fill
the chip takes the long-skeleton kernel -- and the two agree with each other on the frames they share (each is tied to the float64 oracle
at test sizes elsewhere: test_gpu_deep.py, test_ik.py, test_gpu_parity.py)."""
import numpy as np
import pytest

from pymotion_amd import _lib

pytestmark = pytest.mark.gpu


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
    import pymotion_amd.ops.skeleton as sk

    J, par = 64, _chain_like(64)
    rot, root, off = _batch(40_000, J, 1)       # 2.56 M joint-frames: above the 2.4 M of deep.hip's kernels
    big = sk.to_root_dual_quat(rot, root, par, off)
    assert "to_root_dq_deep_kernel" in _lib.last_kernel_name() or "to_root_dq_ring_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    small = sk.to_root_dual_quat(rot[:4096], root[:4096], par, off)
    assert "to_root_dq_sched_kernel" in _lib.last_kernel_name() or "to_root_dq_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    assert np.abs(big[:4096] - small).max() <= 1e-5


def test_mirror_by_joint_frames():
    J, par = 72, _chain_like(72)
    rot, root, off = _batch(90_000, J, 2)       # 6.5 M joint-frames
    import ctypes as C

    import torch

    dev = torch.device("cuda:0")
    tr = torch.from_numpy(rot).to(dev)
    out = torch.empty((rot.shape[0], J, 4), device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.call("pm_mirror_rotations_f32", P(tr), par.ctypes.data_as(C.c_void_p), None, 0, rot.shape[0], J, P(out), None)
    assert "mirror_deep_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    out_s = torch.empty((4096, J, 4), device=dev)
    _lib.call("pm_mirror_rotations_f32", P(tr), par.ctypes.data_as(C.c_void_p), None, 0, 4096, J, P(out_s), None)
    assert "mirror_kernel<" in _lib.last_kernel_name(), _lib.last_kernel_name()
    a, b = out[:4096].cpu().numpy(), out_s.cpu().numpy()
    assert np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max() <= 4e-6


def test_from_root_positions_by_joint_frames():
    import pymotion_amd.ops.skeleton as sk
    from oracle import c_oracle as co
    from pymotion_amd import synthetic as syn

    F = 60_000                                   # x 52 = 3.1 M joint-frames
    rot, root, off, par = syn.fk_workload(F, parents=syn.PARENTS_52, seed=3, normalized=True, offset_scale=0.15)
    pos, _ = sk.fk(rot, np.zeros_like(root), off, par)
    big = sk.from_root_positions(pos, par, off)
    assert "from_root_positions_order_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    small = sk.from_root_positions(pos[:2048], par, off)
    assert "from_root_positions_kernel<" in _lib.last_kernel_name(), _lib.last_kernel_name()
    err = np.minimum(np.abs(big[:2048] - small).max(-1), np.abs(big[:2048] + small).max(-1))
    assert np.median(err) <= 1e-6 and np.quantile(err, 0.999) <= 2e-5, (float(np.median(err)), float(np.quantile(err, 0.999)))
    ref = co.from_root_positions(pos[59_000:59_256].astype(np.float64), par, off.astype(np.float64))
    e2 = np.minimum(np.abs(big[59_000:59_256] - ref).max(-1), np.abs(big[59_000:59_256] + ref).max(-1))
    assert np.median(e2) <= 1e-6 and np.quantile(e2, 0.999) <= 2e-5


def test_fk_long_skeleton_by_joint_frames():
    import pymotion_amd.ops.skeleton as sk

    J, par = 128, _chain_like(128)
    rot, root, off = _batch(8192, J, 4)          # 1.05 M joint-frames
    pos, rm = sk.fk(rot, root, off, par)
    assert "fk_stream_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    pos_s, rm_s = sk.fk(rot[:512], root[:512], off, par)
    assert "fk_stream_kernel" not in _lib.last_kernel_name(), _lib.last_kernel_name()
    assert np.abs(pos[:512] - pos_s).max() <= 1e-5 and np.abs(rm[:512] - rm_s).max() <= 1e-5
