"""GPU, BASELINE.json full sizes: parity on slices against the CPU oracle plus size-independent
properties over the WHOLE batch (orthonormality, root pinning, translation equivariance, offset
linearity, encode->decode round trips).  Device-resident tensors through the torch front door."""
import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _mods():
    import torch

    import pymotion_amd.ops.skeleton_torch as skt
    import pymotion_amd.rotations.ortho6d_torch as o6t
    import pymotion_amd.rotations.quat_torch as qt

    return torch, skt, qt, o6t


def _dev_workload(torch, F, parents, seed, normalized=False, scale=0.3):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    J = len(parents)
    rot = torch.randn((F, J, 4), generator=g, device="cuda")
    if normalized:
        rot = rot / rot.norm(dim=-1, keepdim=True)
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(seed), scale)).cuda()
    return rot, root, off, torch.from_numpy(np.asarray(parents))


def _ortho_err(m):
    """max |R R^T - I| with element-wise ops only (no batched GEMM on 23 M tiny matrices)."""
    worst = 0.0
    for i in range(3):
        for k in range(i, 3):
            d = (m[..., i, :] * m[..., k, :]).sum(-1) - (1.0 if i == k else 0.0)
            worst = max(worst, float(d.abs().max()))
    return worst


def _det(m):
    return (m[..., 0, 0] * (m[..., 1, 1] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 1])
            - m[..., 0, 1] * (m[..., 1, 0] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 0])
            + m[..., 0, 2] * (m[..., 1, 0] * m[..., 2, 1] - m[..., 1, 1] * m[..., 2, 0]))


def _slices(F, n=2048):
    return [slice(0, n), slice(F // 2 - 7, F // 2 - 7 + n), slice(F - n, F)]


def test_config2_fk_1m_frames_22_joints():
    torch, skt, qt, _ = _mods()
    F = 1 << 20
    rot, root, off, par = _dev_workload(torch, F, syn.PARENTS_22, 0)
    pos, rm = skt.fk(rot, root, off, par)
    assert pos.shape == (F, 22, 3) and rm.shape == (F, 22, 3, 3) and pos.dtype == torch.float32
    for s in _slices(F):
        p_o, r_o = co.fk(rot[s].cpu().numpy().astype(np.float64), root[s].cpu().numpy().astype(np.float64),
                         off.cpu().numpy().astype(np.float64), par.numpy())
        assert np.abs(pos[s].cpu().numpy() - p_o).max() <= ATOL
        assert np.abs(rm[s].cpu().numpy() - r_o).max() <= ATOL
    # properties over all 2^20 frames
    assert _ortho_err(rm) < 1e-5  # world rotations are orthonormal
    assert float((_det(rm) - 1).abs().max()) < 1e-5
    assert torch.equal(pos[:, 0, :], root)  # root joint sits exactly at global_pos (skeleton.py:49)
    bone = (pos[:, 1:, :] - pos[:, par[1:].cuda(), :]).norm(dim=-1)  # rigid bones: |p_j - p_parent| = |offset_j|
    assert float((bone - off[1:].norm(dim=-1)).abs().max()) < 1e-5
    # translation equivariance and offset linearity
    d = torch.tensor([3.0, -1.5, 0.25], device="cuda")
    pos2, rm2 = skt.fk(rot, root + d, off, par)
    assert torch.equal(rm2, rm) and float((pos2 - pos - d).abs().max()) < 2e-6
    pos3, _ = skt.fk(rot, root, off * 2, par)
    assert float(((pos3 - root[:, None, :]) - 2 * (pos - root[:, None, :])).abs().max()) < 4e-6
    # scale invariance of the internal normalisation (quat.py:423): fk(3 q) == fk(q)
    pos4, rm4 = skt.fk(rot * 3, root, off, par)
    assert float((pos4 - pos).abs().max()) < 5e-6 and float((rm4 - rm).abs().max()) < 5e-6
    # standalone quat.to_matrix on the normalised [F*22, 4] (second half of config 2)
    qn = qt.normalize(rot.view(-1, 4))
    m = qt.to_matrix(qn)
    assert _ortho_err(m) < 1e-5
    s = slice(12345, 12345 + 4096)
    assert np.abs(m[s].cpu().numpy() - co.quat_to_matrix(qn[s].cpu().numpy().astype(np.float64))).max() <= ATOL
    # local joint 0 rotation is the root's world rotation
    assert float((m.view(F, 22, 3, 3)[:, 0] - rm[:, 0]).abs().max()) < 2e-6


def test_config3_dual_quat_round_trip_1m_frames():
    torch, skt, _, _ = _mods()
    F = 1 << 20
    rot, root, off, par = _dev_workload(torch, F, syn.PARENTS_22, 1, normalized=True)
    dq = skt.to_root_dual_quat(rot, root, par, off)
    assert dq.shape == (F, 22, 8)
    for s in _slices(F, 1024):
        d_o = co.to_root_dual_quat(rot[s].cpu().numpy().astype(np.float64), root[s].cpu().numpy().astype(np.float64),
                                   par.numpy(), off.cpu().numpy().astype(np.float64))
        assert np.abs(dq[s].cpu().numpy() - d_o).max() <= ATOL
    # unit dual quaternion: |qr| = 1 and qr . qd = 0
    qr, qd = dq[..., :4], dq[..., 4:]
    assert float((qr.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((qr * qd).sum(-1).abs().max()) < 1e-5
    t, q = skt.from_root_dual_quat(dq, par)  # (translations, rotations): skeleton.py:204
    assert float((q - rot).abs().max()) <= ATOL
    assert float((t[:, 1:, :] - off[1:]).abs().max()) <= ATOL
    assert float((t[:, 0, :] - root).abs().max()) <= ATOL


def test_config4_fused_ortho6d_fk_256k_frames_52_joints():
    torch, skt, _, o6t = _mods()
    F = 1 << 18
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    x = torch.randn((F, 52, 3, 2), generator=g, device="cuda")
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off = torch.from_numpy(syn.make_offsets(52, np.random.default_rng(4), 0.15)).cuda()
    par = torch.from_numpy(syn.PARENTS_52)
    pos, rm, q = skt.fk_from_ortho6d(x, root, off, par, return_quat=True)
    # north_star's bar on ALL frames, no conditioning mask: Gram-Schmidt is evaluated as c3 = (a x b) / |a x b|, c2 = c3 x c1
    # with Kahan's difference of products (common.hpp: o6d2m), whose error does not grow as the two columns approach
    # (anti-)parallel (round 2 compared at 5e-5 on the frames whose every joint had |cos| < 0.999).
    def up_to_sign(a_, b_):  # quat.from_matrix picks one of four branches: a record on a branch tie may come out negated
        return torch.minimum((a_ - b_).abs().amax(-1), (a_ + b_).abs().amax(-1))

    # (i) the reference chain on the GPU: ortho6d.to_quat -> fk (two launches, quaternions through HBM)
    q2 = o6t.to_quat(x)
    p2, r2 = skt.fk(q2, root, off, par)
    assert float(up_to_sign(q, q2).max()) < ATOL
    assert float(((q - q2).abs().amax(-1) > ATOL).float().mean()) < 1e-5  # sign flips: only ever on branch ties
    assert float((pos - p2).abs().max()) < ATOL and float((rm - r2).abs().max()) < ATOL
    p3, r3 = skt.fk_from_ortho6d(x, root, off, par)  # without the quaternion output: Gram-Schmidt matrix used directly
    assert float((p3 - pos).abs().max()) < ATOL and float((r3 - rm).abs().max()) < ATOL
    assert _ortho_err(r3) < 1e-5
    # (ii) the float64 oracle on slices spread over the batch, every frame of them, both variants
    for s0 in (0, 777, F // 2 + 13, F - 4096):
        sl = slice(s0, s0 + 4096)
        xs = x[sl].cpu().numpy().astype(np.float64)
        p_o, r_o, q_o = co.fk_from_ortho6d(xs, root[sl].cpu().numpy().astype(np.float64), off.cpu().numpy().astype(np.float64),
                                           par.numpy(), return_quat=True)
        for pp_, rr_, what in ((pos, rm, "with quaternions"), (p3, r3, "without")):
            assert np.abs(pp_[sl].cpu().numpy() - p_o).max() <= ATOL, what
            assert np.abs(rr_[sl].cpu().numpy() - r_o).max() <= ATOL, what
        qg = q[sl].cpu().numpy()
        assert np.minimum(np.abs(qg - q_o).max(-1), np.abs(qg + q_o).max(-1)).max() <= ATOL
        assert (np.abs(qg - q_o).max(-1) > ATOL).mean() < 1e-4
    assert _ortho_err(rm) < 2e-5
    assert torch.equal(pos[:, 0, :], root)
