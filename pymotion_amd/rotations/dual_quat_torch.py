"""Dual quaternions ``[..., 8] = [qr(4), qd(4)]`` -- drop-in for ``pymotion.rotations.dual_quat_torch``.

Reference: ``pymotion/rotations/dual_quat_torch.py``.  One gfx950 kernel per call, fp32 on the GPU.
``normalize`` / ``is_unit`` keep the reference's whole-batch branch (one host read of three device
counters where the reference's Python ``if`` synchronises).
"""
import torch

from .. import _backend, _ops


def _be():
    return _backend.torch_backend()


def from_rotation_translation(rotations: torch.Tensor, translations: torch.Tensor) -> torch.Tensor:
    """``dq = [q, 0.5 (0,t) (x) q]``.  Reference: dual_quat_torch.py:12-36."""
    return _ops.dq_from_rt(_be(), rotations, translations)


def from_translation(translations: torch.Tensor) -> torch.Tensor:
    """``[1,0,0,0, 0, t/2]``.  Reference: dual_quat_torch.py:39-61."""
    return _ops.dq_from_t(_be(), translations)


def to_rotation_translation(dq: torch.Tensor):
    """-> ``(rotations [..., 4], translations [..., 3])``, ``t = (2 qd (x) conj(qr))[1:]``.
    Reference: dual_quat_torch.py:64-85."""
    return _ops.dq_to_rt(_be(), dq)


def normalize(dq: torch.Tensor) -> torch.Tensor:
    """Unit dual quaternion: divide by ``|qr|``; if the batch as a whole is then not unit
    (``is_unit``), also remove the component of ``qd`` along ``qr`` -- the reference decides this
    ONCE for the whole batch.  Reference: dual_quat_torch.py:88-117."""
    return _ops.dq_normalize(_be(), dq)


def is_unit(dq: torch.Tensor, atol: float = 1e-03) -> bool:
    """``|qr|^2 ~ 1`` and ``qr . qd ~ 0`` for every element (or ``|qr|^2 ~ 0`` for every element).
    Reference: dual_quat_torch.py:120-143."""
    return _ops.dq_is_unit(_be(), dq, atol)


def unroll(dq: torch.Tensor, dim: int) -> torch.Tensor:
    """Dual-quaternion continuity along ``dim``: the sign is decided by the real part and applied to
    all eight components.  Reference: dual_quat_torch.py:146-174."""
    return _ops.dq_unroll(_be(), dq, dim)
