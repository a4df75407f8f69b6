"""Dual quaternions ``[..., 8] = [qr(4), qd(4)]`` -- drop-in for ``pymotion.rotations.dual_quat``.

Reference: ``pymotion/rotations/dual_quat.py``.  One gfx950 kernel per call, fp32 on the GPU.
``normalize`` / ``is_unit`` keep the reference's whole-batch branch (one host read of three device
counters where the reference's Python ``if`` synchronises).
"""
import numpy as np

from .. import _backend, _ops


def _be():
    return _backend.numpy_backend()


def from_rotation_translation(rotations: np.array, translations: np.array) -> np.array:
    """``dq = [q, 0.5 (0,t) (x) q]``.  Reference: dual_quat.py:12-36."""
    return _ops.dq_from_rt(_be(), rotations, translations)


def from_translation(translations: np.array) -> np.array:
    """``[1,0,0,0, 0, t/2]``.  Reference: dual_quat.py:39-59."""
    return _ops.dq_from_t(_be(), translations)


def to_rotation_translation(dq: np.array):
    """-> ``(rotations [..., 4], translations [..., 3])``, ``t = (2 qd (x) conj(qr))[1:]``.
    Reference: dual_quat.py:62-83."""
    return _ops.dq_to_rt(_be(), dq)


def normalize(dq: np.array) -> np.array:
    """Unit dual quaternion: divide by ``|qr|``; if the batch as a whole is then not unit
    (``is_unit``), also remove the component of ``qd`` along ``qr`` -- the reference decides this
    ONCE for the whole batch.  Reference: dual_quat.py:86-115."""
    return _ops.dq_normalize(_be(), dq)


def is_unit(dq: np.array, atol: float = 1e-03) -> bool:
    """``|qr|^2 ~ 1`` and ``qr . qd ~ 0`` for every element (or ``|qr|^2 ~ 0`` for every element).
    Reference: dual_quat.py:118-136."""
    return _ops.dq_is_unit(_be(), dq, atol)


def unroll(dq: np.array, axis: int) -> np.array:
    """Dual-quaternion continuity along ``axis``: the sign is decided by the real part and applied to
    all eight components.  Reference: dual_quat.py:139-167."""
    return _ops.dq_unroll(_be(), dq, axis)
