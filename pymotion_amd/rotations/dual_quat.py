"""Dual quaternions ``[..., 8] = [qr(4), qd(4)]`` -- drop-in for ``pymotion.rotations.dual_quat``.

Reference: ``pymotion/rotations/dual_quat.py``.  One gfx950 kernel per call, fp32 on the GPU.
Not covered here: ``unroll`` (sequential in time).
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
