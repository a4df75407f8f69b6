"""Zhou et al. 6D rotations ``[..., 3, 2]`` (first two COLUMNS of R) -- drop-in for
``pymotion.rotations.ortho6d``.

Reference: ``pymotion/rotations/ortho6d.py``.  One gfx950 kernel per call, fp32 on the GPU.
Degenerate input (a zero column): NaN, like the NumPy reference which divides by the raw norm (:83-85).
"""
import numpy as np

from .. import _backend, _ops


def _be():
    return _backend.numpy_backend()


def from_quat(quaternions: np.array) -> np.array:
    """Reference: ortho6d.py:14-28 (to_matrix then the first two columns)."""
    return _ops.o6d_from_quat(_be(), quaternions)


def from_matrix(rotmats: np.array) -> np.array:
    """``rotmats[..., :2]`` as a CONTIGUOUS array (the reference returns a view).
    Reference: ortho6d.py:31-47."""
    return _ops.o6d_from_matrix(_be(), rotmats)


def to_quat(ortho6D: np.array) -> np.array:
    """Gram-Schmidt then the reference's matrix -> quaternion.  Reference: ortho6d.py:50-64."""
    return _ops.o6d_to_quat(_be(), ortho6D)


def to_matrix(ortho6D: np.array) -> np.array:
    """Gram-Schmidt on the two columns, third column = cross product.
    Reference: ortho6d.py:67-90."""
    return _ops.o6d_to_matrix(_be(), ortho6D)
