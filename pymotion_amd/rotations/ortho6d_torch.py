"""Zhou et al. 6D rotations ``[..., 3, 2]`` (first two COLUMNS of R) -- drop-in for
``pymotion.rotations.ortho6d_torch``.

Reference: ``pymotion/rotations/ortho6d_torch.py``.  One gfx950 kernel per call, fp32 on the GPU.
Degenerate input (a zero column): finite zeros, like ``F.normalize(eps=1e-12)`` in the torch twin (:84-89).
"""
import torch

from .. import _backend, _ops


def _be():
    return _backend.torch_backend()


def from_quat(quaternions: torch.Tensor) -> torch.Tensor:
    """Reference: ortho6d_torch.py:15-29 (to_matrix then the first two columns)."""
    return _ops.o6d_from_quat(_be(), quaternions)


def from_matrix(rotmats: torch.Tensor) -> torch.Tensor:
    """``rotmats[..., :2]`` as a CONTIGUOUS array (the reference returns a view).
    Reference: ortho6d_torch.py:32-48."""
    return _ops.o6d_from_matrix(_be(), rotmats)


def to_quat(ortho6D: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt then the reference's matrix -> quaternion.  Reference: ortho6d_torch.py:51-65."""
    return _ops.o6d_to_quat(_be(), ortho6D)


def to_matrix(ortho6D: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt on the two columns, third column = cross product.
    Reference: ortho6d_torch.py:68-96."""
    return _ops.o6d_to_matrix(_be(), ortho6D)
