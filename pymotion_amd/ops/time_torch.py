"""Operations along the time axis -- drop-in for ``pymotion.ops.time_torch``."""
import torch

from .. import _backend, _ops


def _be():
    return _backend.torch_backend()


def interpolate_positions(sample_times: torch.Tensor, original_times: torch.Tensor, positions: torch.Tensor, dim: int,
                          method: str = "linear") -> torch.Tensor:
    """Linear interpolation of ``positions`` (time along ``dim``) at ``sample_times``.
    Reference: ops/time_torch.py:4-66.  One streaming gather kernel (``csrc/interp.hip``); the searchsorted
    over the two 1-D time tensors runs with torch ops where they live (no host synchronisation)."""
    return _ops.interpolate_positions(_be(), sample_times, original_times, positions, dim, method)
