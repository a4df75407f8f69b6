"""Operations along the time axis -- drop-in for ``pymotion.ops.time``."""
import numpy as np

from .. import _backend, _ops


def _be():
    return _backend.numpy_backend()


def interpolate_positions(sample_times: np.array, original_times: np.array, positions: np.array, axis: int,
                          method: str = "linear") -> np.array:
    """Linear interpolation of ``positions`` (time along ``axis``) at ``sample_times``; times outside
    ``original_times`` extrapolate from the first / last interval.  Reference: ops/time.py:4-66.

    One streaming gather kernel on the GPU (``csrc/interp.hip``); the searchsorted over the two 1-D time
    arrays (time.py:49-54) stays on the host.  Result dtype = what ``(1 - weights) * positions`` promotes to."""
    return _ops.interpolate_positions(_be(), sample_times, original_times, positions, axis, method)
