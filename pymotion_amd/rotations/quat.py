"""Quaternion algebra and conversions -- drop-in for ``pymotion.rotations.quat``.

Same function names, positional arguments and conventions as the reference
(``pymotion/rotations/quat.py``): quaternions are ``[..., [w,x,y,z]]``, matrices
``[..., 3, 3]`` row-major.  Every function launches one hand-written gfx950 kernel from
``libpmhip.so`` (fp32 on the GPU); there is no CPU fallback.
NumPy arrays are copied to the GPU and back; outputs use the dtype the reference would return.
"""
import numpy as np

from .. import _backend, _ops


def _be():
    return _backend.numpy_backend()


def from_scaled_angle_axis(scaledaxis: np.array) -> np.array:
    """``[..., 3]`` scaled axis (|v| = angle) -> quaternion.  Reference: quat.py:6-21
    (a zero vector gives NaN there and here)."""
    return _ops.quat_from_scaled_angle_axis(_be(), scaledaxis)


def from_angle_axis(angle: np.array, axis: np.array) -> np.array:
    """``angle [..., 1]`` (radians), unit ``axis [..., 3]`` -> ``[cos(a/2), sin(a/2) axis]``.
    Reference: quat.py:24-40."""
    return _ops.quat_from_angle_axis(_be(), angle, axis)


def from_euler(euler: np.array, order) -> np.array:
    """Euler angles ``[..., 3]`` (radians) with per-element axis order (NumPy array of
    'x'|'y'|'z', same leading shape) -> quaternion ``q0 (x) (q1 (x) q2)``.
    Reference: quat.py:43-82."""
    return _ops.quat_from_euler(_be(), euler, order)


def from_matrix(rotmats: np.array) -> np.array:
    """Rotation matrices ``[..., 3, 3]`` -> quaternions; the reference's 4-branch selection and
    final normalise, sign NOT canonicalised.  Reference: quat.py:85-156."""
    return _ops.quat_from_matrix(_be(), rotmats)


def to_euler(quaternions: np.array, order) -> np.array:
    """Quaternion -> intrinsic Euler angles in ``[0, 2pi)`` for the given per-element order.
    Reference: quat.py:159-227."""
    return _ops.quat_to_euler(_be(), quaternions, order)


def to_scaled_angle_axis(quaternions: np.array) -> np.array:
    """Reference: quat.py:230-244."""
    return _ops.quat_to_scaled_angle_axis(_be(), quaternions)


def to_angle_axis(quaternions: np.array):
    """-> ``(angle [..., 1], axis [..., 3])``; axis is 0 where sin(a/2) <= 1e-8.
    Reference: quat.py:247-273."""
    return _ops.quat_to_angle_axis(_be(), quaternions)


def to_matrix(quaternions: np.array) -> np.array:
    """Quaternions -> ``[..., 3, 3]`` (no normalisation).  Reference: quat.py:276-317
    (float64 output like the reference, :306)."""
    return _ops.quat_to_matrix(_be(), quaternions)


def mul_vec(q: np.array, v: np.array) -> np.array:
    """Rotate vectors ``[..., 3]`` by unit quaternions.  Reference: quat.py:320-334."""
    return _ops.quat_mul_vec(_be(), q, v)


def mul(q0: np.array, q1: np.array) -> np.array:
    """Hamilton product (broadcasts leading dims).  Reference: quat.py:337-361."""
    return _ops.quat_mul(_be(), q0, q1)


def length(quaternions: np.array) -> np.array:
    """Reference: quat.py:364-376."""
    return _ops.quat_length(_be(), quaternions)


def inverse(quaternions: np.array) -> np.array:
    """Inverse of a UNIT quaternion = conjugate.  Reference: quat.py:379-393."""
    return _ops.quat_conjugate(_be(), quaternions)


def conjugate(quaternions: np.array) -> np.array:
    """Reference: quat.py:396-408."""
    return _ops.quat_conjugate(_be(), quaternions)


def normalize(quaternions: np.array, eps: float = 1e-8) -> np.array:
    """``q / (|q| + eps)`` -- eps is added to the norm.  Reference: quat.py:411-423."""
    return _ops.quat_normalize(_be(), quaternions, eps)


def slerp(q0: np.array, q1: np.array, t, shortest: bool = True) -> np.array:
    """Spherical interpolation, ``t`` a float or ``[..., 1]``.  Reference: quat.py:465-501."""
    return _ops.quat_slerp(_be(), q0, q1, t, shortest)


def unroll(quaternions: np.array, axis: int) -> np.array:
    """Remove double-cover sign flips along ``axis``: each quaternion takes the sign closest to its
    (already corrected) predecessor, the first one is kept.  A prefix-XOR scan on the GPU instead of the
    reference's Python loop over frames; returns a new array (the reference flips its argument in
    place through a view).  Reference: quat.py:426-462."""
    return _ops.quat_unroll(_be(), quaternions, axis)


def from_to(v1: np.array, v2: np.array, normalize_input: bool = True) -> np.array:
    """Quaternion rotating direction ``v1`` onto ``v2``; parallel -> identity, anti-parallel -> a half turn
    about an axis orthogonal to ``v1`` (``isclose`` thresholds of the reference, evaluated per element in
    the kernel instead of masked scatters).  Reference: quat.py:504-576."""
    return _ops.quat_from_to(_be(), v1, v2, normalize_input)


def from_to_axis(v1: np.array, v2: np.array, rot_axis: np.array, normalize_input: bool = True) -> np.array:
    """Same angle as ``from_to`` but about the given axis (sign from ``(v1 x v2) . rot_axis``).
    Reference: quat.py:579-650."""
    return _ops.quat_from_to_axis(_be(), v1, v2, rot_axis, normalize_input)
