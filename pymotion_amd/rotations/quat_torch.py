"""Quaternion algebra and conversions -- drop-in for ``pymotion.rotations.quat_torch``.

Same function names, positional arguments and conventions as the reference
(``pymotion/rotations/quat_torch.py``): quaternions are ``[..., [w,x,y,z]]``, matrices
``[..., 3, 3]`` row-major.  Every function launches one hand-written gfx950 kernel from
``libpmhip.so`` (fp32 on the GPU); there is no CPU fallback.
Tensors may live on a HIP device (zero-copy, torch's current stream) or on the CPU (copied over and back).
"""
import torch

from .. import _backend, _ops


def _be():
    return _backend.torch_backend()


def from_scaled_angle_axis(scaledaxis: torch.Tensor) -> torch.Tensor:
    """``[..., 3]`` scaled axis (|v| = angle) -> quaternion.  Reference: quat_torch.py:6-21
    (a zero vector gives NaN there and here)."""
    return _ops.quat_from_scaled_angle_axis(_be(), scaledaxis)


def from_angle_axis(angle: torch.Tensor, axis: torch.Tensor) -> torch.Tensor:
    """``angle [..., 1]`` (radians), unit ``axis [..., 3]`` -> ``[cos(a/2), sin(a/2) axis]``.
    Reference: quat_torch.py:24-40."""
    return _ops.quat_from_angle_axis(_be(), angle, axis)


def from_euler(euler: torch.Tensor, order) -> torch.Tensor:
    """Euler angles ``[..., 3]`` (radians) with per-element axis order (NumPy array of
    'x'|'y'|'z', same leading shape) -> quaternion ``q0 (x) (q1 (x) q2)``.
    Reference: quat_torch.py:43-82."""
    return _ops.quat_from_euler(_be(), euler, order)


def from_matrix(rotmats: torch.Tensor) -> torch.Tensor:
    """Rotation matrices ``[..., 3, 3]`` -> quaternions; the reference's 4-branch selection and
    final normalise, sign NOT canonicalised.  Reference: quat_torch.py:85-156."""
    return _ops.quat_from_matrix(_be(), rotmats)


def to_euler(quaternions: torch.Tensor, order) -> torch.Tensor:
    """Quaternion -> intrinsic Euler angles in ``[0, 2pi)`` for the given per-element order.
    Reference: quat_torch.py:159-227."""
    return _ops.quat_to_euler(_be(), quaternions, order)


def to_scaled_angle_axis(quaternions: torch.Tensor) -> torch.Tensor:
    """Reference: quat_torch.py:230-244."""
    return _ops.quat_to_scaled_angle_axis(_be(), quaternions)


def to_angle_axis(quaternions: torch.Tensor):
    """-> ``(angle [..., 1], axis [..., 3])``; axis is 0 where sin(a/2) <= 1e-8.
    Reference: quat_torch.py:247-273."""
    return _ops.quat_to_angle_axis(_be(), quaternions)


def to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """Quaternions -> ``[..., 3, 3]`` (no normalisation).  Reference: quat_torch.py:276-317
    (default-dtype output like the torch twin, :318)."""
    return _ops.quat_to_matrix(_be(), quaternions)


def mul_vec(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Rotate vectors ``[..., 3]`` by unit quaternions.  Reference: quat_torch.py:320-334."""
    return _ops.quat_mul_vec(_be(), q, v)


def mul(q0: torch.Tensor, q1: torch.Tensor) -> torch.Tensor:
    """Hamilton product (broadcasts leading dims).  Reference: quat_torch.py:337-361."""
    return _ops.quat_mul(_be(), q0, q1)


def length(quaternions: torch.Tensor) -> torch.Tensor:
    """Reference: quat_torch.py:364-376."""
    return _ops.quat_length(_be(), quaternions)


def inverse(quaternions: torch.Tensor) -> torch.Tensor:
    """Inverse of a UNIT quaternion = conjugate.  Reference: quat_torch.py:379-393."""
    return _ops.quat_conjugate(_be(), quaternions)


def conjugate(quaternions: torch.Tensor) -> torch.Tensor:
    """Reference: quat_torch.py:396-408."""
    return _ops.quat_conjugate(_be(), quaternions)


def normalize(quaternions: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """``q / (|q| + eps)`` -- eps is added to the norm.  Reference: quat_torch.py:411-423."""
    return _ops.quat_normalize(_be(), quaternions, eps)


def slerp(q0: torch.Tensor, q1: torch.Tensor, t, shortest: bool = True) -> torch.Tensor:
    """Spherical interpolation, ``t`` a float or ``[..., 1]``.  Reference: quat_torch.py:465-501."""
    return _ops.quat_slerp(_be(), q0, q1, t, shortest)


def unroll(quaternions: torch.Tensor, dim: int) -> torch.Tensor:
    """Remove double-cover sign flips along ``dim``: each quaternion takes the sign closest to its
    (already corrected) predecessor, the first one is kept.  A prefix-XOR scan on the GPU instead of the
    reference's Python loop over frames; returns a new array (the reference flips its argument in
    place through a view).  Reference: quat_torch.py:441-477."""
    return _ops.quat_unroll(_be(), quaternions, dim)


def from_to(v1: torch.Tensor, v2: torch.Tensor, normalize_input: bool = True) -> torch.Tensor:
    """Quaternion rotating direction ``v1`` onto ``v2``; parallel -> identity, anti-parallel -> a half turn
    about an axis orthogonal to ``v1`` (``isclose`` thresholds of the reference, evaluated per element in
    the kernel instead of masked scatters).  Reference: quat_torch.py:521-601."""
    return _ops.quat_from_to(_be(), v1, v2, normalize_input)


def from_to_axis(v1: torch.Tensor, v2: torch.Tensor, rot_axis: torch.Tensor, normalize_input: bool = True) -> torch.Tensor:
    """Same angle as ``from_to`` but about the given axis (sign from ``(v1 x v2) . rot_axis``).
    Reference: quat_torch.py:603-676."""
    return _ops.quat_from_to_axis(_be(), v1, v2, rot_axis, normalize_input)
