"""Skeleton operations -- drop-in for ``pymotion.ops.skeleton`` (hot-path subset).

A skeleton is ``(parents [J] int, offsets [J, 3] float)`` plus per-frame local rotations and
the root's global position (reference: ``pymotion/ops/skeleton.py:6-13``).  ``parents``
must list every joint after its parent (``parents[i] < i``), the order the reference's
in-place loops rely on; anything else raises ``ValueError``.

Each function is ONE launch of a hand-written gfx950 kernel (``pymotion_amd/csrc``): a
64-lane wave walks the parent chains of a tile of frames with the skeleton constants in
scalar registers, rotation inputs and transform outputs staged through LDS so that all HBM
traffic is contiguous 16-byte-per-lane loads / stores.
"""
import numpy as np

from .. import _backend, _ops


def _be():
    return _backend.numpy_backend()


def fk(rot: np.array, global_pos: np.array, offsets: np.array, parents) -> np.array:
    """Forward kinematics.  Reference: ops/skeleton.py:16-61.

    rot ``[..., J, 4]`` (normalised internally), global_pos ``[..., 3]``, offsets ``[J, 3]`` or
    ``[..., J, 3]``, parents ``[J]`` -> ``(positions [..., J, 3], rotmats [..., J, 3, 3])``.
    Output is float64 like the reference (:44), as contiguous arrays (the reference returns views).
    """
    return _ops.fk(_be(), rot, global_pos, offsets, parents)


def fk_from_ortho6d(ortho6D: np.array, global_pos: np.array, offsets: np.array, parents, return_quat: bool = False):
    """``fk(ortho6d.to_quat(x), ...)`` fused into one kernel (no quaternion round trip through HBM).
    Reference chain: rotations/ortho6d.py to_quat -> ops/skeleton.py fk.
    Returns ``(positions, rotmats[, quats])``."""
    return _ops.fk_from_ortho6d(_be(), ortho6D, global_pos, offsets, parents, return_quat)


def from_global_rotations(global_quats: np.array, parents) -> np.array:
    """World-space quaternions -> local: ``conj(q_parent) (x) q_child``.
    Reference: ops/skeleton.py:64-93."""
    return _ops.from_global_rotations(_be(), global_quats, parents)


def from_root_dual_quat(dq: np.array, parents):
    """Root-centred dual quaternions -> ``(translations [..., J, 3], rotations [..., J, 4])`` -- in
    that order, like the reference's return statement (ops/skeleton.py:204; its docstring says
    the opposite).  ``translations[..., 0, :]`` is the root's global position.
    Reference: ops/skeleton.py:173-204."""
    return _ops.from_root_dual_quat(_be(), dq, parents)


def to_root_dual_quat(rotations: np.array, global_pos: np.array, parents, offsets: np.array):
    """Skeleton pose -> root-centred dual quaternions ``[..., J, 8]``.  NOTE the argument order
    (parents before offsets) differs from ``fk``, as in the reference.  Inputs are not
    normalised; ``offsets[0]`` must be 0.  The joint axis is -2 for any number of leading dims
    (the reference reads ``shape[1]``, which is only right for ``[F, J, 4]``).
    Reference: ops/skeleton.py:207-244."""
    return _ops.to_root_dual_quat(_be(), rotations, global_pos, parents, offsets)


def mirror(
    local_rotations: np.array,
    global_translation: np.array,
    parents,
    offsets: np.array,
    end_sites: np.array = None,
    joints_mapping=None,
    mode: str = "all",
    axis: str = "X",
):
    """Mirror a skeleton pose along ``axis``.  ``mode='all'``: perfect mirror, topology mirrored too
    (offsets / end sites change sign); ``mode='symmetry'``: joints swapped through ``joints_mapping``,
    skeleton unchanged.  One fused kernel (a quaternion tree walk with the sign convention of
    fk -> from_matrix, then permute / negate / from_global_rotations; csrc/mirror.hip).
    ``mode='positions'``: positions are mirrored and the rotations recovered by ``from_root_positions``
    (twist is not preserved).  Unlike the reference's
    'symmetry' mode the caller's ``global_translation`` is not modified in place.
    Returns ``(local_rotations, global_translation, offsets, end_sites)``.
    Reference: ops/skeleton.py:247-344 (and _true_mirror below it)."""
    return _ops.mirror(_be(), local_rotations, global_translation, parents, offsets, end_sites, joints_mapping, mode, axis)


def from_root_positions(positions: np.array, parents, offsets: np.array) -> np.array:
    """Root-centred joint positions ``[..., J, 3]`` -> local rotations ``[..., J, 4]``: every joint with
    children is turned so that its first child points where the positions say (``quat.from_to``), further
    children fix the roll (``quat.from_to_axis``); joints without children keep the identity.  One O(J)
    walk per frame on the GPU instead of the reference's fk-per-joint loop.
    Reference: ops/skeleton.py:96-170."""
    return _ops.from_root_positions(_be(), positions, parents, offsets)
