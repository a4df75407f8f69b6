"""Deterministic synthetic skeletons and workloads.

No BVH file ships with the reference (``test.bvh`` is git-ignored there), so the
22-joint topology is the one implied by the joint names of the reference's
README.md:49; the 52-joint tree is an SMPL-H-like body (22) + 2 x 15 hand joints.
Both satisfy ``parents[i] < i`` (the order ``ops/skeleton.py:51-58`` relies on).
Host-side NumPy only: shared by bench.py, the tests and ``__graft_entry__.smoke``.
"""
import numpy as np

# Hips -> {LeftHip chain, RightHip chain, Chest -> Chest3 -> Chest4 -> {Neck->Head, L collar chain, R collar chain}}
PARENTS_22 = np.array(
    [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 11, 18, 19, 20], dtype=np.int32
)
JOINT_NAMES_22 = [
    "Hips", "LeftHip", "LeftKnee", "LeftAnkle", "LeftToe", "RightHip", "RightKnee", "RightAnkle",
    "RightToe", "Chest", "Chest3", "Chest4", "Neck", "Head", "LeftCollar", "LeftShoulder", "LeftElbow",
    "LeftWrist", "RightCollar", "RightShoulder", "RightElbow", "RightWrist",
]

_SMPL_BODY = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19]


def _smplh_parents():
    p = list(_SMPL_BODY)
    for wrist in (20, 21):
        base = len(p)
        for finger in range(5):
            p += [wrist, base + 3 * finger, base + 3 * finger + 1]
    return np.array(p, dtype=np.int32)


PARENTS_52 = _smplh_parents()


def random_parents(J, rng):
    """A random valid topology: parents[0] = 0, parents[i] uniform in [0, i)."""
    p = np.zeros(J, dtype=np.int32)
    for i in range(1, J):
        p[i] = rng.integers(0, i)
    return p


def depth_of(parents):
    d = np.zeros(len(parents), dtype=np.int32)
    for i in range(1, len(parents)):
        d[i] = d[parents[i]] + 1
    return d


def make_offsets(J, rng, scale=0.3):
    """Metre-scale bone offsets, row 0 = 0 (``to_root_dual_quat`` asserts it, ops/skeleton.py:227)."""
    off = rng.uniform(-scale, scale, (J, 3)).astype(np.float32)
    off[0] = 0
    return off


def fk_workload(F, parents=PARENTS_22, seed=0, normalized=False, offset_scale=0.3):
    """rot ~ N(0,1) fp32 (not pre-normalised unless asked), root U(-2,2) m."""
    rng = np.random.default_rng(seed)
    J = len(parents)
    rot = rng.standard_normal((F, J, 4), dtype=np.float32)
    if normalized:
        rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = make_offsets(J, rng, offset_scale)
    return rot, root, off, np.asarray(parents, dtype=np.int32)


def o6d_workload(F, parents=PARENTS_52, seed=0, offset_scale=0.15):
    """generic (non-orthonormal) 6D inputs, exercises Gram-Schmidt."""
    rng = np.random.default_rng(seed)
    J = len(parents)
    x = rng.standard_normal((F, J, 3, 2), dtype=np.float32)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = make_offsets(J, rng, offset_scale)
    return x, root, off, np.asarray(parents, dtype=np.int32)


def write_synthetic_bvh(path, n_frames=48, seed=7):
    """Deterministic 22-joint BVH (joint names of the reference's README.md:49, topology
    synthetic.PARENTS_22, metre-scale offsets, End Sites on the leaves).  Angles drift smoothly over
    several turns so that consecutive quaternions cross the double cover and `unroll` has work to do."""
    rng = np.random.default_rng(seed)
    names, parents = JOINT_NAMES_22, PARENTS_22
    J = len(names)
    off = make_offsets(J, rng, 0.3).astype(np.float64).round(6)
    kids = [[] for _ in range(J)]
    for j in range(1, J):
        kids[parents[j]].append(j)
    lines = ["HIERARCHY"]

    def emit(j, depth):
        tab = "\t" * depth
        lines.append(f"{tab}{'ROOT' if j == 0 else 'JOINT'} {names[j]}")
        lines.append(tab + "{")
        lines.append(f"{tab}\tOFFSET {off[j, 0]:.6f} {off[j, 1]:.6f} {off[j, 2]:.6f}")
        if j == 0:
            lines.append(f"{tab}\tCHANNELS 6 Xposition Yposition Zposition Zrotation Xrotation Yrotation")
        else:
            lines.append(f"{tab}\tCHANNELS 3 " + ("Zrotation Xrotation Yrotation" if j % 2 else "Yrotation Zrotation Xrotation"))
        for c in kids[j]:
            emit(c, depth + 1)
        if not kids[j]:
            lines.append(f"{tab}\tEnd Site")
            lines.append(tab + "\t{")
            lines.append(f"{tab}\t\tOFFSET 0.000000 0.100000 0.000000")
            lines.append(tab + "\t}")
        lines.append(tab + "}")

    emit(0, 0)
    lines += ["MOTION", f"Frames: {n_frames}", "Frame Time: 0.016667"]
    t = np.arange(n_frames)[:, None, None]
    rate = rng.uniform(-25, 25, (1, J, 3))
    ang = rng.uniform(-180, 180, (1, J, 3)) + rate * t + rng.normal(0, 2, (n_frames, J, 3))
    root = np.cumsum(rng.normal(0, 0.01, (n_frames, 3)), axis=0) + [0.0, 0.9, 0.0]
    for f in range(n_frames):
        vals = list(root[f]) + list(ang[f, 0])
        for j in range(1, J):
            vals += list(ang[f, j])
        lines.append(" ".join(f"{v:.6f}" for v in vals))
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
