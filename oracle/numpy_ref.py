"""NumPy restatement of the reference's skeleton hot path -- TEST INFRASTRUCTURE ONLY.

Purpose: (1) the *cost-equivalent* CPU baseline that ``bench.py`` times on the GPU box
(``cpu_baseline.kind == "port"``): same algorithm, same dtype promotions, same number and
shape of NumPy kernels as ``pymotion/ops/skeleton.py`` -- float64 ``[F,J,4,4]`` scratch,
normalise + quaternion->matrix, then one batched ``np.matmul`` per joint in parent order;
(2) a second, independent oracle next to the C one (``pm_oracle.c``).

It is validated against the imported reference in this container by
``oracle/make_golden.py --check-numpy-ref`` (outputs <= 1e-12 apart, timing within the
band recorded in DESIGN.md) and against the committed goldens by ``tests/test_oracle.py``.
Nothing under ``pymotion_amd`` imports this module.
"""
import numpy as np

EPS = 1e-8


# ---- quaternion helpers (rotations/quat.py) --------------------------------------------------

def q_normalize(q, eps=EPS):
    # quat.py:411-423 -- eps is added to the norm, dtype of q is kept
    return q / (np.linalg.norm(q, axis=-1)[..., None] + eps)


def q_to_matrix(q):
    # quat.py:276-317 -- products in q's dtype, result array is always float64
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2, y2, z2 = x + x, y + y, z + z
    xx, yy, wx = x * x2, y * y2, w * x2
    xy, yz, wy = x * y2, y * z2, w * y2
    xz, zz, wz = x * z2, z * z2, w * z2
    out = np.empty(q.shape[:-1] + (3, 3))
    out[..., 0, 0] = 1.0 - (yy + zz)
    out[..., 0, 1] = xy - wz
    out[..., 0, 2] = xz + wy
    out[..., 1, 0] = xy + wz
    out[..., 1, 1] = 1.0 - (xx + zz)
    out[..., 1, 2] = yz - wx
    out[..., 2, 0] = xz - wy
    out[..., 2, 1] = yz + wx
    out[..., 2, 2] = 1.0 - (xx + yy)
    return out


def _cross(a, b):
    # quat.py:653-674 (three-way concatenate, kept for cost equivalence)
    return np.concatenate(
        [
            a[..., 1:2] * b[..., 2:3] - a[..., 2:3] * b[..., 1:2],
            a[..., 2:3] * b[..., 0:1] - a[..., 0:1] * b[..., 2:3],
            a[..., 0:1] * b[..., 1:2] - a[..., 1:2] * b[..., 0:1],
        ],
        axis=-1,
    )


def q_mul(a, b):
    # quat.py:337-361
    aw, ax, ay, az = (a[..., i:i + 1] for i in range(4))
    bw, bx, by, bz = (b[..., i:i + 1] for i in range(4))
    return np.concatenate(
        (
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + bw * ax + ay * bz - az * by,
            aw * by + bw * ay + az * bx - ax * bz,
            aw * bz + bw * az + ax * by - ay * bx,
        ),
        axis=-1,
    )


def q_mul_vec(q, v):
    # quat.py:320-334
    t = 2.0 * _cross(q[..., 1:], v)
    return v + q[..., 0][..., None] * t + _cross(q[..., 1:], t)


def q_conj(q):
    # quat.py:396-408
    return np.concatenate((q[..., 0:1], -q[..., 1:]), axis=-1)


# ---- dual quaternions (rotations/dual_quat.py) ---------------------------------------------------

def dq_from_rt(q, t):
    # dual_quat.py:12-36 -- the (0,t) quaternion is float64 zeros, so the result is float64
    tq = np.zeros(t.shape[:-1] + (4,))
    tq[..., 1:] = t
    return np.concatenate((q, 0.5 * q_mul(tq, q)), axis=-1)


def dq_to_rt(dq):
    # dual_quat.py:62-83
    dq = dq.copy()
    qr, qd = dq[..., :4], dq[..., 4:]
    return qr, (2 * q_mul(qd, q_conj(qr)))[..., 1:]


# ---- skeleton ops (ops/skeleton.py) -----------------------------------------------------------------

def fk(rot, global_pos, offsets, parents):
    """ops/skeleton.py:16-61, cost-equivalent: f64 [...,J,4,4] scratch, J-1 batched 4x4 matmuls."""
    T = np.zeros(rot.shape[:-1] + (4, 4))
    T[..., :3, :3] = q_to_matrix(q_normalize(rot))
    T[..., :3, 3] = offsets
    T[..., 3, 3] = 1
    T[..., 0, :3, 3] = global_pos
    for j in range(1, len(parents)):
        T[..., j, :, :] = np.matmul(T[..., parents[j], :, :], T[..., j, :, :])
    return T[..., :3, 3], T[..., :3, :3]


def fk_chunked(rot, global_pos, offsets, parents, chunk=1 << 17):
    """fk over the leading (frame) axis in chunks so the f64 scratch (128 B/joint) stays bounded."""
    F = rot.shape[0]
    pos = np.empty(rot.shape[:-1] + (3,))
    rm = np.empty(rot.shape[:-1] + (3, 3))
    for s in range(0, F, chunk):
        e = min(F, s + chunk)
        off = offsets[s:e] if offsets.ndim == 3 else offsets
        p, r = fk(rot[s:e], global_pos[s:e], off, parents)
        pos[s:e], rm[s:e] = p, r
    return pos, rm


def to_root_dual_quat(rotations, global_pos, parents, offsets):
    """ops/skeleton.py:207-244 (joint axis -2; depth-1 joints stay local)."""
    assert (offsets[0] == 0).all()
    J = rotations.shape[-2]
    q = rotations.copy()
    t = np.tile(offsets, rotations.shape[:-2] + (1, 1))
    t[..., 0, :] = global_pos
    for j in range(1, J):
        p = parents[j]
        if p == 0:
            continue
        t[..., j, :] = q_mul_vec(q[..., p, :], t[..., j, :]) + t[..., p, :]
        q[..., j, :] = q_mul(q[..., p, :], q[..., j, :])
    return dq_from_rt(q, t)


def from_root_dual_quat(dq, parents):
    """ops/skeleton.py:173-204 -> (translations, rotations); reverse joint order."""
    J = dq.shape[-2]
    q, t = dq_to_rt(dq.copy())
    for j in reversed(range(1, J)):
        p = parents[j]
        if p == 0:
            continue
        inv = q_conj(q[..., p, :])
        t[..., j, :] = q_mul_vec(inv, t[..., j, :] - t[..., p, :])
        q[..., j, :] = q_mul(inv, q[..., j, :])
    return t, q
