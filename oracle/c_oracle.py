"""ctypes front-end of ``oracle/_build/liboracle.so`` (see ``pm_oracle.c``) -- TEST INFRASTRUCTURE ONLY.

Array-in / array-out with the reference's shapes (``[..., C]`` element-wise,
``[..., J, C]`` skeleton ops); dtype float32 -> ``*_f32`` entry points, anything
else -> ``*_f64``.  Each wrapper names the reference function it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_ORACLE_ASAN=1: the AddressSanitizer / UBSan build (`make -C oracle asan`; the process must have libasan preloaded)
_ASAN = os.environ.get("PM_ORACLE_ASAN") == "1"
_SO = os.path.join(_HERE, "_build", "liboracle_asan.so" if _ASAN else "liboracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("pm_oracle.c", "pm_oracle_impl.h")
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "_build/" + os.path.basename(_SO)], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _prep(a, dt):
    return np.ascontiguousarray(np.asarray(a), dtype=dt)


def _dt(*arrs):
    return np.float32 if all(np.asarray(a).dtype == np.float32 for a in arrs) else np.float64


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _call(name, dt, *args):
    fn = getattr(lib(), f"{name}_{'f32' if dt == np.float32 else 'f64'}")
    fn.restype = None
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(_p(a))
        elif isinstance(a, float):
            conv.append(C.c_float(a) if dt == np.float32 else C.c_double(a))
        elif isinstance(a, tuple):  # (ctype, value)
            conv.append(a[0](a[1]))
        elif a is None:
            conv.append(C.c_void_p(0))
        else:
            raise TypeError(type(a))
    fn(*conv)


def _i64(v):
    return (C.c_int64, int(v))


def _i32(v):
    return (C.c_int32, int(v))


def _ew(name, ins, in_w, out_w, extra=()):
    """Generic element-wise dispatch: ins[k] has trailing width in_w[k]."""
    dt = _dt(*ins)
    ins = [_prep(a, dt) for a in ins]
    lead = np.broadcast_shapes(*[a.shape[:a.ndim - len(w)] for a, w in zip(ins, in_w)])
    ins = [np.ascontiguousarray(np.broadcast_to(a, lead + tuple(w))) for a, w in zip(ins, in_w)]
    n = int(np.prod(lead, dtype=np.int64))
    outs = [np.empty(lead + tuple(w), dtype=dt) for w in out_w]
    return dt, ins, n, outs


# ---- element-wise -------------------------------------------------------------------------

def quat_normalize(q, eps=1e-8):
    """rotations/quat.py:411-423"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(4,)])
    _call("oracle_quat_normalize", dt, q, _i64(n), float(eps), o)
    return o


def quat_length(q):
    """rotations/quat.py:364-376"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [()])
    _call("oracle_quat_length", dt, q, _i64(n), o)
    return o


def quat_to_matrix(q):
    """rotations/quat.py:276-317"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(3, 3)])
    _call("oracle_quat_to_matrix", dt, q, _i64(n), o)
    return o


def quat_from_matrix(m):
    """rotations/quat.py:85-156"""
    dt, (m,), n, (o,) = _ew("", [m], [(3, 3)], [(4,)])
    _call("oracle_quat_from_matrix", dt, m, _i64(n), o)
    return o


def quat_mul(a, b):
    """rotations/quat.py:337-361"""
    dt, (a, b), n, (o,) = _ew("", [a, b], [(4,), (4,)], [(4,)])
    _call("oracle_quat_mul", dt, a, b, _i64(n), o)
    return o


def quat_mul_vec(q, v):
    """rotations/quat.py:320-334"""
    dt, (q, v), n, (o,) = _ew("", [q, v], [(4,), (3,)], [(3,)])
    _call("oracle_quat_mul_vec", dt, q, v, _i64(n), o)
    return o


def quat_conjugate(q):
    """rotations/quat.py:396-408"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(4,)])
    _call("oracle_quat_conjugate", dt, q, _i64(n), o)
    return o


def dq_from_rt(q, t):
    """rotations/dual_quat.py:12-36"""
    dt, (q, t), n, (o,) = _ew("", [q, t], [(4,), (3,)], [(8,)])
    _call("oracle_dq_from_rt", dt, q, t, _i64(n), o)
    return o


def dq_to_rt(dq):
    """rotations/dual_quat.py:62-83 -> (rotations, translations)"""
    dt, (dq,), n, (q, t) = _ew("", [dq], [(8,)], [(4,), (3,)])
    _call("oracle_dq_to_rt", dt, dq, _i64(n), q, t)
    return q, t


def dq_from_t(t):
    """rotations/dual_quat.py:39-59"""
    dt, (t,), n, (o,) = _ew("", [t], [(3,)], [(8,)])
    _call("oracle_dq_from_t", dt, t, _i64(n), o)
    return o


def o6d_to_matrix(x, eps=0.0):
    """rotations/ortho6d.py:67-90 (eps=0: NumPy behaviour; 1e-12: torch twin)"""
    dt, (x,), n, (o,) = _ew("", [x], [(3, 2)], [(3, 3)])
    _call("oracle_o6d_to_matrix", dt, x, _i64(n), float(eps), o)
    return o


def o6d_to_quat(x, eps=0.0):
    """rotations/ortho6d.py:50-64"""
    dt, (x,), n, (o,) = _ew("", [x], [(3, 2)], [(4,)])
    _call("oracle_o6d_to_quat", dt, x, _i64(n), float(eps), o)
    return o


def o6d_from_quat(q):
    """rotations/ortho6d.py:14-28"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(3, 2)])
    _call("oracle_o6d_from_quat", dt, q, _i64(n), o)
    return o


def o6d_from_matrix(m):
    """rotations/ortho6d.py:31-47"""
    dt, (m,), n, (o,) = _ew("", [m], [(3, 3)], [(3, 2)])
    _call("oracle_o6d_from_matrix", dt, m, _i64(n), o)
    return o


def quat_from_angle_axis(angle, axis):
    """rotations/quat.py:24-40 ; angle [...,1], axis [...,3]"""
    dt, (angle, axis), n, (o,) = _ew("", [angle, axis], [(1,), (3,)], [(4,)])
    _call("oracle_quat_from_angle_axis", dt, angle, axis, _i64(n), o)
    return o


def quat_from_scaled_angle_axis(v):
    """rotations/quat.py:6-21"""
    dt, (v,), n, (o,) = _ew("", [v], [(3,)], [(4,)])
    _call("oracle_quat_from_scaled_angle_axis", dt, v, _i64(n), o)
    return o


def quat_to_angle_axis(q):
    """rotations/quat.py:247-273 -> (angle[...,1], axis[...,3])"""
    dt, (q,), n, (a, ax) = _ew("", [q], [(4,)], [(1,), (3,)])
    _call("oracle_quat_to_angle_axis", dt, q, _i64(n), a, ax)
    return a, ax


def quat_to_scaled_angle_axis(q):
    """rotations/quat.py:230-244"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(3,)])
    _call("oracle_quat_to_scaled_angle_axis", dt, q, _i64(n), o)
    return o


def quat_from_to(v1, v2, normalize_input=True):
    """rotations/quat.py:504-576"""
    dt, (v1, v2), n, (o,) = _ew("", [v1, v2], [(3,), (3,)], [(4,)])
    _call("oracle_quat_from_to", dt, v1, v2, _i64(n), (C.c_int, int(bool(normalize_input))), o)
    return o


def quat_from_to_axis(v1, v2, axis, normalize_input=True):
    """rotations/quat.py:579-650"""
    dt, (v1, v2, axis), n, (o,) = _ew("", [v1, v2, axis], [(3,), (3,), (3,)], [(4,)])
    _call("oracle_quat_from_to_axis", dt, v1, v2, axis, _i64(n), (C.c_int, int(bool(normalize_input))), o)
    return o


def from_root_positions(positions, parents, offsets):
    """ops/skeleton.py:96-170, literal procedure (fk per aligned joint)"""
    dt = _dt(positions, offsets)
    P = _prep(positions, dt)
    lead, J = P.shape[:-2], P.shape[-2]
    F = int(np.prod(lead, dtype=np.int64))
    o = np.empty(lead + (J, 4), dtype=dt)
    _call("oracle_from_root_positions", dt, P, _parents(parents), _prep(offsets, dt), _i64(F), _i32(J), o)
    return o


def quat_unroll(q, axis):
    """rotations/quat.py:426-462"""
    dt = _dt(q)
    a = np.moveaxis(_prep(q, dt), axis, 0)
    shp = a.shape
    a = np.ascontiguousarray(a).reshape(shp[0], -1, 4)
    o = np.empty_like(a)
    _call("oracle_quat_unroll", dt, a, _i64(a.shape[0]), _i32(a.shape[1]), o)
    return np.moveaxis(o.reshape(shp), 0, axis)


def interpolate_positions(sample_times, original_times, positions, axis):
    """ops/time.py:4-66 (any time axis; the reference's own broadcast needs it second to last)"""
    pos = np.asarray(positions)
    dt = np.float32 if (pos.dtype == np.float32 and np.asarray(sample_times).dtype == np.float32) else np.float64
    pos = np.ascontiguousarray(pos, dtype=dt)
    st = np.ascontiguousarray(sample_times, dtype=dt)
    ot = np.ascontiguousarray(original_times, dtype=dt)
    ax = axis % pos.ndim
    A = int(np.prod(pos.shape[:ax], dtype=np.int64))
    T = pos.shape[ax]
    B = int(np.prod(pos.shape[ax + 1:], dtype=np.int64))
    o = np.empty(pos.shape[:ax] + (len(st),) + pos.shape[ax + 1:], dtype=dt)
    _call("oracle_interpolate_positions", dt, st, ot, pos, _i64(A), _i64(T), _i64(len(st)), _i64(B), o)
    return o


_AX = {"x": 0, "y": 1, "z": 2}


def encode_order(order, lead):
    """['x'|'y'|'z'] string array [...,3] -> contiguous uint8 codes broadcast to lead+(3,)."""
    order = np.asarray(order)
    if order.dtype.kind in "US":
        codes = np.vectorize(lambda s: _AX[str(s)], otypes=[np.uint8])(order)
    else:
        codes = order.astype(np.uint8)
    return np.ascontiguousarray(np.broadcast_to(codes, tuple(lead) + (3,)))


def quat_from_euler(euler, order):
    """rotations/quat.py:43-82"""
    dt, (e,), n, (o,) = _ew("", [euler], [(3,)], [(4,)])
    codes = encode_order(order, e.shape[:-1])
    _call("oracle_quat_from_euler", dt, e, codes, _i64(n), o)
    return o


def quat_to_euler(q, order):
    """rotations/quat.py:159-227"""
    dt, (q,), n, (o,) = _ew("", [q], [(4,)], [(3,)])
    codes = encode_order(order, q.shape[:-1])
    _call("oracle_quat_to_euler", dt, q, codes, _i64(n), o)
    return o


def quat_slerp(q0, q1, t, shortest=True):
    """rotations/quat.py:465-501 ; t scalar or [...,1]"""
    t = np.asarray(t, dtype=_dt(q0, q1))
    if t.ndim == 0:
        t = t.reshape(1)
    dt, (q0, q1, t), n, (o,) = _ew("", [q0, q1, t], [(4,), (4,), (1,)], [(4,)])
    _call("oracle_quat_slerp", dt, q0, q1, t, _i64(n), (C.c_int, int(bool(shortest))), o)
    return o


# ---- skeleton ops --------------------------------------------------------------------------

def _parents(parents):
    return np.ascontiguousarray(np.asarray(parents), dtype=np.int32)


def fk(rot, global_pos, offsets, parents):
    """ops/skeleton.py:16-61 -> (positions [...,J,3], rotmats [...,J,3,3])"""
    dt = _dt(rot, global_pos, offsets)
    rot = _prep(rot, dt)
    lead, J = rot.shape[:-2], rot.shape[-2]
    F = int(np.prod(lead, dtype=np.int64))
    gp = np.ascontiguousarray(np.broadcast_to(_prep(global_pos, dt), lead + (3,)))
    off = _prep(offsets, dt)
    per_frame = off.ndim > 2
    if per_frame:
        off = np.ascontiguousarray(np.broadcast_to(off, lead + (J, 3)))
    pos = np.empty(lead + (J, 3), dtype=dt)
    rm = np.empty(lead + (J, 3, 3), dtype=dt)
    _call("oracle_fk", dt, rot, gp, off, (C.c_int, int(per_frame)), _parents(parents), _i64(F), _i32(J), pos, rm)
    return pos, rm


def to_root_dual_quat(rotations, global_pos, parents, offsets):
    """ops/skeleton.py:207-244 (joint axis = -2)"""
    dt = _dt(rotations, global_pos, offsets)
    rot = _prep(rotations, dt)
    lead, J = rot.shape[:-2], rot.shape[-2]
    F = int(np.prod(lead, dtype=np.int64))
    gp = np.ascontiguousarray(np.broadcast_to(_prep(global_pos, dt), lead + (3,)))
    off = _prep(offsets, dt)
    dq = np.empty(lead + (J, 8), dtype=dt)
    _call("oracle_to_root_dq", dt, rot, gp, _parents(parents), off, _i64(F), _i32(J), dq)
    return dq


def from_root_dual_quat(dq, parents):
    """ops/skeleton.py:173-204 -> (translations, rotations)"""
    dt = _dt(dq)
    dq = _prep(dq, dt)
    lead, J = dq.shape[:-2], dq.shape[-2]
    F = int(np.prod(lead, dtype=np.int64))
    t = np.empty(lead + (J, 3), dtype=dt)
    q = np.empty(lead + (J, 4), dtype=dt)
    _call("oracle_from_root_dq", dt, dq, _parents(parents), _i64(F), _i32(J), t, q)
    return t, q


def fk_from_ortho6d(o6d, global_pos, offsets, parents, eps=0.0, return_quat=False):
    """rotations/ortho6d.py:50-64 then ops/skeleton.py:16-61"""
    dt = _dt(o6d, global_pos, offsets)
    x = _prep(o6d, dt)
    lead, J = x.shape[:-3], x.shape[-3]
    F = int(np.prod(lead, dtype=np.int64))
    gp = np.ascontiguousarray(np.broadcast_to(_prep(global_pos, dt), lead + (3,)))
    off = _prep(offsets, dt)
    per_frame = off.ndim > 2
    pos = np.empty(lead + (J, 3), dtype=dt)
    rm = np.empty(lead + (J, 3, 3), dtype=dt)
    qo = np.empty(lead + (J, 4), dtype=dt) if return_quat else None
    _call("oracle_fk_from_ortho6d", dt, x, gp, off, (C.c_int, int(per_frame)), _parents(parents), _i64(F),
          _i32(J), float(eps), pos, rm, qo)
    return (pos, rm, qo) if return_quat else (pos, rm)


def from_global_rotations(global_quats, parents):
    """ops/skeleton.py:64-93"""
    dt = _dt(global_quats)
    g = _prep(global_quats, dt)
    lead, J = g.shape[:-2], g.shape[-2]
    F = int(np.prod(lead, dtype=np.int64))
    o = np.empty_like(g)
    _call("oracle_from_global_rotations", dt, g, _parents(parents), _i64(F), _i32(J), o)
    return o
