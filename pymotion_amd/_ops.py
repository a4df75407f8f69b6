"""Backend-agnostic bodies of the public functions (NumPy and torch front doors share them).

Every function flattens the reference's ``[..., C]`` / ``[..., J, C]`` shapes to the
``[N, C]`` / ``[F, J, C]`` the C ABI takes, launches ONE kernel from ``libpmhip.so`` and
reshapes back.  Reference citations (``pymotion/...:line``) are on the public wrappers in
``pymotion_amd/rotations`` and ``pymotion_amd/ops``.
"""
import ctypes as C

import numpy as np

from . import _lib

_AXIS = {"x": 0, "y": 1, "z": 2}


def _prod(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def _bshape(*shapes):
    return tuple(np.broadcast_shapes(*shapes))


def _pipe_np(a, lead, trail):
    """host array -> [N, *trail] view (broadcast over `lead` if needed) for the chunked pipeline"""
    a = np.asarray(a)
    if a.shape != tuple(lead) + tuple(trail):
        a = np.broadcast_to(a, tuple(lead) + tuple(trail))
    return a.reshape((-1,) + tuple(trail)) if a.flags.c_contiguous else np.ascontiguousarray(a).reshape((-1,) + tuple(trail))


def ew(be, fname, ins, in_trail, out_trail, out_dtypes, pre=(), mid=()):
    """Generic element-wise launch.

    ins[k] has trailing shape in_trail[k]; leading shapes broadcast against each other.
    C call: fname(in_ptrs..., *pre, N, *mid, out_ptrs..., stream).
    """
    if be.name == "numpy":
        shapes = [np.shape(x) for x in ins]
        if all(tuple(sh[len(sh) - len(t):]) == tuple(t) for sh, t in zip(shapes, in_trail)):
            lead = _bshape(*[sh[: len(sh) - len(t)] for sh, t in zip(shapes, in_trail)])
            n = _prod(lead)
            per = 4 * (sum(_prod(t) for t in in_trail) + sum(_prod(t) for t in out_trail))
            # (an operand that broadcasts over the batch -- one quaternion, one `t` against N records -- would have to be
            # expanded on the host, staged and sent over the bus N times over: those calls take the plain path, whose
            # dev_in expands on the way to the device buffer only)
            full = all(tuple(sh[: len(sh) - len(t)]) == tuple(lead) for sh, t in zip(shapes, in_trail))
            if full and be.wants_pipeline(n, per):
                from ._backend import pipelined_frames

                res = pipelined_frames(
                    n, [(_pipe_np(x, lead, t), True) for x, t in zip(ins, in_trail)],
                    [(tuple(t), np.dtype(dt)) for t, dt in zip(out_trail, out_dtypes)],
                    lambda ip, op, cnt, st: _lib.call(fname, *ip, *pre, cnt, *mid, *op, st))
                res = [r.reshape(tuple(lead) + tuple(t)) for r, t in zip(res, out_trail)]
                return res[0] if len(res) == 1 else tuple(res)
    be.begin(*ins)
    try:
        leads = [be.shape(x)[: len(be.shape(x)) - len(t)] for x, t in zip(ins, in_trail)]
        for x, t in zip(ins, in_trail):
            if tuple(be.shape(x)[len(be.shape(x)) - len(t):]) != tuple(t):
                raise ValueError(f"{fname}: expected trailing shape {tuple(t)}, got array of shape {be.shape(x)}")
        lead = _bshape(*leads)
        n = _prod(lead)
        in_ptrs = [be.dev_in(x, lead + tuple(t)) for x, t in zip(ins, in_trail)]
        outs = [be.dev_out(lead + tuple(t)) for t in out_trail]
        if n > 0:
            _lib.call(fname, *in_ptrs, *pre, n, *mid, *[p for p, _ in outs], be.stream())
        res = [be.result(h, dt) for (_, h), dt in zip(outs, out_dtypes)]
    finally:
        be.end()
    return res[0] if len(res) == 1 else tuple(res)


# ---- quaternions ---------------------------------------------------------------------------------

def quat_normalize(be, q, eps=1e-8):
    return ew(be, "pm_quat_normalize_f32", [q], [(4,)], [(4,)], [be.result_dtype(q)], mid=[C.c_float(eps)])


def quat_length(be, q):
    return ew(be, "pm_quat_length_f32", [q], [(4,)], [()], [be.result_dtype(q)])


def quat_to_matrix(be, q):
    return ew(be, "pm_quat_to_matrix_f32", [q], [(4,)], [(3, 3)], [be.always64])


def quat_from_matrix(be, m):
    return ew(be, "pm_quat_from_matrix_f32", [m], [(3, 3)], [(4,)], [be.result_dtype(m)])


def quat_mul(be, q0, q1):
    return ew(be, "pm_quat_mul_f32", [q0, q1], [(4,), (4,)], [(4,)], [be.result_dtype(q0, q1)])


def quat_mul_vec(be, q, v):
    return ew(be, "pm_quat_mul_vec_f32", [q, v], [(4,), (3,)], [(3,)], [be.result_dtype(q, v)])


def quat_conjugate(be, q):
    return ew(be, "pm_quat_conjugate_f32", [q], [(4,)], [(4,)], [be.result_dtype(q)])


def quat_from_angle_axis(be, angle, axis):
    return ew(be, "pm_quat_from_angle_axis_f32", [angle, axis], [(1,), (3,)], [(4,)], [be.result_dtype(angle, axis)])


def quat_from_scaled_angle_axis(be, v):
    return ew(be, "pm_quat_from_scaled_angle_axis_f32", [v], [(3,)], [(4,)], [be.result_dtype(v)])


def quat_to_angle_axis(be, q):
    dt = be.result_dtype(q)
    return ew(be, "pm_quat_to_angle_axis_f32", [q], [(4,)], [(1,), (3,)], [dt, dt])


def quat_to_scaled_angle_axis(be, q):
    return ew(be, "pm_quat_to_scaled_angle_axis_f32", [q], [(4,)], [(3,)], [be.result_dtype(q)])


def _order_codes(order, lead):
    """The reference takes a NumPy array of 'x'|'y'|'z' strings shaped like euler (quat.py:51-53).
    Encode as uint8 codes and compress what repeats: a constant array becomes one triple (mode 0), an array whose
    rows repeat with the period of the last leading axis -- an order per joint, tiled over the frames, which is what
    `BVH.get_data` and every caller with a BVH-style clip builds -- becomes a [P, 3] table (mode P); anything else
    stays an order per element (mode 1).  Returns (codes, mode) for pm_quat_{from,to}_euler_f32."""
    order = np.asarray(order)
    if order.dtype.kind in "US" and order.dtype.itemsize == (4 if order.dtype.kind == "U" else 1):
        # single-character strings: reinterpret the code points instead of comparing strings three times
        v = np.ascontiguousarray(order).view(np.uint32 if order.dtype.kind == "U" else np.uint8)
        if v.size and (v.min() < 120 or v.max() > 122):  # ord('x') .. ord('z')
            raise ValueError("order entries must be 'x', 'y' or 'z'")
        codes = v.astype(np.uint8)  # (subtracting on the uint32 view would promote to int64: 40x slower)
        codes -= 120
    elif order.dtype.kind in "US":
        codes = np.zeros(order.shape, dtype=np.uint8)
        for ch, v in _AXIS.items():
            codes[order == ch] = v
        if not np.isin(order, list(_AXIS)).all():
            raise ValueError("order entries must be 'x', 'y' or 'z'")
    else:
        codes = order.astype(np.uint8)
    if codes.shape[-1] != 3:
        raise ValueError("order must have a trailing dimension of 3")
    if tuple(codes.shape[:-1]) != tuple(lead):
        # same assertion as the reference (quat.py:60-62)
        raise AssertionError("euler and order must have the same shape except for the last dimension")
    flat = codes.reshape(-1, 3)
    if len(lead) >= 2 and lead[-1] >= 2 and len(flat) > lead[-1]:
        per = flat.reshape(-1, lead[-1] * 3)  # rows = frames: the cheap test first (wide rows compare fast)
        if (per == per[0]).all():
            table = np.ascontiguousarray(per[0].reshape(lead[-1], 3))
            return (table[0].copy(), 0) if (table == table[0]).all() else (table, int(lead[-1]))
    if len(flat) and (flat == flat[0]).all():
        return np.ascontiguousarray(flat[0]), 0
    return np.ascontiguousarray(flat), 1


def _order_table(table, lead):
    """One order per joint, given as a [J, 3] table for euler [..., J, 3] (what a BVH header holds): the compressed
    form of _order_codes without ever materialising the tiled array."""
    codes, _ = _order_codes(np.asarray(table), np.asarray(table).shape[:-1])
    if codes.ndim == 1:
        return codes, 0
    J = codes.shape[0]
    if not lead or lead[-1] != J:
        raise ValueError(f"order table has {J} rows, euler has {lead[-1] if lead else 0} joints")
    return codes, (J if J >= 2 else 0)


def _euler_like(be, fname, x, trail, out_trail, order, out_dtype, per_joint_table=False):
    lead = be.shape(x)[: len(be.shape(x)) - 1]
    codes, per_elem = _order_table(order, lead) if per_joint_table else _order_codes(order, lead)
    be.begin(x)
    try:
        n = _prod(lead)
        xp = be.dev_in(x, tuple(lead) + tuple(trail))
        op = be.dev_in(codes, None, dtype=_u8(be))
        out_p, h = be.dev_out(tuple(lead) + tuple(out_trail))
        if n > 0:
            _lib.call(fname, xp, op, per_elem, n, out_p, be.stream())
        res = be.result(h, out_dtype)
    finally:
        be.end()
    return res


def _u8(be):
    return np.uint8 if be.name == "numpy" else be.torch.uint8


def quat_from_euler(be, euler, order, per_joint_table=False):
    return _euler_like(be, "pm_quat_from_euler_f32", euler, (3,), (4,), order, be.result_dtype(euler), per_joint_table)


def quat_to_euler(be, q, order, per_joint_table=False):
    return _euler_like(be, "pm_quat_to_euler_f32", q, (4,), (3,), order, be.always64, per_joint_table)


def quat_slerp(be, q0, q1, t, shortest=True):
    dt = be.result_dtype(q0, q1)
    if not hasattr(t, "shape") or len(be.shape(t)) == 0:
        t = np.full((1,), float(t), dtype=np.float32)
    return ew(be, "pm_quat_slerp_f32", [q0, q1, t], [(4,), (4,), (1,)], [(4,)], [dt], mid=[int(bool(shortest))])


def _maybe_1d(be, arrs):
    """The reference accepts single vectors [3] and returns a single quaternion (quat.py:531-533, 573-574)."""
    if len(be.shape(arrs[0])) == 1:
        return [a[None] for a in arrs], True
    return list(arrs), False


def quat_from_to(be, v1, v2, normalize_input=True):
    assert be.shape(v1)[-1] == 3 and be.shape(v2)[-1] == 3, "Input vectors must have shape [..., 3]"
    assert be.shape(v1) == be.shape(v2), "Input vectors must have the same shape"
    (v1, v2), single = _maybe_1d(be, (v1, v2))
    r = ew(be, "pm_quat_from_to_f32", [v1, v2], [(3,), (3,)], [(4,)], [be.result_dtype(v1, v2)],
           mid=[int(bool(normalize_input))])
    return r[0] if single else r


def quat_from_to_axis(be, v1, v2, rot_axis, normalize_input=True):
    assert be.shape(v1)[-1] == 3 and be.shape(v2)[-1] == 3, "Input vectors must have shape [..., 3]"
    assert be.shape(v1) == be.shape(v2), "Input vectors must have the same shape"
    assert be.shape(v1) == be.shape(rot_axis), "Input vectors and rotation axis must have the same shape"
    (v1, v2, rot_axis), single = _maybe_1d(be, (v1, v2, rot_axis))
    r = ew(be, "pm_quat_from_to_axis_f32", [v1, v2, rot_axis], [(3,), (3,), (3,)], [(4,)],
           [be.result_dtype(v1, v2, rot_axis)], mid=[int(bool(normalize_input))])
    return r[0] if single else r


def _unroll(be, x, axis, width, fname):
    shp = be.shape(x)
    if len(shp) < 2 or shp[-1] != width:
        raise ValueError(f"expected [..., {width}] with an unroll axis, got {shp}")
    nd = len(shp)
    ax = axis % nd
    if ax == nd - 1:
        raise ValueError("the unroll axis cannot be the component axis")
    dt = be.result_dtype(x)
    B, T, S = _prod(shp[:ax]), shp[ax], _prod(shp[ax + 1:-1])
    if S <= 64 and (B == 1 or T * S * (width // 4) >= 512):  # (tiny clips would leave most of a 1024-dwordx4 tile idle: the wide form below)
        # [B, T, S, W] as it lies: the axes in front of the unroll axis are a batch of independent clips, scanned in one launch --
        # no transposition (a [B, T, J, 4] batch unrolled along T used to be moved to [T, B J, 4] and back: two more full copies)
        be.begin(x)
        try:
            xp = be.dev_in(x)
            op, oh = be.dev_out(shp)
            if B > 0 and T > 0 and S > 0:
                _unroll_scan(be, 0 if width == 4 else 1, xp, None, B, T, S, op, fname.replace("_f32", "_batched_f32"))
            res = be.result(oh, dt)
        finally:
            be.end()
        return res
    moved = be.moveaxis(x, ax, 0)          # unroll axis first; every other index is an independent series
    mshape = be.shape(moved)
    be.begin(x)
    try:
        xp = be.dev_in(moved)
        op, oh = be.dev_out(mshape)
        if T > 0 and S > 0:
            ws = be.scratch(_lib.lib().pm_quat_unroll_workspace_bytes(T, B * S))
            _lib.call(fname, xp, T, B * S, op, ws, be.stream())
        res = be.result(oh, dt)
    finally:
        be.end()
    return be.moveaxis(res, 0, ax)


def _unroll_scan(be, kind, xp, order, B, T, S, op, plain):
    """one one-pass scan (kind 0 quat.unroll, 1 dual_quat.unroll, 2 the BVH ingest; S <= 64) through the backend's workspace pair -- no reset
    launch in front of it (pm_unroll_onepass_f32) -- or, for calls too big for the pair, through the plain entry point `plain`"""
    from . import _backend

    nbytes = _lib.lib().pm_quat_unroll_batched_workspace_bytes(B, T, S)
    pair, key = be.unroll_pair(nbytes)
    if pair is None:
        ws = be.scratch(nbytes)
        if kind == 2:
            _lib.call(plain, xp, order, T, S, op, ws, be.stream())
        else:
            _lib.call(plain, xp, B, T, S, op, ws, be.stream())
        return
    use, other, other_words = pair.take()
    dirtied = C.c_int64(0)
    try:
        _lib.call("pm_unroll_onepass_f32", kind, xp, order, B, T, S, op, use, C.byref(dirtied), other, other_words, be.stream())
    except Exception:
        _backend._unroll_pair_drop(key, pair)  # (whatever state the blocks are in: the next call makes a fresh pair; the NumPy door's blocks are freed)
        raise
    pair.done(dirtied.value)
    _backend._unroll_pair_release(key, pair)


def bvh_rotations(be, euler_deg, order_table):
    """io/bvh.py:352-359 (BVH.get_data): quat.normalize(quat.unroll(quat.from_euler(np.radians(euler_deg), order), axis=0)) for
    euler_deg [T, J, 3] in DEGREES and one Euler order per joint (a [J, 3] table of 'x' | 'y' | 'z' or axis codes).  Up to 64 joints
    this is ONE kernel (pm_bvh_rotations_f32: 12 B in, 16 B out per joint and frame, one trip over the bus through the NumPy
    door); beyond, the three ops on the device."""
    shp = be.shape(euler_deg)
    if len(shp) != 3 or shp[-1] != 3:
        raise ValueError(f"expected [frames, joints, 3] Euler angles, got {shp}")
    T, J = shp[0], shp[1]
    codes, _ = _order_codes(np.asarray(order_table), np.asarray(order_table).shape[:-1])
    table = np.ascontiguousarray(np.broadcast_to(codes.reshape(-1, 3), (J, 3)), dtype=np.uint8)
    if J > 64:
        q = quat_from_euler(be, be.radians(euler_deg), order_table, per_joint_table=True)
        return quat_normalize(be, quat_unroll(be, q, 0))
    dt = be.result_dtype(euler_deg)
    if be.name == "numpy":
        a = np.asarray(euler_deg)
        if a.dtype == np.float64 and a.size and max(a.max(), -a.min()) > 360.0:
            # the file's float64 degrees are rounded to fp32 on their way to the device: bring channels that wound up past a turn back to
            # [-360, 360] first, EXACTLY (a - 720 k with k = rint(a / 720): 720 k is an integer and within a factor of two of a, so the
            # difference is exact in float64; a quaternion is 720-degree periodic in each Euler angle, so not even its sign changes) --
            # the fp32 rounding is then <= 1.5e-5 degrees whatever the channel's winding.  Files whose channels stay within a
            # turn -- nearly all -- pay the two reductions of the test only.
            k = np.rint(a * (1.0 / 720.0))
            k *= 720.0
            euler_deg = a - k
    elif be.name == "torch":
        torch = be.torch
        if euler_deg.dtype == torch.float64 and euler_deg.numel() and float(euler_deg.abs().max()) > 360.0:
            # the same exact wrap for float64 tensors (the two doors then agree on wound-up channels)
            euler_deg = euler_deg - 720.0 * torch.round(euler_deg * (1.0 / 720.0))
    be.begin(euler_deg)
    try:
        xp = be.dev_in(euler_deg)
        op, oh = be.dev_out((T, J, 4))
        if T > 0 and J > 0:
            _unroll_scan(be, 2, xp, table.ctypes.data_as(C.c_void_p), 1, T, J, op, "pm_bvh_rotations_f32")
        res = be.result(oh, dt)
    finally:
        be.end()
    return res


def quat_unroll(be, q, axis):
    return _unroll(be, q, axis, 4, "pm_quat_unroll_f32")


def dq_unroll(be, dq, axis):
    return _unroll(be, dq, axis, 8, "pm_dq_unroll_f32")


# ---- resampling along the time axis (ops/time.py) -------------------------------------------------------

def interpolate_positions(be, sample_times, original_times, positions, axis, method="linear"):
    """ops/time.py:4-66 / ops/time_torch.py.  The reference's final broadcast (time.py:61-64) is only
    well-formed when the time axis is the second to last one; here any axis works (same values there)."""
    if method != "linear":
        raise ValueError("Only linear interpolation is supported yet.")
    shp = be.shape(positions)
    nd = len(shp)
    if nd < 1:
        raise ValueError("positions needs a time axis")
    ax = axis % nd
    T = shp[ax]
    if be.shape(original_times) != (T,):
        raise ValueError("Wrong shape of data. Positions along the axis dimension must be equal to the length of original_times.")
    if len(be.shape(sample_times)) != 1:
        raise ValueError("sample_times must be a 1D array")
    if T < 2:
        raise ValueError("linear interpolation needs at least two original times")
    S = be.shape(sample_times)[0]
    A, B = _prod(shp[:ax]), _prod(shp[ax + 1:])
    be.begin(positions)
    try:
        idx, w = be.interp_coefficients(sample_times, original_times)
        dt = be.result_dtype(w, positions)  # what (1 - weights) * positions promotes to
        pp = be.dev_in(positions)
        ip = be.dev_in(idx, dtype=be.i32)
        wp = be.dev_in(w)
        op, oh = be.dev_out(shp[:ax] + (S,) + shp[ax + 1:])
        if A * S * B > 0:
            _lib.call("pm_interpolate_linear_f32", pp, ip, wp, A, T, S, B, op, be.stream())
        res = be.result(oh, dt)
    finally:
        be.end()
    return res


# ---- dual quaternions ----------------------------------------------------------------------------

def dq_from_rt(be, q, t):
    # NumPy reference: float64 out (dual_quat.py:32); torch twin: rotations.dtype (dual_quat_torch.py:32)
    dt = be.always64 if be.name == "numpy" else q.dtype
    return ew(be, "pm_dq_from_rt_f32", [q, t], [(4,), (3,)], [(8,)], [dt])


def dq_to_rt(be, dq):
    dt = be.result_dtype(dq)
    return ew(be, "pm_dq_to_rt_f32", [dq], [(8,)], [(4,), (3,)], [dt, dt])


def dq_from_t(be, t):
    dt = be.always64 if be.name == "numpy" else t.dtype
    return ew(be, "pm_dq_from_t_f32", [t], [(3,)], [(8,)], [dt])


def _unit_from_flags(flags):
    # dual_quat.py:130-136: all |qr|^2 ~ 0 -> True; else (all ~ 1) and (all qr.qd ~ 0)
    not_zero, not_one, not_orth = flags
    if not_zero == 0:
        return True
    return not_one == 0 and not_orth == 0


def dq_is_unit(be, dq, atol=1e-3):
    shp = be.shape(dq)
    if shp[-1] != 8:
        raise ValueError(f"dq must be [..., 8], got {shp}")
    be.begin(dq)
    try:
        n = _prod(shp[:-1])
        dp = be.dev_in(dq)
        fp, fh = be.flags_alloc()
        if n > 0:
            _lib.call("pm_dq_unit_flags_f32", dp, n, C.c_float(atol), fp, be.stream())
        flags = be.flags_read(fh)
    finally:
        be.end()
    return _unit_from_flags(flags)


def dq_normalize(be, dq):
    shp = be.shape(dq)
    if shp[-1] != 8:
        raise ValueError(f"dq must be [..., 8], got {shp}")
    dt = be.result_dtype(dq)
    be.begin(dq)
    try:
        n = _prod(shp[:-1])
        dp = be.dev_in(dq)
        op, oh = be.dev_out(shp)
        fp, fh = be.flags_alloc()
        if n > 0:
            # first pass: plain division by |qr| and the is_unit verdict on THAT result (dual_quat.py:102-106)
            _lib.call("pm_dq_normalize_f32", dp, n, 0, C.c_float(1e-3), op, fp, be.stream())
            if not _unit_from_flags(be.flags_read(fh)):
                # whole-batch branch of the reference: also make qd orthogonal to qr (:107-113)
                _lib.call("pm_dq_normalize_f32", dp, n, 1, C.c_float(1e-3), op, None, be.stream())
        res = be.result(oh, dt)
    finally:
        be.end()
    return res


# ---- ortho6d -------------------------------------------------------------------------------------------

def o6d_eps(be):
    # NumPy reference divides by the raw norm (ortho6d.py:83-85: zero column -> NaN); the torch
    # twin uses F.normalize(eps=1e-12) (ortho6d_torch.py:84-89: zero column -> zeros).
    return 0.0 if be.name == "numpy" else 1e-12


def o6d_to_matrix(be, x):
    return ew(be, "pm_o6d_to_matrix_f32", [x], [(3, 2)], [(3, 3)], [be.result_dtype(x)], mid=[C.c_float(o6d_eps(be))])


def o6d_to_quat(be, x):
    return ew(be, "pm_o6d_to_quat_f32", [x], [(3, 2)], [(4,)], [be.result_dtype(x)], mid=[C.c_float(o6d_eps(be))])


def o6d_from_quat(be, q):
    return ew(be, "pm_o6d_from_quat_f32", [q], [(4,)], [(3, 2)], [be.always64])


def o6d_from_matrix(be, m):
    return ew(be, "pm_o6d_from_matrix_f32", [m], [(3, 3)], [(3, 2)], [be.result_dtype(m)])


# ---- skeleton ops --------------------------------------------------------------------------------------

def _parents_host(be, parents, J):
    p = be.host_ints(parents)
    if p.ndim != 1 or p.shape[0] != J:
        raise ValueError(f"parents must have shape [{J}], got {p.shape}")
    return p


def fk(be, rot, global_pos, offsets, parents):
    shp = be.shape(rot)
    if len(shp) < 2 or shp[-1] != 4:
        raise ValueError(f"rot must be [..., n_joints, 4], got {shp}")
    lead, J = shp[:-2], shp[-2]
    p = _parents_host(be, parents, J)
    oshape = be.shape(offsets)
    per_frame = len(oshape) > 2
    out_dt = be.always64 if be.name == "numpy" else rot.dtype  # skeleton.py:44 / skeleton_torch.py:45-49
    if be.wants_pipeline(_prod(lead), 4 * (J * (16 + (3 if per_frame else 0)) + 3) ):
        from ._backend import pipelined_frames

        off_np = _pipe_np(offsets, lead, (J, 3)) if per_frame else np.ascontiguousarray(np.asarray(offsets), dtype=np.float32).reshape(J, 3)
        pos, rm = pipelined_frames(
            _prod(lead), [(_pipe_np(rot, lead, (J, 4)), True), (_pipe_np(global_pos, lead, (3,)), True), (off_np, per_frame)],
            [((J, 3), np.dtype(out_dt)), ((J, 3, 3), np.dtype(out_dt))],
            lambda ip, op, n, st: _lib.call("pm_fk_f32", ip[0], ip[1], ip[2], int(per_frame), p.ctypes.data_as(C.c_void_p), n, J, op[0], op[1], st))
        return pos.reshape(lead + (J, 3)), rm.reshape(lead + (J, 3, 3))
    be.begin(rot, global_pos, offsets)
    try:
        F = _prod(lead)
        rp = be.dev_in(rot)
        gp = be.dev_in(global_pos, lead + (3,))
        op = be.dev_in(offsets, lead + (J, 3) if per_frame else (J, 3))
        pos_p, pos_h = be.dev_out(lead + (J, 3))
        rm_p, rm_h = be.dev_out(lead + (J, 3, 3))
        if F > 0:
            _lib.call("pm_fk_f32", rp, gp, op, int(per_frame), p.ctypes.data_as(C.c_void_p), F, J, pos_p, rm_p, be.stream())
        res = be.result(pos_h, out_dt), be.result(rm_h, out_dt)
    finally:
        be.end()
    return res


def fk_from_ortho6d(be, o6d, global_pos, offsets, parents, return_quat=False):
    shp = be.shape(o6d)
    if len(shp) < 3 or shp[-2:] != (3, 2):
        raise ValueError(f"ortho6D must be [..., n_joints, 3, 2], got {shp}")
    lead, J = shp[:-3], shp[-3]
    p = _parents_host(be, parents, J)
    per_frame = len(be.shape(offsets)) > 2
    out_dt = be.always64 if be.name == "numpy" else o6d.dtype
    q_dt = be.result_dtype(o6d)
    if be.wants_pipeline(_prod(lead), 4 * (J * 24 + 3)):
        from ._backend import pipelined_frames

        off_np = _pipe_np(offsets, lead, (J, 3)) if per_frame else np.ascontiguousarray(np.asarray(offsets), dtype=np.float32).reshape(J, 3)
        outs = [((J, 3), np.dtype(out_dt)), ((J, 3, 3), np.dtype(out_dt))] + ([((J, 4), np.dtype(q_dt))] if return_quat else [])
        eps = C.c_float(o6d_eps(be))
        res = pipelined_frames(
            _prod(lead), [(_pipe_np(o6d, lead, (J, 3, 2)), True), (_pipe_np(global_pos, lead, (3,)), True), (off_np, per_frame)], outs,
            lambda ip, op, n, st: _lib.call("pm_fk_from_ortho6d_f32", ip[0], ip[1], ip[2], int(per_frame), p.ctypes.data_as(C.c_void_p), n, J,
                                            eps, op[0], op[1], op[2] if return_quat else None, st))
        out = (res[0].reshape(lead + (J, 3)), res[1].reshape(lead + (J, 3, 3)))
        return out + (res[2].reshape(lead + (J, 4)),) if return_quat else out
    be.begin(o6d, global_pos, offsets)
    try:
        F = _prod(lead)
        xp = be.dev_in(o6d)
        gp = be.dev_in(global_pos, lead + (3,))
        op = be.dev_in(offsets, lead + (J, 3) if per_frame else (J, 3))
        pos_p, pos_h = be.dev_out(lead + (J, 3))
        rm_p, rm_h = be.dev_out(lead + (J, 3, 3))
        q_p, q_h = be.dev_out(lead + (J, 4)) if return_quat else (None, None)
        if F > 0:
            _lib.call("pm_fk_from_ortho6d_f32", xp, gp, op, int(per_frame), p.ctypes.data_as(C.c_void_p), F, J,
                      C.c_float(o6d_eps(be)), pos_p, rm_p, q_p, be.stream())
        res = (be.result(pos_h, out_dt), be.result(rm_h, out_dt))
        if return_quat:
            res = res + (be.result(q_h, q_dt),)
    finally:
        be.end()
    return res


_ROOT_OFFSET_CHECKED = {}


def _assert_root_offset_is_zero(be, offsets):
    """The reference's ``assert (offsets[0] == 0).all()`` (skeleton.py:227 / skeleton_torch.py:242).  On a HIP
    tensor that comparison is a device->host synchronisation per call; the verdict is remembered per tensor OBJECT
    (weak reference, never its address: freed blocks are reused) and in-place version, so a loop over clips of one
    skeleton pays it once and every new tensor is checked."""
    if be.name != "torch" or not getattr(offsets, "is_cuda", False):
        off0 = np.asarray(offsets[0].detach().cpu() if be.name == "torch" else offsets[0])
        assert (off0 == 0).all()
        return
    if be._memo_get(_ROOT_OFFSET_CHECKED, "offsets", offsets):
        return
    assert bool((offsets[0] == 0).all())
    be._memo_put(_ROOT_OFFSET_CHECKED, "offsets", offsets, True)


_OFFSETS_SCALE = {}


def _offsets_abs_max(be, offsets):
    """max |offsets| as a host float: the scale hint of pm_to_root_dq_hint_f32.  The NumPy door has the table on the host; on a HIP
    tensor the reduction is a device->host synchronisation, remembered per tensor object and in-place version like the root-offset
    check above (which already synchronises once per tensor)."""
    if be.name != "torch" or not getattr(offsets, "is_cuda", False):
        a = np.asarray(offsets.detach().cpu() if be.name == "torch" else offsets, dtype=np.float64)
        return float(np.abs(a).max()) if a.size else 0.0
    v = be._memo_get(_OFFSETS_SCALE, "offsets_scale", offsets)
    if v is None:
        v = float(offsets.detach().abs().max())
        be._memo_put(_OFFSETS_SCALE, "offsets_scale", offsets, v)
    return v


def to_root_dual_quat(be, rotations, global_pos, parents, offsets):
    shp = be.shape(rotations)
    if len(shp) < 2 or shp[-1] != 4:
        raise ValueError(f"rotations must be [..., n_joints, 4], got {shp}")
    lead, J = shp[:-2], shp[-2]  # joint axis is -2 (the reference's shape[1] is a latent bug)
    p = _parents_host(be, parents, J)
    if be.shape(offsets) != (J, 3):
        raise ValueError(f"offsets must be [{J}, 3], got {be.shape(offsets)}")
    _assert_root_offset_is_zero(be, offsets)  # skeleton.py:227
    hint = C.c_float(_offsets_abs_max(be, offsets))
    out_dt = be.always64 if be.name == "numpy" else rotations.dtype
    if be.wants_pipeline(_prod(lead), 4 * (J * 12 + 3)):
        from ._backend import pipelined_frames

        off_np = np.ascontiguousarray(np.asarray(offsets), dtype=np.float32)
        (dq,) = pipelined_frames(
            _prod(lead), [(_pipe_np(rotations, lead, (J, 4)), True), (_pipe_np(global_pos, lead, (3,)), True), (off_np, False)],
            [((J, 8), np.dtype(out_dt))],
            lambda ip, op, n, st: _lib.call("pm_to_root_dq_hint_f32", ip[0], ip[1], p.ctypes.data_as(C.c_void_p), ip[2], n, J, op[0], hint, st))
        return dq.reshape(lead + (J, 8))
    be.begin(rotations, global_pos, offsets)
    try:
        F = _prod(lead)
        rp = be.dev_in(rotations)
        gp = be.dev_in(global_pos, lead + (3,))
        op = be.dev_in(offsets)
        dq_p, dq_h = be.dev_out(lead + (J, 8))
        if F > 0:
            _lib.call("pm_to_root_dq_hint_f32", rp, gp, p.ctypes.data_as(C.c_void_p), op, F, J, dq_p, hint, be.stream())
        res = be.result(dq_h, out_dt)
    finally:
        be.end()
    return res


def from_root_dual_quat(be, dq, parents):
    shp = be.shape(dq)
    if len(shp) < 2 or shp[-1] != 8:
        raise ValueError(f"dq must be [..., n_joints, 8], got {shp}")
    lead, J = shp[:-2], shp[-2]
    p = _parents_host(be, parents, J)
    dt = be.result_dtype(dq)
    if be.wants_pipeline(_prod(lead), 4 * J * 15):
        from ._backend import pipelined_frames

        t, q = pipelined_frames(
            _prod(lead), [(_pipe_np(dq, lead, (J, 8)), True)], [((J, 3), np.dtype(dt)), ((J, 4), np.dtype(dt))],
            lambda ip, op, n, st: _lib.call("pm_from_root_dq_f32", ip[0], p.ctypes.data_as(C.c_void_p), n, J, op[0], op[1], st))
        return t.reshape(lead + (J, 3)), q.reshape(lead + (J, 4))  # (translations, rotations): skeleton.py:204
    be.begin(dq)
    try:
        F = _prod(lead)
        dp = be.dev_in(dq)
        t_p, t_h = be.dev_out(lead + (J, 3))
        q_p, q_h = be.dev_out(lead + (J, 4))
        if F > 0:
            _lib.call("pm_from_root_dq_f32", dp, p.ctypes.data_as(C.c_void_p), F, J, t_p, q_p, be.stream())
        res = be.result(t_h, dt), be.result(q_h, dt)  # (translations, rotations): skeleton.py:204
    finally:
        be.end()
    return res


def from_global_rotations(be, global_quats, parents):
    shp = be.shape(global_quats)
    if len(shp) < 2 or shp[-1] != 4:
        raise ValueError(f"global_quats must be [..., n_joints, 4], got {shp}")
    lead, J = shp[:-2], shp[-2]
    p = _parents_host(be, parents, J)
    dt = be.result_dtype(global_quats)
    be.begin(global_quats)
    try:
        F = _prod(lead)
        gp = be.dev_in(global_quats)
        o_p, o_h = be.dev_out(lead + (J, 4))
        if F > 0:
            _lib.call("pm_from_global_rotations_f32", gp, p.ctypes.data_as(C.c_void_p), F, J, o_p, be.stream())
        res = be.result(o_h, dt)
    finally:
        be.end()
    return res


def from_root_positions(be, positions, parents, offsets):
    shp = be.shape(positions)
    if len(shp) < 2 or shp[-1] != 3:
        raise ValueError(f"positions must be [..., n_joints, 3], got {shp}")
    lead, J = shp[:-2], shp[-2]
    p = _parents_host(be, parents, J)
    if be.shape(offsets) != (J, 3):
        raise ValueError(f"offsets must be [{J}, 3], got {be.shape(offsets)}")
    out_dt = be.always64 if be.name == "numpy" else positions.dtype  # skeleton.py:128 builds float64 identities
    be.begin(positions, offsets)
    try:
        F = _prod(lead)
        pp = be.dev_in(positions)
        op_ = be.dev_in(offsets)
        rp, rh = be.dev_out(lead + (J, 4))
        if F > 0:
            _lib.call("pm_from_root_positions_f32", pp, p.ctypes.data_as(C.c_void_p), op_, F, J, rp, be.stream())
        res = be.result(rh, out_dt)
    finally:
        be.end()
    return res


_MIRROR_AXIS = {"X": 0, "Y": 1, "Z": 2}


def mirror(be, local_rotations, global_translation, parents, offsets, end_sites=None, joints_mapping=None,
           mode="all", axis="X"):
    if mode not in ("all", "symmetry", "positions"):
        raise ValueError("Invalid mode. Choose 'symmetry', 'all', or 'positions'")
    if axis not in _MIRROR_AXIS:
        raise ValueError("Invalid axis. Choose 'X', 'Y', or 'Z'")
    if mode == "positions":
        # skeleton.py:332-341: true mirror -> fk -> root-centred positions -> IK on the ORIGINAL skeleton
        m_rots, m_gpos, m_offsets, _ = mirror(be, local_rotations, global_translation, parents, offsets, end_sites, None, "all", axis)
        pos, _ = fk(be, m_rots, m_gpos, m_offsets, parents)
        pos = pos - pos[..., 0:1, :]
        return from_root_positions(be, pos, parents, offsets), m_gpos, offsets, end_sites
    shp = be.shape(local_rotations)
    if len(shp) < 2 or shp[-1] != 4:
        raise ValueError(f"local_rotations must be [..., n_joints, 4], got {shp}")
    lead, J = shp[:-2], shp[-2]
    p = _parents_host(be, parents, J)
    mp = None
    if mode == "symmetry":
        if joints_mapping is None:
            raise ValueError("joints_mapping must be provided for mode 'symmetry'")
        if len(joints_mapping) != J:
            raise ValueError("joints_mapping must have the same length as the number of joints")
        mp = be.host_ints(joints_mapping, slot="joints_mapping")
    ax = _MIRROR_AXIS[axis]
    # world rotations come out of fk (float64 through the NumPy door, rot.dtype through torch), then
    # from_matrix / from_global_rotations keep that dtype
    out_dt = be.always64 if be.name == "numpy" else local_rotations.dtype
    be.begin(local_rotations)
    try:
        F = _prod(lead)
        rp = be.dev_in(local_rotations)
        op, oh = be.dev_out(shp)
        if F > 0:
            _lib.call("pm_mirror_rotations_f32", rp, p.ctypes.data_as(C.c_void_p),
                      mp.ctypes.data_as(C.c_void_p) if mp is not None else None, ax, F, J, op, be.stream())
        rots = be.result(oh, out_dt)
    finally:
        be.end()
    # sign flips of the translation (and, for the true mirror, of the skeleton itself): new arrays -- the
    # reference's 'symmetry' mode writes into the caller's global_translation (skeleton.py:325)
    clone = (lambda t: t.copy()) if be.name == "numpy" else (lambda t: t.clone())
    gt = clone(global_translation)
    gt[..., ax] = -gt[..., ax]
    if mode == "all":
        offsets = clone(offsets)
        offsets[:, ax] = -offsets[:, ax]
        if end_sites is not None:
            end_sites = clone(end_sites)
            end_sites[:, ax] = -end_sites[:, ax]
    return rots, gt, offsets, end_sites
