"""Property-based GPU parity (hypothesis, derandomised): random skeleton sizes / topologies / leading
dimensions / dtypes / memory layouts through both Python doors against the CPU oracle.
Complements the fixed-size cases of test_gpu_parity.py; every example is small, the value is in the mix
(J on both sides of every kernel-shape switch: 23|24 joints, 64|65, odd joint counts, F not a multiple of
any tile, stars and chains, strided views, per-frame offsets, half / bfloat16 / float64 tensors)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from conftest import assert_close
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

import pymotion_amd.ops.skeleton as sk  # noqa: E402
import pymotion_amd.rotations.quat as quat  # noqa: E402

# PM_FUZZ_SCALE=10 PM_FUZZ_RANDOM=1: a one-off long session with fresh examples (the committed runs are derandomised)
_SCALE = int(os.environ.get("PM_FUZZ_SCALE", "1"))
_DERAND = os.environ.get("PM_FUZZ_RANDOM") != "1"
FUZZ = settings(max_examples=40 * _SCALE, deadline=None, derandomize=_DERAND)
f64 = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731


@st.composite
def skeletons(draw):
    J = draw(st.one_of(st.integers(1, 30), st.integers(60, 70), st.sampled_from([22, 23, 24, 52, 64, 65, 129])))
    kind = draw(st.sampled_from(["random", "chain", "star", "bfs"]))
    seed = draw(st.integers(0, 2**16))
    rng = np.random.default_rng(seed)
    if kind == "chain":
        par = np.maximum(np.arange(J) - 1, 0)
    elif kind == "star":
        par = np.zeros(J, dtype=np.int64)
    elif kind == "bfs":
        par = np.maximum((np.arange(J) - 1) // 2, 0)  # binary heap order: almost no joint follows its parent
    else:
        par = np.array([0] + [rng.integers(0, i) for i in range(1, J)])
    lead = draw(st.sampled_from([(0,), (1,), (3,), (17,), (2, 5), (61,), (3, 1, 7), (130,)]))
    return J, par.astype(np.int32), lead, rng


def _layout(rng, a, how):
    """the same values in a different memory layout / dtype"""
    if how == "f64":
        return a.astype(np.float64)
    if how == "strided":
        big = np.zeros(a.shape[:-1] + (2 * a.shape[-1],), dtype=a.dtype)
        big[..., ::2] = a
        return big[..., ::2]
    if how == "fortran":
        return np.asfortranarray(a)
    return a


@FUZZ
@given(skeletons(), st.sampled_from(["c", "f64", "strided", "fortran"]), st.booleans())
def test_fuzz_fk_numpy_door(sk_, how, per_frame_offsets):
    J, par, lead, rng = sk_
    rot = rng.standard_normal(lead + (J, 4)).astype(np.float32)
    gpos = rng.uniform(-2, 2, lead + (3,)).astype(np.float32)
    off = rng.uniform(-0.2, 0.2, (lead + (J, 3)) if per_frame_offsets else (J, 3)).astype(np.float32)
    pos, rm = sk.fk(_layout(rng, rot, how), gpos, off, par)
    F = int(np.prod(lead))
    off_o = f64(off).reshape((F, J, 3) if per_frame_offsets else (J, 3))
    p_o, r_o = co.fk(f64(rot).reshape(F, J, 4), f64(gpos).reshape(F, 3), off_o, par)
    assert pos.shape == lead + (J, 3) and rm.shape == lead + (J, 3, 3) and pos.dtype == np.float64
    tol = 1e-5 * max(1.0, J / 32)  # error grows with chain depth
    assert_close(pos.reshape(F, J, 3), p_o, tol, "pos")
    assert_close(rm.reshape(F, J, 3, 3), r_o, tol, "rotmats")


@FUZZ
@given(skeletons())
def test_fuzz_dual_quat_round_trip_and_oracle(sk_):
    J, par, lead, rng = sk_
    F = int(np.prod(lead))
    rot = rng.standard_normal(lead + (J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    gpos = rng.uniform(-2, 2, lead + (3,)).astype(np.float32)
    off = rng.uniform(-0.2, 0.2, (J, 3)).astype(np.float32)
    off[0] = 0
    d = sk.to_root_dual_quat(rot, gpos, par, off)
    d_o = co.to_root_dual_quat(f64(rot).reshape(F, J, 4), f64(gpos).reshape(F, 3), par, f64(off))
    tol = 1e-5 * max(1.0, J / 32)
    assert_close(d.reshape(F, J, 8), d_o, tol, "to_root_dq")
    t, q = sk.from_root_dual_quat(d, par)
    assert t.shape == lead + (J, 3) and q.shape == lead + (J, 4)
    assert_close(q, rot, 4 * tol, "round trip rot")
    if F:
        assert_close(t[..., 1:, :], np.broadcast_to(off[1:], lead + (J - 1, 3)), 4 * tol, "round trip offsets")
        assert_close(t[..., 0, :], gpos, 4 * tol, "round trip root")
    g = sk.from_global_rotations(rot, par)
    assert_close(g.reshape(F, J, 4), co.from_global_rotations(f64(rot).reshape(F, J, 4), par), 1e-5, "from_global_rotations")


@FUZZ
@given(skeletons(), st.sampled_from(["X", "Y", "Z"]))
def test_fuzz_mirror_twice_is_identity(sk_, axis):
    J, par, lead, rng = sk_
    if len(lead) != 1 or lead[0] == 0:
        lead = (9,)
    rot = rng.standard_normal(lead + (J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    root = rng.uniform(-1, 1, lead + (3,)).astype(np.float32)
    off = rng.uniform(-0.2, 0.2, (J, 3)).astype(np.float32)
    r1, g1, o1, _ = sk.mirror(rot, root, par, off, None, None, "all", axis)
    r2, g2, o2, _ = sk.mirror(r1, g1, par, o1, None, None, "all", axis)
    tol = 2e-5 * max(1.0, J / 32)
    assert np.minimum(np.abs(r2 - rot).max(-1), np.abs(r2 + rot).max(-1)).max() <= tol
    assert_close(g2, root, 1e-7)
    assert_close(o2, off, 1e-7)
    # and the mirrored pose is the mirror image: same fk positions up to the flipped coordinate
    p0, _ = sk.fk(rot, root, off, par)
    p1, _ = sk.fk(r1, g1, o1, par)
    flip = np.ones(3)
    flip["XYZ".index(axis)] = -1
    assert_close(p1, p0 * flip, 2e-5 * max(1.0, J / 8), "mirrored positions")


@FUZZ
@given(st.sampled_from([((5, 4), (4,)), ((3, 1, 4), (1, 6, 4)), ((7, 4), (7, 4)), ((4,), (2, 3, 4)), ((0, 4), (4,))]),
       st.integers(0, 2**16))
def test_fuzz_elementwise_broadcasting(shapes, seed):
    """binary element-wise ops broadcast like NumPy does for the reference (quat.py:337-361, :320-334)"""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shapes[0]).astype(np.float32)
    b = rng.standard_normal(shapes[1]).astype(np.float32)
    want = co.quat_mul(*np.broadcast_arrays(f64(a), f64(b)))
    assert_close(quat.mul(a, b), want, 1e-5, "mul")
    v = rng.standard_normal(shapes[1][:-1] + (3,)).astype(np.float32)
    qa, vv = np.broadcast_arrays(f64(a)[..., :1], f64(v)[..., :1])
    lead = qa.shape[:-1]
    want = co.quat_mul_vec(np.broadcast_to(f64(a), lead + (4,)), np.broadcast_to(f64(v), lead + (3,)))
    assert_close(quat.mul_vec(a, v), want, 1e-5, "mul_vec")


@settings(max_examples=24 * _SCALE, deadline=None, derandomize=_DERAND)
@given(skeletons(), st.sampled_from(["float16", "bfloat16", "float32", "float64"]), st.sampled_from(["cuda", "cpu"]))
def test_fuzz_torch_door_dtypes_and_devices(sk_, dtype, device):
    """the torch door computes in fp32 and returns rot.dtype where the input lives (skeleton_torch.py:45-49)"""
    import torch

    import pymotion_amd.ops.skeleton_torch as skt

    J, par, lead, rng = sk_
    F = int(np.prod(lead))
    dt = getattr(torch, dtype)
    rot = torch.from_numpy(rng.standard_normal(lead + (J, 4)).astype(np.float32)).to(device=device, dtype=dt)
    gpos = torch.from_numpy(rng.uniform(-2, 2, lead + (3,)).astype(np.float32)).to(device=device, dtype=dt)
    off = torch.from_numpy(rng.uniform(-0.2, 0.2, (J, 3)).astype(np.float32)).to(device=device, dtype=dt)
    pos, rm = skt.fk(rot, gpos, off, torch.from_numpy(par))
    assert pos.dtype == dt and rm.dtype == dt and pos.device.type == device and tuple(pos.shape) == lead + (J, 3)
    # the oracle sees exactly the (rounded) values the kernel was given
    p_o, r_o = co.fk(rot.double().cpu().numpy().reshape(F, J, 4), gpos.double().cpu().numpy().reshape(F, 3),
                     off.double().cpu().numpy(), par)
    eps = {"float16": 1e-3, "bfloat16": 8e-3, "float32": 1e-5, "float64": 1e-5}[dtype]
    scale = max(1.0, float(np.abs(p_o).max())) if F else 1.0
    assert_close(pos.double().cpu().numpy().reshape(F, J, 3), p_o, eps * scale * max(1.0, J / 32), "pos")
    assert_close(rm.double().cpu().numpy().reshape(F, J, 3, 3), r_o, eps * max(1.0, J / 32), "rotmats")


# ---- more of the surface: fused ortho6d, element-wise ops on odd sizes / layouts, frame-coupled ops ----------

@FUZZ
@given(skeletons(), st.booleans(), st.booleans())
def test_fuzz_fk_from_ortho6d(sk_, per_frame_offsets, want_quat):
    _check_fk_from_ortho6d(sk_, per_frame_offsets, want_quat)


@settings(max_examples=60 * _SCALE, deadline=None, derandomize=_DERAND)
@given(st.data(), st.booleans(), st.booleans())
def test_fuzz_fk_from_ortho6d_on_wide_skeletons(data, per_frame_offsets, want_quat):
    """the fused ortho6d source on 24 ... 512 joints: the pipelined tiles' four-joints-a-step walk, the wave-per-frame walk beyond 128 joints
    (and where it goes first below), the four-frame tiles for what both decline, the two-launch path at 511 / 512 joints"""
    _check_fk_from_ortho6d(data.draw(wide_skeletons()), per_frame_offsets, want_quat)


def _check_fk_from_ortho6d(sk_, per_frame_offsets, want_quat):
    import pymotion_amd.rotations.ortho6d as o6

    J, par, lead, rng = sk_
    F = int(np.prod(lead))
    x = rng.standard_normal(lead + (J, 3, 2)).astype(np.float32)
    gpos = rng.uniform(-2, 2, lead + (3,)).astype(np.float32)
    off = rng.uniform(-0.2, 0.2, (lead + (J, 3)) if per_frame_offsets else (J, 3)).astype(np.float32)
    out = sk.fk_from_ortho6d(x, gpos, off, par, return_quat=want_quat)
    q_chain = o6.to_quat(x)                      # the two-launch chain through the same library
    p_c, r_c = sk.fk(q_chain, gpos, off, par)
    # Gram-Schmidt on near-parallel columns is ill-conditioned: compare where the chain itself is stable
    x64 = f64(x).reshape(F, J, 3, 2)
    a, b = x64[..., 0], x64[..., 1]
    sin2 = 1 - (np.einsum("fjk,fjk->fj", a, b) ** 2) / (np.einsum("fjk,fjk->fj", a, a) * np.einsum("fjk,fjk->fj", b, b) + 1e-30)
    ok = (sin2.min(axis=1) > 1e-2) if F else np.zeros(0, bool)
    tol = 2e-5 * max(1.0, J / 16)
    assert_close(out[0].reshape(F, J, 3)[ok], p_c.reshape(F, J, 3)[ok], tol, "fused vs chain pos")
    assert_close(out[1].reshape(F, J, 3, 3)[ok], r_c.reshape(F, J, 3, 3)[ok], tol, "fused vs chain rotmats")
    if want_quat:
        assert_close(out[2].reshape(F, J, 4)[ok], q_chain.reshape(F, J, 4)[ok], 1e-5, "fused quats")
    if F:
        assert_close(out[0][..., 0, :], gpos, 0, "root position is global_pos, bit for bit")


@FUZZ
@given(st.integers(0, 700), st.sampled_from(["c", "f64", "strided", "fortran"]), st.integers(0, 2**16))
def test_fuzz_elementwise_sizes_and_layouts(n, how, seed):
    import pymotion_amd.rotations.dual_quat as dq
    import pymotion_amd.rotations.ortho6d as o6

    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    qu = q / (np.linalg.norm(q, axis=-1, keepdims=True) + 1e-30)
    t = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    L = lambda a: _layout(rng, a, how)  # noqa: E731
    assert_close(quat.normalize(L(q)), co.quat_normalize(f64(q)), 1e-5, "normalize")
    assert_close(quat.to_matrix(L(qu)), co.quat_to_matrix(f64(qu)), 1e-5, "to_matrix")
    m = co.quat_to_matrix(f64(qu)).astype(np.float32)
    got, want = quat.from_matrix(L(m)), co.quat_from_matrix(f64(m))
    if n:
        assert np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max() <= 1e-5
    assert_close(quat.mul_vec(L(qu), L(t)), co.quat_mul_vec(f64(qu), f64(t)), 1e-5, "mul_vec")
    d = dq.from_rotation_translation(L(qu), L(t))
    assert_close(d, co.dq_from_rt(f64(qu), f64(t)), 1e-5, "dq.from_rt")
    r2, t2 = dq.to_rotation_translation(d)
    assert_close(r2, qu, 1e-5, "dq round trip rot")
    assert_close(t2, t, 2e-5, "dq round trip trans")
    x = rng.standard_normal((n, 3, 2)).astype(np.float32)
    mm = o6.to_matrix(L(x))
    if n:
        good = np.abs(np.linalg.det(f64(mm)) - 1) < 1e-3
        assert good.mean() > 0.9 and np.abs(mm @ np.swapaxes(mm, -1, -2) - np.eye(3))[good].max() < 1e-4
    assert_close(o6.from_quat(L(qu)), co.o6d_from_quat(f64(qu)), 1e-5, "o6d.from_quat")


@FUZZ
@given(st.sampled_from([(1, 4), (2, 4), (64, 4), (65, 3, 4), (300, 22, 4), (7, 2, 3, 4), (257, 1, 4)]), st.integers(0, 2),
       st.integers(0, 2**16))
def test_fuzz_unroll_any_axis(shape, axis, seed):
    axis = axis % (len(shape) - 1)
    rng = np.random.default_rng(seed)
    q = rng.standard_normal(shape).astype(np.float32)
    # a smooth series along the axis with random sign flips sprinkled in: unroll must remove exactly those
    base = np.cumsum(rng.standard_normal(shape) * 0.05, axis=axis) + rng.standard_normal(shape[-1])
    base /= np.linalg.norm(base, axis=-1, keepdims=True)
    flips = rng.random(shape[:-1]) < 0.3
    q = (base * np.where(flips, -1.0, 1.0)[..., None]).astype(np.float32)
    got = quat.unroll(q, axis)
    assert_close(got, co.quat_unroll(f64(q), axis), 0, "unroll is sign flips only: exact")
    d = np.sum(np.take(got, range(1, shape[axis]), axis) * np.take(got, range(0, shape[axis] - 1), axis), axis=-1)
    assert (d >= 0).all()


@settings(max_examples=150 * _SCALE, deadline=None, derandomize=_DERAND)
@given(st.one_of(st.integers(1, 300), st.integers(300, 40_000)), st.integers(1, 64), st.sampled_from([4, 8]), st.integers(0, 2**16))
def test_fuzz_unroll_clips_one_pass_against_the_sequential_oracle(T, S, W, seed):
    """clips of 1..64 series (the look-back scan: 1 to ~600 chained tiles, one to three status words per tile), zero rows,
    NaN rows and exactly orthogonal steps sprinkled in: bit-for-bit the reference's loop"""
    rng = np.random.default_rng(seed)
    T = min(T, 2_000_000 // (S * W))
    base = np.cumsum(rng.normal(0, 0.08, (T, S, W)), axis=0) + rng.normal(0, 1, (1, S, W))
    q = (base * rng.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)
    for _ in range(int(rng.integers(0, 6))):
        t, s_ = int(rng.integers(0, T)), int(rng.integers(0, S))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            q[t, s_] = 0.0
        elif kind == 1:
            q[t, s_, :4] = np.nan
        elif t > 0:  # exactly orthogonal to its predecessor: dot == 0
            q[t - 1, s_, :4] = (1.0, 0.0, 0.0, 0.0)
            q[t, s_, :4] = (0.0, 1.0, 0.0, 0.0)
    with np.errstate(all="ignore"):
        ref = co.quat_unroll(f64(q[..., :4]), 0)
    if W == 4:
        got = quat.unroll(q, 0)
        assert (np.isnan(got) == np.isnan(ref)).all()
        np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(ref).astype(np.float32))
    else:
        import pymotion_amd.rotations.dual_quat as dq

        got = dq.unroll(q, 0)
        flipped = np.signbit(np.nan_to_num(ref[..., 0])) != np.signbit(np.nan_to_num(q[..., 0]))
        flipped |= (np.nan_to_num(ref[..., 1]) != np.nan_to_num(q[..., 1]))
        want = np.where(flipped[..., None], -q, q)
        ok = ~np.isnan(q[..., :4]).any(axis=-1) & (q[..., :4] != 0).any(axis=-1)
        np.testing.assert_array_equal(got[ok], want[ok])
        np.testing.assert_array_equal(np.abs(np.nan_to_num(got)), np.abs(np.nan_to_num(q)))


@settings(max_examples=60 * _SCALE, deadline=None, derandomize=_DERAND)
@given(st.integers(1, 300), st.one_of(st.integers(1, 80), st.integers(80, 6000)), st.integers(1, 64), st.integers(0, 2**16))
def test_fuzz_unroll_batches_of_clips(B, T, S, seed):
    """[B, T, S, 4] along T: B independent look-back chains in one launch (single-tile clips, chained clips, every tile size
    the dispatch picks), a few resets sprinkled in -- bit for bit the sequential oracle along that axis"""
    rng = np.random.default_rng(seed)
    T = max(1, min(T, 400_000 // (B * S)))
    base = np.cumsum(rng.normal(0, 0.08, (B, T, S, 4)), axis=1) + rng.normal(0, 1, (B, 1, S, 4))
    q = (base * rng.choice([-1.0, 1.0], (B, T, S, 1))).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):
        q[int(rng.integers(0, B)), int(rng.integers(0, T)), int(rng.integers(0, S))] = 0.0
    np.testing.assert_array_equal(quat.unroll(q, 1), co.quat_unroll(f64(q), 1).astype(np.float32))


@FUZZ
@given(st.sampled_from([(3, 3), (50, 22, 3), (2, 9, 5, 3), (40, 4), (17, 1), (6, 2, 2)]), st.integers(0, 3), st.integers(0, 2**16))
def test_fuzz_interpolate_any_axis(shape, axis, seed):
    import pymotion_amd.ops.time as tm

    axis = axis % len(shape)
    if shape[axis] < 2:
        axis = 0
    rng = np.random.default_rng(seed)
    Tn = shape[axis]
    orig = np.cumsum(rng.uniform(0.01, 1.0, Tn))
    sample = rng.uniform(orig[0] - 0.5, orig[-1] + 0.5, rng.integers(0, 40))
    p = rng.uniform(-2, 2, shape).astype(np.float32)
    got = tm.interpolate_positions(sample, orig, p, axis)
    want = co.interpolate_positions(sample, orig, f64(p), axis)
    assert got.dtype == np.float64
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    assert_close(got, want, 2e-6 * scale, "interpolate")


@FUZZ
@given(skeletons())
def test_fuzz_from_root_positions_reproduces_the_pose(sk_):
    """positions -> rotations -> fk must give the positions back wherever the bone lengths are consistent
    (they are: the positions come from fk with the same offsets); ill-conditioned bones are judged in bulk"""
    J, par, lead, rng = sk_
    if len(lead) != 1 or lead[0] < 8:
        lead = (11,)  # (the statements below are medians over the frames: a randomised run of round 6 drew ONE frame of a 69-joint chain with a
                      # near anti-parallel alignment in it -- 6.5e-4 against a bar of 2.9e-4 "in bulk")
    rot = rng.standard_normal(lead + (J, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=-1, keepdims=True)
    off = rng.uniform(0.05, 0.3, (J, 3)).astype(np.float32) * rng.choice([-1.0, 1.0], (J, 3)).astype(np.float32)
    off[0] = 0
    zero = np.zeros(lead + (3,), np.float32)
    pos, _ = sk.fk(rot, zero, off, par)
    r = sk.from_root_positions(pos.astype(np.float32), par, off)
    r_or = co.from_root_positions(f64(pos.astype(np.float32)), par, f64(off))
    d = np.minimum(np.abs(r - r_or).max(-1), np.abs(r + r_or).max(-1))
    # quaternion by quaternion against the oracle: equal in bulk; where a further child lies close to the roll axis the roll is
    # decided by digits fp32 does not have (a tree with five children on one joint: 3 % of its quaternions off by 1e-3..4e-3,
    # found by a randomised run; a near anti-parallel alignment deep in a 60-chain: one quaternion off by 0.47) -- those may differ
    dep = np.zeros(J, int)
    for j in range(1, J):
        dep[j] = dep[par[j]] + 1
    thr = 1e-3 * max(1.0, dep.max() / 8.0)  # (the world rotations the local ones are derived from are fp32 products down the chain)
    assert np.median(d) <= 1e-5 * max(1.0, dep.max() / 8.0), (np.median(d), d.max())
    assert d.size < 200 or (d > thr).mean() < 0.06, ((d > thr).mean(), d.max())  # (a fraction of 30 quaternions says nothing)
    # Fed back through fk the rotations give the positions back -- on chains.  (Not where a joint has several children: the
    # reference's roll about the first child's direction does not in general bring the further children home -- its own
    # result misses them by 0.18 on a three-joint star with these bone lengths, and the kernel reproduces the reference.)
    if J < 2 or np.bincount(par[1:], minlength=J).max() <= 1:
        pos2, _ = sk.fk(r, zero, off, par)
        # (every alignment is good to ~1e-6 rad, and a joint k levels up moves the end of the chain by that times the distance;
        # a frame with a near anti-parallel alignment somewhere -- axis = a x b with |a x b| ~ 1e-3 -- is off by more: in bulk)
        e_f = np.abs(pos2 - pos).reshape(-1, J * 3).max(axis=1)
        bar = 2e-6 * J * max(1.0, float(np.abs(pos).max()))
        assert np.median(e_f) <= bar and e_f.max() < 0.05, (np.median(e_f), (e_f > bar).mean(), e_f.max(), bar)


@settings(max_examples=200 * _SCALE, deadline=None, derandomize=_DERAND)
@given(skeletons(), st.sampled_from([0.05, 0.9, 1.0, 7.0, 30.0, 4.0e3, 2.5e6]), st.sampled_from([0.0, 3.0, 16.0, 900.0, 1.0e7]), st.booleans())
def test_fuzz_fk_at_every_magnitude(sk_, bone_scale, root_scale, per_frame_offsets):
    _check_fk_at_a_magnitude(sk_, bone_scale, root_scale, per_frame_offsets)


@st.composite
def wide_skeletons(draw):
    """24 ... 512 joints, trees wider than deep in several ways (what round 5's joint-parallel fk walks and the eight / sixteen-chain
    to_root_dual_quat take) next to ones that must be declined (windowed parents: deep; chains)"""
    J = draw(st.one_of(st.integers(24, 128), st.integers(129, 512), st.sampled_from([52, 92, 93, 100, 101, 104, 128, 129, 192, 250, 251, 511, 512])))
    kind = draw(st.sampled_from(["random", "random", "bfs", "star", "broom", "fingers", "window8", "chain2"]))
    seed = draw(st.integers(0, 2**16))
    rng = np.random.default_rng(seed)
    if kind == "random":
        par = np.array([0] + [rng.integers(0, i) for i in range(1, J)])
    elif kind == "bfs":
        k = int(rng.integers(2, 6))
        par = np.maximum((np.arange(J) - 1) // k, 0)
    elif kind == "star":
        par = np.zeros(J, dtype=np.int64)
    elif kind == "broom":
        h = int(rng.integers(2, 30))
        par = np.maximum(np.arange(J) - 1, 0)
        par[h:] = h - 1
    elif kind == "fingers":  # a short trunk with chains of three hanging off random trunk joints
        t = int(rng.integers(3, 20))
        par = list(np.maximum(np.arange(t) - 1, 0))
        while len(par) < J:
            a = int(rng.integers(0, t))
            for i in range(3):
                if len(par) < J:
                    par.append(a if i == 0 else len(par) - 1)
        par = np.array(par)
    elif kind == "window8":
        par = np.array([0] + [rng.integers(max(0, i - 8), i) for i in range(1, J)])
    else:
        par = np.maximum(np.arange(J) - 1, 0)
        par[J // 2] = 0
    lead = draw(st.sampled_from([(1,), (3,), (4,), (5,), (17,), (2, 5), (61,), (130,)]))
    return J, par.astype(np.int32), lead, rng


@settings(max_examples=120 * _SCALE, deadline=None, derandomize=_DERAND)
@given(wide_skeletons(), st.sampled_from([0.05, 0.9, 1.0, 30.0, 4.0e3]), st.sampled_from([0.0, 3.0, 16.0, 900.0]), st.booleans())
def test_fuzz_fk_joint_parallel_walks_at_every_magnitude(sk_, bone_scale, root_scale, per_frame_offsets):
    """the same statement on the skeletons of round 5's walks (four joints of a frame at a time in the pipelined kernel, a wave per frame with its
    lanes over the joints): whichever kernel the table takes"""
    _check_fk_at_a_magnitude(sk_, bone_scale, root_scale, per_frame_offsets)


@FUZZ
@given(wide_skeletons(), st.sampled_from([0.3, 30.0]))
def test_fuzz_dual_quat_on_wide_skeletons(sk_, scale):
    """to_root_dual_quat / from_root_dual_quat on the same skeletons (two ... sixteen chains per frame, the lane-per-frame kernels at test sizes
    decline most of them): the oracle's value within max(1e-5, 3 ulp of the largest component), and the round trip"""
    J, par, lead, rng = sk_
    rot = rng.standard_normal(lead + (J, 4))
    rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
    gpos = (rng.uniform(-7, 7, lead + (3,)) * scale).astype(np.float32)
    off = (rng.uniform(-1, 1, (J, 3)) * scale).astype(np.float32)
    off[0] = 0
    d = sk.to_root_dual_quat(rot, gpos, par, off)
    F = int(np.prod(lead))
    d_o = co.to_root_dual_quat(f64(rot).reshape(F, J, 4), f64(gpos).reshape(F, 3), par, f64(off))
    ulp = 2.0 ** (np.floor(np.log2(np.abs(d_o).max())) - 23)
    dd = np.zeros(J, int)
    for j in range(1, J):
        dd[j] = dd[par[j]] + 1
    # (chains: randomised runs of round 5 read 3.6 ulp on a 65-deep chain of 30-unit bones and 4.1 ulp on a 55-deep one, of round 6 3.4 ulp on a 32-deep one
    # -- the precise step rotated its bones in fp32, one rounding a joint.  From twelve levels on it now rotates them in float64 and scales its
    # fixed-point words to the range they have (dq.hip: dq_step_rot_f64, fx_scale_exact): 1.8 ulp on those chains; a search over 6000 skeletons up to
    # 512 joints read at most 2.7 ulp up to 95 levels, 3.9 up to 191, 6.2 at 256.  INTEGRATION.md)
    assert np.abs(d.reshape(F, J, 8) - d_o).max() <= max(1e-5, 3 * ulp) * max(1.0, dd.max() / 64.0)
    t, q = sk.from_root_dual_quat(d, par)
    assert np.abs(q - rot).max() <= 4e-6 * max(1.0, dd.max() / 32.0)


def _check_fk_at_a_magnitude(sk_, bone_scale, root_scale, per_frame_offsets):
    """the per-tile arithmetic of fk (fp32 walk / float64 rotations + fixed-point chain) over bone and root
    magnitudes from millimetres to thousands of kilometres, both sides of the decision thresholds, every walk shape:
    rotations <= 2e-6 whatever the positions do; positions within max(1e-5, 3 ulp of the largest coordinate) of what those
    rotation errors imply (|dR_parent t_j| summed down the chain)"""
    J, par, lead, rng = sk_
    rot = rng.standard_normal(lead + (J, 4)).astype(np.float32)
    gpos = (rng.uniform(-1, 1, lead + (3,)) * root_scale).astype(np.float32)
    off = (rng.uniform(-1, 1, (lead + (J, 3)) if per_frame_offsets else (J, 3)) * bone_scale).astype(np.float32)
    pos, rm = sk.fk(rot, gpos, off, par)
    p_o, r_o = co.fk(f64(rot), f64(gpos), f64(off), par)
    if pos.size:
        scale = np.abs(p_o).max()
        ulp3 = max(1e-5, 3 * 2.0 ** (np.floor(np.log2(scale)) - 23)) if scale > 0 else 1e-5
        d = np.zeros(J, int)
        for j in range(1, J):
            d[j] = d[par[j]] + 1
        depth = 1 + int(d.max())
        # rotations: fp32 products down the chain (129-deep chains accumulate 129 of them)
        assert np.abs(rm - r_o).max() <= 2e-6 * max(1.0, depth / 12.0)
        # positions: p_j = p_parent + R_parent t_j, so an error dR in the parent's rotation moves the joint by |dR t_j| whatever
        # the translation chain does.  What is asserted is that the chain adds no more than 3 ulp of the largest coordinate
        # (max(1e-5, .) at metre scale) on top of what the rotation errors of THIS result imply -- a statement that holds at
        # every ratio of bone length to root distance (a fixed "n ulp of the largest coordinate" does not: with small roots
        # the coordinates are all bones, and 4e-7 of rotation error is 3-4 ulp of them).
        e_rot = np.linalg.norm((rm - r_o).reshape(lead + (J, 9)), axis=-1)          # [..., J]
        t_len = np.linalg.norm(f64(off), axis=-1) * np.ones(lead + (J,))             # [..., J]
        implied = np.zeros(lead + (J,))
        for j in range(1, J):
            implied[..., j] = implied[..., par[j]] + e_rot[..., par[j]] * t_len[..., j]
        err = np.linalg.norm(pos - p_o, axis=-1)
        slack = err - implied
        assert slack.max() <= ulp3 * np.sqrt(3.0), (slack.max(), ulp3, scale, depth)
        np.testing.assert_array_equal(pos[..., 0, :].astype(np.float32), gpos)
