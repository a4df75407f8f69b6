"""BVH ingest and quat/dual_quat.unroll: host parser and oracle on the CPU, the GPU
path (from_euler -> unroll -> normalize, then fk) against vectors the reference produced from the
same committed synthetic22.bvh (oracle/make_golden.py: gen_bvh)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_close, golden
from oracle import c_oracle as co
from pymotion_amd.io.bvh import BVH

BVH_PATH = os.path.join(GOLDEN, "synthetic22.bvh")


def test_bvh_parser_matches_reference_load():
    g = golden("bvh.npz")
    want = g.get("load", "out64")
    b = BVH()
    b.load(BVH_PATH)
    d = b.data
    assert [n.decode() for n in g.get("load", "in")["names"]] == list(d["names"])
    np.testing.assert_array_equal(d["parents"], want["parents"])
    np.testing.assert_array_equal(d["end_sites_parents"], want["end_sites_parents"])
    np.testing.assert_array_equal(np.vectorize("xyz".index)(d["rot_order"]), want["rot_order"])
    for k in ("offsets", "end_sites", "positions", "rotations"):
        np.testing.assert_array_equal(d[k], want[k], err_msg=k)
    assert d["frame_time"] == float(want["frame_time"])
    assert d["positions"].shape == (48, 22, 3) and d["rotations"].shape == (48, 22, 3)


@pytest.mark.parametrize("case", ["unroll_smooth_ax0", "unroll_random_ax0", "unroll_ax1", "unroll_4d_ax-3"])
def test_oracle_unroll_matches_reference(case):
    g = golden("bvh.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")["out"]
    got = co.quat_unroll(i["q"].astype(np.float64), int(np.asarray(i["axis"]).reshape(-1)[0]))
    assert_close(got, want, 0.0, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["unroll_smooth_ax0", "unroll_random_ax0", "unroll_ax1", "unroll_4d_ax-3"])
def test_gpu_unroll_matches_reference(case):
    import torch

    import pymotion_amd.rotations.quat as quat
    import pymotion_amd.rotations.quat_torch as quat_t

    g = golden("bvh.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")["out"]
    ax = int(np.asarray(i["axis"]).reshape(-1)[0])
    q0 = i["q"].copy()
    got = quat.unroll(i["q"], ax)
    assert got.shape == i["q"].shape and got.dtype == np.float32
    assert_close(got, want, 1e-7, case)  # only signs change
    np.testing.assert_array_equal(i["q"], q0)  # the argument is not modified (the reference flips it in place)
    got_t = quat_t.unroll(torch.from_numpy(i["q"]).cuda(), ax)
    assert_close(got_t.cpu().numpy(), want, 1e-7, case + " torch")


@pytest.mark.gpu
def test_gpu_dq_unroll_and_long_sequences():
    import pymotion_amd.rotations.dual_quat as dq
    import pymotion_amd.rotations.quat as quat

    g = golden("bvh.npz")
    i, want = g.get("dq_unroll_ax0", "in"), g.get("dq_unroll_ax0", "out64")["out"]
    assert_close(dq.unroll(i["dq"], 0), want, 1e-7, "dq unroll")
    # many chunks (T > 1024), more series than one block (S > 64), ragged tail: against the sequential oracle
    rng = np.random.default_rng(3)
    # (S > 8192: more than one series block in the parity pass; 24 < S: several series blocks in the apply pass)
    for T, S in ((5000, 22), (1025, 70), (1, 3), (2, 1), (64, 64), (4097, 130), (300, 8200), (257, 25),
                 (3, 70_001), (2, 1_700_000)):  # short and very wide: 1-D grids (more than 65535 series blocks), thread-per-series scan
        base = np.cumsum(rng.normal(0, 0.08, (T, S, 4)), axis=0) + rng.normal(0, 1, (1, S, 4))
        q = (base * rng.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)
        got = quat.unroll(q, 0)
        ref = co.quat_unroll(q.astype(np.float64), 0)
        assert_close(got, ref, 1e-7, f"T={T} S={S}")
        d = np.sum(got[1:] * got[:-1], axis=-1)
        assert (d >= 0).all()  # the defining property: neighbours never sit on opposite covers
    # dual quaternions on long, wide clips: the sign of the real part decides, all 8 floats follow
    for T, S in ((1500, 22), (260, 30)):
        base = np.cumsum(rng.normal(0, 0.08, (T, S, 8)), axis=0) + rng.normal(0, 1, (1, S, 8))
        d8 = (base * rng.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)
        got = dq.unroll(d8, 0)
        sign = co.quat_unroll(d8[..., :4].astype(np.float64), 0)[..., :1] / d8[..., :1]
        assert_close(got, d8 * np.sign(sign), 1e-7, f"dq T={T} S={S}")


@pytest.mark.gpu
def test_gpu_bvh_get_data_and_fk_match_reference():
    import pymotion_amd.ops.skeleton as sk

    g = golden("bvh.npz")
    b = BVH()
    b.load(BVH_PATH)
    rots, pos, parents, offsets, end_sites, end_sites_parents = b.get_data()
    want = g.get("get_data", "out64")
    assert rots.shape == (48, 22, 4)
    assert_close(rots, want["rots"], 1e-6, "get_data rots")
    np.testing.assert_array_equal(pos, want["pos"])
    # README.md:73-78 of the reference: fk on the file's data
    p, r = sk.fk(rots, pos[:, 0, :], offsets, parents)
    wfk = g.get("fk", "out64")
    assert_close(p, wfk["pos"], 1e-5, "fk pos from BVH")
    assert_close(r, wfk["rotmats"], 1e-5, "fk rotmats from BVH")
    # set_data: quaternions back to the file's Euler channels (degrees); compare as rotations, modulo 360
    b.set_data(rots, pos)
    d = np.abs(b.data["rotations"] - g.get("set_data", "out64")["rotations"])
    d = np.minimum(d, 360.0 - d)
    assert d.max() < 0.05  # fp32 atan2 near gimbal configurations, in degrees


@pytest.mark.gpu
@pytest.mark.parametrize("T,J", [(1, 1), (2, 5), (63, 22), (1025, 22), (5000, 31), (70_000, 22), (3000, 64), (1500, 65), (40, 130)])
def test_gpu_fused_bvh_rotations_vs_oracle_chain(T, J):
    """pm_bvh_rotations_f32 = normalize(unroll(from_euler(radians(deg), order), axis=0)) (io/bvh.py:352-359) in one launch up to 64
    joints (one tile, several tiles with look-back, the 64-series limit), the three ops on the device beyond; both doors."""
    import torch

    from pymotion_amd import _backend, _ops

    rng = np.random.default_rng(T * 1000 + J)
    deg = np.cumsum(rng.normal(0, 6, (T, J, 3)), axis=0) + rng.uniform(-180, 180, (1, J, 3))  # drifting angles: many cover crossings
    orders = np.array([list(o) for o in ("zxy", "xyz", "yzx", "zyx", "xzy", "yxz", "zxz")])[rng.integers(0, 7, J)]
    def chain(d):
        w = co.quat_unroll(co.quat_from_euler(np.radians(d), np.tile(orders, (T, 1, 1))), 0)
        return w / (np.linalg.norm(w, axis=-1, keepdims=True) + 1e-8)

    # NumPy door: the file's float64 degrees, as the reference converts them (np.radians in float64, io/bvh.py:352).  Channels that wound
    # up past a turn are brought back to [-360, 360] exactly on the host before the fp32 cast (_ops.bvh_rotations), so the bar does not
    # grow with the winding: 2e-6 flat (round 4: 2e-6 + 1e-7 x max |angle| in radians, against the fp32-rounded degrees)
    got = _ops.bvh_rotations(_backend.numpy_backend(), deg, orders)
    assert got.shape == (T, J, 4) and got.dtype == np.float64
    assert_close(got, chain(deg), 2e-6, "fused get_data rotations")
    if T > 1:
        assert (np.sum(got[1:] * got[:-1], axis=-1) >= 0).all()
    # torch door: the caller's fp32 degrees ARE the input (angles of thousands of degrees at fp32: the kernel's float64 pi / 180 product
    # is rounded to fp32 once more, 1e-4 degrees = 2e-6 rad there)
    deg32 = deg.astype(np.float32)
    tol = 2e-6 + 1e-7 * np.abs(deg).max() * np.pi / 180
    got_t = _ops.bvh_rotations(_backend.torch_backend(), torch.from_numpy(deg32).cuda(), orders)
    assert got_t.dtype == torch.float32
    assert_close(got_t.cpu().numpy(), chain(deg32.astype(np.float64)), tol, "fused get_data rotations, torch door")


@pytest.mark.gpu
def test_gpu_fused_bvh_rotations_raw_abi():
    import ctypes as C

    import torch

    from pymotion_amd import _lib

    T, J = 100, 4
    deg = (torch.rand((T, J, 3), device="cuda") * 360 - 180).contiguous()
    out = torch.empty((T, J, 4), device="cuda")
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, 70)) + 8, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    order = np.tile(np.array([2, 0, 1], np.uint8), (70, 1))
    h = _lib.lib()
    assert h.pm_bvh_rotations_f32(p(deg), order.ctypes.data_as(C.c_void_p), T, J, p(out), p(ws), None) == _lib.PM_OK
    torch.cuda.synchronize()
    assert float((out.norm(dim=-1) - 1).abs().max()) < 1e-6
    assert h.pm_bvh_rotations_f32(p(deg), order.ctypes.data_as(C.c_void_p), T, 65, p(out), p(ws), None) == _lib.PM_EUNSUPPORTED
    bad = order.copy()
    bad[2, 1] = 3
    assert h.pm_bvh_rotations_f32(p(deg), bad.ctypes.data_as(C.c_void_p), T, J, p(out), p(ws), None) == _lib.PM_EINVAL
    assert h.pm_bvh_rotations_f32(p(deg), None, T, J, p(out), p(ws), None) == _lib.PM_EINVAL


@pytest.mark.gpu
def test_gpu_one_pass_scans_without_the_reset_launch():
    """pm_unroll_onepass_f32: the scans of quat.unroll / dual_quat.unroll / the BVH ingest on a PAIR of zeroed workspaces the caller
    alternates -- no reset launch.  A sequence of calls of different kinds and shapes (one tile, many tiles, batches of clips, a call that
    dirties nothing) gives the plain entry points' results bit for bit; after every call the block it used holds non-zero words only below
    the count it reported and the other block is zero again; the argument checks"""
    import ctypes as C

    import torch

    from pymotion_amd import _lib

    h = _lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    cap = 1 << 20
    pair = torch.zeros((2, cap // 8), dtype=torch.int64, device="cuda")
    dirty = [0, 0]
    cur = 0
    rng = np.random.default_rng(3)

    def walk(n, w):
        base = np.cumsum(rng.normal(0, 0.3, n + (w,)), axis=-3) + rng.normal(0, 1, n[:-2] + (1, n[-1], w))
        return (base * rng.choice([-1.0, 1.0], n + (1,))).astype(np.float32)

    calls = [(0, 1, 3000, 22), (1, 1, 50_000, 22), (0, 64, 300, 22), (2, 1, 1000, 22), (0, 1, 40, 5), (0, 1, 200_000, 22), (1, 7, 9000, 31), (2, 1, 70_000, 64),
             (0, 1, 3000, 22)]
    for kind, B, T, S in calls:
        w = 8 if kind == 1 else 4
        order = None
        if kind == 2:
            x = torch.from_numpy(rng.uniform(-180, 180, (T, S, 3)).astype(np.float32)).cuda()
            order_np = np.ascontiguousarray(rng.integers(0, 3, (S, 3)).astype(np.uint8))
            order_np[:, 1] = (order_np[:, 0] + 1) % 3
            order_np[:, 2] = (order_np[:, 0] + 2) % 3
            order = order_np.ctypes.data_as(C.c_void_p)
        else:
            x = torch.from_numpy(walk((B, T, S), w)).cuda()
        out = torch.empty((B, T, S, w if kind != 2 else 4), device="cuda")
        ref = torch.empty_like(out)
        ws = torch.empty(int(h.pm_quat_unroll_batched_workspace_bytes(B, T, S)), dtype=torch.uint8, device="cuda")
        if kind == 2:
            assert h.pm_bvh_rotations_f32(p(x), order, T, S, p(ref), p(ws), None) == _lib.PM_OK
        else:
            assert getattr(h, "pm_quat_unroll_batched_f32" if kind == 0 else "pm_dq_unroll_batched_f32")(p(x), B, T, S, p(ref), p(ws), None) == _lib.PM_OK
        n = C.c_int64(-1)
        rc = h.pm_unroll_onepass_f32(kind, p(x), order, B, T, S, p(out), p(pair[cur]), C.byref(n), p(pair[1 - cur]), dirty[1 - cur], None)
        assert rc == _lib.PM_OK, h.pm_last_error_string()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (kind, B, T, S)
        assert 0 <= n.value <= cap // 8
        assert not bool(pair[1 - cur].any()), "the other workspace was not zeroed"
        assert not bool(pair[cur][n.value:].any()), ("non-zero words beyond the reported count", kind, B, T, S, n.value, pair[cur].nonzero().flatten()[-4:].tolist())
        dirty[1 - cur], dirty[cur] = 0, n.value
        cur = 1 - cur
    assert any(d > 0 for d in dirty)
    # an empty scan still leaves the other workspace clean (the caller swaps the two on PM_OK)
    x0 = torch.zeros((1, 0, 22, 4), device="cuda")
    n = C.c_int64(-1)
    assert h.pm_unroll_onepass_f32(0, p(x0), None, 1, 0, 22, p(x0), p(pair[cur]), C.byref(n), p(pair[1 - cur]), dirty[1 - cur], None) == _lib.PM_OK
    torch.cuda.synchronize()
    assert n.value == 0 and not bool(pair[1 - cur].any())
    x = torch.zeros((1, 10, 65, 4), device="cuda")
    n = C.c_int64(-1)
    assert h.pm_unroll_onepass_f32(0, p(x), None, 1, 10, 65, p(x), p(pair[0]), C.byref(n), p(pair[1]), 0, None) == _lib.PM_EUNSUPPORTED
    assert h.pm_unroll_onepass_f32(0, p(x), None, 1, 10, 22, p(x), p(pair[0]), C.byref(n), p(pair[0]), 4, None) == _lib.PM_EINVAL
    assert h.pm_unroll_onepass_f32(3, p(x), None, 1, 10, 22, p(x), p(pair[0]), C.byref(n), p(pair[1]), 0, None) == _lib.PM_EINVAL
    assert h.pm_unroll_onepass_f32(0, p(x), None, 1, 10, 22, p(x), p(pair[0]), None, p(pair[1]), 0, None) == _lib.PM_EINVAL


@pytest.mark.gpu
def test_gpu_unroll_doors_alternate_their_workspace_pair():
    """both doors keep a pair per thread, device and stream: repeated calls of changing shapes agree with the oracle (a pair out of step would
    make the scan wait for words nobody writes -- the suite runs under a timeout), and a second stream gets a pair of its own"""
    import torch

    import pymotion_amd.rotations.quat as quat
    import pymotion_amd.rotations.quat_torch as quat_t

    rng = np.random.default_rng(9)
    side = torch.cuda.Stream()
    for T, S in ((5000, 22), (90_000, 22), (300, 4), (70_000, 40), (5000, 22), (100_000, 3)):
        base = np.cumsum(rng.normal(0, 0.3, (T, S, 4)), axis=0) + rng.normal(0, 1, (1, S, 4))
        q = (base * rng.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)
        want = co.quat_unroll(q.astype(np.float64), 0).astype(np.float32)
        np.testing.assert_array_equal(quat.unroll(q, 0).astype(np.float32), want)
        qt = torch.from_numpy(q).cuda()
        np.testing.assert_array_equal(quat_t.unroll(qt, 0).cpu().numpy(), want)
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            got = quat_t.unroll(qt, 0)
        side.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_config1_bvh_1000_frames_gpu_vs_numpy_cpu_reference(tmp_path):
    """BASELINE.json configs[0]: NumPy fk on a 22-joint BVH, 1000 frames (CPU) -- the same arrays through the
    GPU path must agree to 1e-5.  The file is generated on the fly (seeded), read by the build's own loader."""
    import pymotion_amd.ops.skeleton as sk
    from oracle import numpy_ref as nr
    from pymotion_amd import synthetic as syn

    path = str(tmp_path / "clip1000.bvh")
    syn.write_synthetic_bvh(path, n_frames=1000, seed=123)
    b = BVH()
    b.load(path)
    rots, pos, parents, offsets, _, _ = b.get_data()
    assert rots.shape == (1000, 22, 4) and rots.dtype == np.float64
    assert np.abs(np.linalg.norm(rots, axis=-1) - 1).max() < 1e-6
    assert (np.sum(rots[1:] * rots[:-1], axis=-1) >= 0).all()  # unrolled along frames
    p_gpu, r_gpu = sk.fk(rots, pos[:, 0, :], offsets, parents)
    p_cpu, r_cpu = nr.fk(rots, pos[:, 0, :], offsets, parents)  # the reference algorithm, float64
    assert_close(p_gpu, np.ascontiguousarray(p_cpu), 1e-5, "config 1 positions")
    assert_close(r_gpu, np.ascontiguousarray(r_cpu), 1e-5, "config 1 rotation matrices")


# ---- resets: a neighbour dot product of exactly 0 / NaN restarts the accumulated sign (tests/golden/degenerate.npz) ----

def test_oracle_unroll_with_resets_matches_reference():
    g = golden("degenerate.npz")
    i, want = g.get("unroll_resets", "in"), g.get("unroll_resets", "out64")["out"]
    got = co.quat_unroll(i["q"].astype(np.float64), 0)
    assert (np.isnan(got) == np.isnan(want)).all()
    np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))


@pytest.mark.gpu
def test_gpu_unroll_with_resets_matches_reference():
    """zero-padded rows, exactly orthogonal steps and NaN rows, inside / at the edge of / across the 256-frame chunks and
    64-frame sub-tiles of the scan: the kernel must forget the sign accumulated before them, like the reference's loop"""
    import torch

    import pymotion_amd.rotations.dual_quat as dq
    import pymotion_amd.rotations.quat as quat
    import pymotion_amd.rotations.quat_torch as quat_t

    g = golden("degenerate.npz")
    i, want = g.get("unroll_resets", "in"), g.get("unroll_resets", "out64")["out"]
    for got in (quat.unroll(i["q"], 0), quat_t.unroll(torch.from_numpy(i["q"]).cuda(), 0).cpu().numpy()):
        assert (np.isnan(got) == np.isnan(want)).all()
        np.testing.assert_array_equal(np.nan_to_num(got).astype(np.float32), np.nan_to_num(want).astype(np.float32))
    # the same series as columns of a wide clip (thread-per-series chunk scan) and with the unroll axis last-but-one
    wide = np.tile(i["q"], (1, 40, 1))
    got = quat.unroll(wide, 0)
    np.testing.assert_array_equal(np.nan_to_num(got).astype(np.float32), np.nan_to_num(np.tile(want, (1, 40, 1))).astype(np.float32))
    got = quat.unroll(np.ascontiguousarray(np.moveaxis(i["q"], 0, 1)), 1)
    np.testing.assert_array_equal(np.nan_to_num(got).astype(np.float32), np.nan_to_num(np.moveaxis(want, 0, 1)).astype(np.float32))
    # a long clip: the ballot scan over chunks (> 256 chunks), resets sprinkled over it
    T = 256 * 300 + 17
    rng = np.random.default_rng(8)
    q = rng.standard_normal((T, 3, 4)).astype(np.float32)
    q[rng.random((T, 3)) < 0.001] = 0
    q[255::256, 1] = 0
    got = quat.unroll(q, 0)
    ref = co.quat_unroll(q.astype(np.float64), 0)
    np.testing.assert_array_equal(got.astype(np.float32), ref.astype(np.float32))
    i8, want8 = g.get("dq_unroll_resets", "in"), g.get("dq_unroll_resets", "out64")["out"]
    got8 = dq.unroll(i8["dq"], 0)
    assert (np.isnan(got8) == np.isnan(want8)).all()
    np.testing.assert_array_equal(np.nan_to_num(got8).astype(np.float32), np.nan_to_num(want8).astype(np.float32))


# ---- the one-pass (look-back) form for clips of at most 64 series, and the three-pass form behind it ----------------------

def _clip_with_flips_and_resets(T, S, W, seed):
    rng = np.random.default_rng(seed)
    base = np.cumsum(rng.normal(0, 0.08, (T, S, W)), axis=0) + rng.normal(0, 1, (1, S, W))
    q = (base * rng.choice([-1.0, 1.0], (T, S, 1))).astype(np.float32)
    if T > 40:  # resets: zero rows, at tile / word edges too
        for t in (1, 31, 32, 33, T // 2, T - 2):
            q[t, rng.integers(0, S)] = 0.0
    return q


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"PM_UNROLL_R": "16"}, {"PM_UNROLL_R": "32"}, {"PM_UNROLL_R": "4"}, {"PM_UNROLL_ONEPASS": "0"}])
def test_gpu_unroll_one_pass_tiles_and_three_pass_agree_with_the_oracle(env, monkeypatch):
    import pymotion_amd.rotations.dual_quat as dq
    import pymotion_amd.rotations.quat as quat
    from pymotion_amd import _lib

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with _lib.variant("tuning" if env else "prod"):
        for T, S in ((5000, 22), (30_000, 1), (3000, 64), (70_001, 3), (9000, 63), (47, 22), (1, 5), (2, 64)):
            q = _clip_with_flips_and_resets(T, S, 4, seed=T + S)
            got = quat.unroll(q, 0)
            ref = co.quat_unroll(q.astype(np.float64), 0)
            np.testing.assert_array_equal(got, ref.astype(np.float32), err_msg=f"T={T} S={S} {env}")
        for T, S in ((6000, 22), (2500, 64), (3, 2)):
            d8 = _clip_with_flips_and_resets(T, S, 8, seed=T)
            got = dq.unroll(d8, 0)
            sref = co.quat_unroll(d8[..., :4].astype(np.float64), 0)
            flipped = np.signbit(sref[..., 0]) != np.signbit(d8[..., 0])
            flipped |= (sref[..., 1] != d8[..., 1]) & (d8[..., 0] == 0)
            want = np.where(flipped[..., None], -d8, d8)
            np.testing.assert_array_equal(np.abs(got), np.abs(d8))
            nz = (d8[..., :4] != 0).any(axis=-1)
            np.testing.assert_array_equal(got[nz], want[nz], err_msg=f"dq T={T} S={S} {env}")


@pytest.mark.gpu
def test_gpu_unroll_full_size_property():
    """2^20 frames x 22 series through the one-pass kernel (5632 chained tiles): neighbours never end on opposite covers,
    nothing but signs changes, and the first frame keeps its sign"""
    import torch

    import pymotion_amd.rotations.quat_torch as quat_t

    g = torch.Generator(device="cuda").manual_seed(5)
    T, S = 1 << 20, 22
    q = torch.randn((T, S, 4), device="cuda", generator=g)
    out = quat_t.unroll(q, 0)
    assert torch.equal(out.abs(), q.abs())
    assert torch.equal(out[0], q[0])
    assert bool(((out[1:] * out[:-1]).sum(-1) >= 0).all())
    # twice: the workspace (ticket, statuses) is reset by every call
    assert torch.equal(quat_t.unroll(q, 0), out)


@pytest.mark.gpu
def test_gpu_unroll_batches_of_clips_without_transposing():
    """[B, T, J, 4] unrolled along T: B independent look-back scans in one launch (short clips: one tile each; long clips:
    chained tiles per clip), torch and NumPy doors, quaternions and dual quaternions, against the oracle along that axis"""
    import ctypes as C

    import torch

    import pymotion_amd.rotations.dual_quat as dq
    import pymotion_amd.rotations.quat as quat
    import pymotion_amd.rotations.quat_torch as quat_t
    from pymotion_amd import _lib

    rng = np.random.default_rng(12)
    for B, T, S in ((1000, 60, 22), (3, 20_000, 22), (7, 513, 64), (2, 1, 5), (40, 300, 1), (5, 4097, 31)):
        base = np.cumsum(rng.normal(0, 0.08, (B, T, S, 4)), axis=1) + rng.normal(0, 1, (B, 1, S, 4))
        q = (base * rng.choice([-1.0, 1.0], (B, T, S, 1))).astype(np.float32)
        if T > 40:
            q[B // 2, T // 2, 0] = 0.0  # a reset inside one clip only
        ref = co.quat_unroll(q.astype(np.float64), 1).astype(np.float32)
        np.testing.assert_array_equal(quat.unroll(q, 1), ref, err_msg=f"B={B} T={T} S={S}")
        np.testing.assert_array_equal(quat_t.unroll(torch.from_numpy(q).cuda(), 1).cpu().numpy(), ref)
        np.testing.assert_array_equal(quat.unroll(q, -3), ref)
    # 5-D: two batch axes in front, two series axes behind
    q5 = rng.standard_normal((3, 4, 200, 2, 11, 4)).astype(np.float32)
    np.testing.assert_array_equal(quat.unroll(q5, 2), co.quat_unroll(q5.astype(np.float64), 2).astype(np.float32))
    # dual quaternions: the real part decides
    d8 = rng.standard_normal((6, 700, 22, 8)).astype(np.float32)
    got = dq.unroll(d8, 1)
    sref = co.quat_unroll(d8[..., :4].astype(np.float64), 1)
    flipped = np.signbit(sref[..., 0]) != np.signbit(d8[..., 0])
    np.testing.assert_array_equal(got, np.where(flipped[..., None], -d8, d8))
    # the C ABI with wide clips in a batch (S > 64: the clips run one after the other through the three-pass scan)
    B, T, S = 3, 700, 70
    qw = torch.randn((B, T, S, 4), device="cuda")
    out = torch.empty_like(qw)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_batched_workspace_bytes(B, T, S)), dtype=torch.uint8, device="cuda")
    _lib.call("pm_quat_unroll_batched_f32", C.c_void_p(qw.data_ptr()), B, T, S, C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), co.quat_unroll(qw.cpu().numpy().astype(np.float64), 1).astype(np.float32))


@pytest.mark.gpu
def test_gpu_unroll_many_tiny_clips_take_the_wide_form():
    """a million two-frame clips: not a workgroup per clip -- the front-end moves the axis and the series-parallel scan runs"""
    import pymotion_amd.rotations.quat as quat

    rng = np.random.default_rng(4)
    q = rng.standard_normal((200_000, 2, 3, 4)).astype(np.float32)
    got = quat.unroll(q, 1)
    flip = np.sum(q[:, 1] * q[:, 0], axis=-1) < 0
    want = q.copy()
    want[:, 1][flip] *= -1
    np.testing.assert_array_equal(got, want)
