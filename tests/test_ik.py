"""quat.from_to / from_to_axis and skeleton.from_root_positions: oracle on the
CPU, the HIP kernels on the GPU, both against vectors produced by the reference (make_golden.py: gen_ik)."""
import numpy as np
import pytest

from conftest import assert_close, golden, up64
from oracle import c_oracle as co

FROM_TO = ["from_to", "from_to_nonorm", "from_to_axis", "from_to_single"]
FRP = ["from_root_positions_J22", "from_root_positions_J52", "from_root_positions_topoJ9", "from_root_positions_starJ6"]


def _call_from_to(mod, case, i):
    if case == "from_to_axis":
        return mod.from_to_axis(i["v1"], i["v2"], i["axis"])
    if case == "from_to_nonorm":
        return mod.from_to(i["v1"], i["v2"], False)
    return mod.from_to(i["v1"], i["v2"])


def _same_rotation_err(a, b):
    """max over elements of min(|a-b|, |a+b|): q and -q are the same rotation (from_matrix branch picks)."""
    return np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max()


@pytest.mark.parametrize("case", FROM_TO)
def test_oracle_from_to(case):
    g = golden("ik.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")["out"]

    class M:
        from_to = staticmethod(lambda a, b, n=True: co.quat_from_to(a, b, n))
        from_to_axis = staticmethod(lambda a, b, c: co.quat_from_to_axis(a, b, c))

    got = _call_from_to(M, case, i)
    assert_close(np.asarray(got).reshape(want.shape), want, 1e-12, case)


@pytest.mark.parametrize("case", FRP)
def test_oracle_from_root_positions(case):
    g = golden("ik.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")["rot"]
    got = co.from_root_positions(i["pos"], i["parents"], i["off"])
    assert_close(got, want, 1e-9, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FROM_TO)
def test_gpu_from_to(case):
    import torch

    import pymotion_amd.rotations.quat as quat
    import pymotion_amd.rotations.quat_torch as quat_t

    g = golden("ik.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")["out"]
    got = _call_from_to(quat, case, i)
    assert got.shape == want.shape
    # elements sitting ON an isclose() threshold (the constructed parallel / anti-parallel rows) may fall on
    # either side in fp32; there both answers are rotations by ~0 or ~pi about the reference's axis
    assert_close(got, want, 2e-5, case)
    it = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in i.items()}
    got_t = _call_from_to(quat_t, case, it)
    assert_close(got_t.cpu().numpy(), want, 2e-5, case + " torch")


@pytest.mark.gpu
@pytest.mark.parametrize("case", FRP)
def test_gpu_from_root_positions(case):
    import torch

    import pymotion_amd.ops.skeleton as sk
    import pymotion_amd.ops.skeleton_torch as skt

    g = golden("ik.npz")
    i, want = g.get(case, "in"), g.get(case, "out64")["rot"]
    got = sk.from_root_positions(i["pos"], i["parents"], i["off"])
    assert got.shape == want.shape and got.dtype == np.float64
    # the reference's own twin test uses 1e-2 here (test_skeleton.py:225); measured 3.6e-6 / 1.9e-6 / 1.4e-6 / 1.6e-7 (round 2
    # asserted 2e-4: its alignment lost digits near (anti-)parallel directions and ignored the reference's "+ 1e-8"s, ik.hip)
    assert _same_rotation_err(got, want) <= 2e-5, _same_rotation_err(got, want)
    # the recovered pose is the reference's pose (the algorithm itself only matches the input positions
    # approximately: roll is fixed from one extra child at a time; the reference tests it at 1e-2)
    pos, _ = sk.fk(got, np.zeros((got.shape[0], 3)), i["off"], i["parents"])
    pos_ref, _ = co.fk(want, np.zeros((got.shape[0], 3)), i["off"].astype(np.float64), i["parents"])
    assert np.abs(pos - pos_ref).max() <= 1e-5
    got_t = skt.from_root_positions(torch.from_numpy(i["pos"]).cuda(), torch.from_numpy(i["parents"]), torch.from_numpy(i["off"]).cuda())
    assert _same_rotation_err(got_t.cpu().numpy(), want) <= 2e-5


def _reference_sensitivity(pos, par, off, ref, draws=3, ulps=1, keep=None):
    """How far the REFERENCE's own (float64) answer moves when its fp32 inputs move by `ulps` ulps: from_to is ill-conditioned where a
    bone has to turn by nearly 180 degrees (the axis of a half turn is any direction perpendicular to the bone), and everything
    below such a joint inherits the twist.  Max over a few random perturbations, per (frame, joint)."""
    s = np.zeros(ref.shape[:2])
    for k in range(draws):
        up = np.random.default_rng(k + 1).random(pos.shape) < 0.5
        pos2 = pos
        for _ in range(ulps):
            pos2 = np.nextafter(pos2, np.where(up, np.inf, -np.inf).astype(np.float32))
        ref2 = co.from_root_positions(pos2.astype(np.float64), par, off.astype(np.float64))
        if keep is not None:
            keep.append(ref2)  # the reference's answers to the perturbed inputs themselves (what a record past a switch is held against)
        s = np.maximum(s, np.minimum(np.abs(ref2 - ref).max(-1), np.abs(ref2 + ref).max(-1)))
    return s


def _record_bar(pos, par, off, ref, k=8, draws=3, keep=None):
    """The bar of one (frame, joint) record: 2e-5, plus what `k` ulps of the fp32 inputs do to the reference's own float64 answer --
    estimated linearly (k x the movement under one ulp) AND directly (the movement under k ulps).  The second catches what the first
    cannot: the reference's answer is DISCONTINUOUS in its inputs -- np.sign(cross . axis) decides which way a roll turns
    (quat.py:628), np.isclose(dot, +-1) picks the parallel / anti-parallel branches (:551-571, :635-645) -- and a record that sits within
    k ulps of such a switch has no well-defined answer at fp32 input precision: either side is the reference's answer to inputs a few
    ulps away.  (Round 4's win2_511 -- 511 joints, depth 334, mostly one-child chains -- read 0.67 on ONE record of 204 400: joint 309 of
    frame 351 sits below an alignment of 177.8 degrees that nothing re-anchors, where the reference itself moves by 1.4e-4 per ulp, and its
    roll's cross . axis is within that of zero: tools/ik_path_diag.py, profiles/r05_ik_deep_tables.txt.  Every other joint of that path is
    within 2 x the reference's own movement.)"""
    lin = _reference_sensitivity(pos, par, off, ref, draws=draws, ulps=1)
    direct = _reference_sensitivity(pos, par, off, ref, draws=draws, ulps=k, keep=keep)
    return 2e-5 + np.maximum(k * lin, direct), lin


@pytest.mark.gpu
@pytest.mark.parametrize("J,F", [(22, 4099), (52, 3001)])
def test_gpu_from_root_positions_large_vs_oracle(J, F):
    """Random poses at test size.  The bar per record: 2e-5, plus -- on the handful of records where the reference's answer is
    itself decided by the last bit of its fp32 inputs -- a small multiple of how far one ulp of input moves THE REFERENCE
    (measured at 4099 x 22: 6 of 90 178 records above 2e-5, the worst 3.8e-5, all within 1.5x the reference's own movement;
    round 2 asserted 2e-4 on everything)."""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import synthetic as syn

    par = syn.PARENTS_22 if J == 22 else syn.PARENTS_52
    rot, root, off, par = syn.fk_workload(F, parents=par, seed=9, normalized=True, offset_scale=0.3 if J == 22 else 0.15)
    pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
    pos = pos.astype(np.float32)
    got = sk.from_root_positions(pos, par, off)
    ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
    err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
    sens = _reference_sensitivity(pos, par, off, ref)
    over = err > 2e-5
    assert over.mean() <= 5e-4, int(over.sum())
    assert (err <= 2e-5 + 8.0 * sens).all(), (float(err.max()), float(((err - 2e-5) / np.maximum(sens, 1e-12))[over].max()))
    assert np.quantile(err, 0.999) <= 2e-5
    p2, _ = sk.fk(got, np.zeros_like(root), off, par)
    p_ref, _ = co.fk(ref, np.zeros((F, 3)), off.astype(np.float64), par)
    assert np.abs(p2 - p_ref).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("J,F,four", [(57, 700, None), (72, 700, None), (96, 520, True), (130, 300, True), (250, 130, None), (300, 70, False)])
def test_gpu_from_root_positions_long_bushy_skeletons_on_four_chains(J, F, four):
    """beyond 56 joints a tree wide enough to shorten the walk again is scheduled onto FOUR chains per frame (16 frames per wave)"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    par = syn.random_parents(J, np.random.default_rng(J))
    rot, root, off, par = syn.fk_workload(F, parents=par, seed=J, normalized=True, offset_scale=0.1)
    pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
    pos = pos.astype(np.float32)
    got = sk.from_root_positions(pos, par, off)
    name = _lib.last_kernel_name()
    assert four is None or name.endswith(", 4>(pm::IkArgs, int)") == four, name  # (None: whichever the schedules of this tree favour)
    ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
    err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
    # the tile kernel's bar, per record: 2e-5, plus -- where an alignment sits within a hair of a half turn, whose twist every joint
    # below inherits -- a small multiple of how far one ulp of the inputs moves the reference's own answer (_reference_sensitivity)
    sens = _reference_sensitivity(pos, par, off, ref)
    assert (err <= 2e-5 + 8.0 * sens).all(), (float(err.max()), float(((err - 2e-5) / np.maximum(sens, 1e-12)).max()))
    assert np.quantile(err, 0.999) <= 2e-5, float(np.quantile(err, 0.999))
    p2, _ = sk.fk(got, np.zeros_like(root), off, par)
    p_ref, _ = co.fk(ref, np.zeros((F, 3)), off.astype(np.float64), par)
    assert np.abs(p2 - p_ref).max() <= 2e-5


def _dfs_humanoid(J, fingers=5):
    """a humanoid stored DEPTH FIRST (the order of a BVH file): hips -> spine (4) -> [neck, head | left arm (4) -> `fingers` x 3 |
    right arm likewise] | left leg (4) | right leg (4); joints left over lengthen the tail of the last finger"""
    par = [0]

    def chain(p, n):
        for _ in range(n):
            par.append(p)
            p = len(par) - 1
        return p

    chest = chain(0, 4)
    chain(chest, 2)
    for _ in range(2):
        wrist = chain(chest, 4)
        for _ in range(fingers):
            chain(wrist, 3)
    chain(0, 4)
    chain(0, 4)
    par = par[:J]
    while len(par) < J:
        par.append(len(par) - 1)
    return np.asarray(par, dtype=np.int32)


def _chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    p[J // 2] = 0
    p[3 * J // 4] = J // 4
    return p


@pytest.mark.usefixtures("lane_per_frame_at_test_sizes")
@pytest.mark.gpu
@pytest.mark.parametrize("J,kind,lane", [(24, "chain", True), (31, "chain", True), (23, "chain", False), (53, "body5", True), (47, "body4", True), (64, "chain", True),
                                         (65, "body5", True), (128, "chain", True), (250, "body5", True), (512, "chain", True), (59, "body6", None),
                                         (52, "smplh", True), (16, "chain", True), (12, "body4", True), (22, "body4", False), (29, "chain", False)])
def test_gpu_from_root_positions_lane_per_frame_on_depth_first_skeletons(J, kind, lane):
    """skeletons stored depth first (every BVH hierarchy) take from_root_positions_order_kernel from 30 joints on and at multiples of four from
    12: one lane per frame, joints streamed through a ring of LDS slots, the further children of a joint read from the ring's window or
    prefetched per lane (at most 16), full and partial tiles of 64 frames; the others stay on the tile kernels"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    par = {"chain": lambda: _chain_like(J), "smplh": lambda: syn.PARENTS_52}.get(kind, lambda: _dfs_humanoid(J, int(kind[-1])))()
    for F in (1, 63, 64, 65, 400):
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=J + F, normalized=True, offset_scale=0.1)
        pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
        pos = pos.astype(np.float32)
        got = sk.from_root_positions(pos, par, off)
        assert lane is None or ("from_root_positions_order_kernel" in _lib.last_kernel_name()) == lane, _lib.last_kernel_name()
        ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
        err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
        # the tile kernel's bar, per record: 2e-5 + 8 x how far one ulp of the inputs moves the reference's own answer (an alignment
        # within a hair of a half turn is decided by the last bit of its input, and every joint below inherits the twist)
        # (beyond 128 joints a twist anywhere is inherited -- and amplified, by tan(turn / 2) per joint -- by hundreds of joints below it,
        # in the reference as here: the kernel's fp32 steps weigh like a few ulps of input each, hence the wider multiple there;
        # measured 35x at 63 x 512)
        sens = _reference_sensitivity(pos, par, off, ref)
        k = 8.0 if J <= 128 else 64.0
        assert (err <= 2e-5 + k * sens).all(), (F, float(err.max()), float(((err - 2e-5) / np.maximum(sens, 1e-12)).max()))
        assert np.median(err) <= (1e-6 if J <= 128 else 1e-5), (F, float(np.median(err)))
        leaves = np.setdiff1d(np.arange(J), par[1:])
        assert (got[:, leaves] == np.array([1, 0, 0, 0], np.float32)).all()   # joints without children keep the exact identity
        p2, _ = sk.fk(got, np.zeros_like(root), off, par)
        p_ref, _ = co.fk(ref, np.zeros((F, 3)), off.astype(np.float64), par)
        assert np.abs(p2 - p_ref).max() <= (2e-5 if J <= 128 else 1e-4), float(np.abs(p2 - p_ref).max())


def _level_order(par):
    """the same tree stored breadth first (the order of the SMPL family's tables): children of a joint in their old order"""
    J = len(par)
    kids = [[] for _ in range(J)]
    for j in range(1, J):
        kids[par[j]].append(j)
    order, q = [], [0]
    while q:
        j = q.pop(0)
        order.append(j)
        q.extend(kids[j])
    new = {o: i for i, o in enumerate(order)}
    p2 = np.zeros(J, np.int32)
    for o in range(1, J):
        p2[new[o]] = new[par[o]]
    return p2


def _windowed_tree(J, w, rng):
    """parents first, every parent at most `w` joints before its child: neither depth first nor breadth first"""
    p = np.zeros(J, np.int32)
    for j in range(1, J):
        p[j] = rng.integers(max(0, j - w), j)
    return p


@pytest.mark.usefixtures("lane_per_frame_at_test_sizes")
@pytest.mark.gpu
@pytest.mark.parametrize("kind,order", [("smplh", True), ("smpl24", True), ("smplx55", None), ("bfs_body4_47", None), ("bfs_body5_53", None), ("win3_40", None),
                                        ("win4_64", None), ("win2_128", None), ("win6_96", None), ("win3_33", None), ("bfs_chain_64", None), ("bfs_chain_128", None),
                                        ("win2_511", None), ("win3_300", None), ("bfs_body5_253", None)])
def test_gpu_from_root_positions_lane_per_frame_on_tables_in_any_order(kind, order):
    """tables that are parents-first but NOT depth first (SMPL-H's level-order 52 joints) take from_root_positions_order_kernel when their
    children sit inside the ring's window or the per-lane queue (ik_order_plan), the tile kernels otherwise: same bars either way, full and
    partial tiles of 64 frames"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    rng = np.random.default_rng(len(kind))
    if kind == "smplh":
        par = syn.PARENTS_52
    elif kind == "smpl24":
        par = np.concatenate([syn.PARENTS_52[:22], [20, 21]]).astype(np.int32)
    elif kind == "smplx55":  # body, jaw and eyes under the head, two hands: level order
        hand = lambda w, b: [w if k % 3 == 0 else b + k - 1 for k in range(15)]  # noqa: E731
        par = np.asarray(list(syn.PARENTS_52[:22]) + [15, 15, 15] + hand(20, 25) + hand(21, 40), np.int32)
    elif kind.startswith("bfs_body"):
        par = _level_order(_dfs_humanoid(int(kind.split("_")[2]), int(kind[8])))
    elif kind.startswith("bfs_chain"):
        par = _level_order(_chain_like(int(kind.split("_")[2])))
    else:
        par = _windowed_tree(int(kind.split("_")[1]), int(kind[3]), rng)
    J = len(par)
    assert (par[1:] < np.arange(1, J)).all()
    dep = np.zeros(J, int)
    for j in range(1, J):
        dep[j] = dep[par[j]] + 1
    depth = int(dep.max())
    took = set()
    for F in (1, 63, 64, 65, 400):
        rot, root, off, par = syn.fk_workload(F, parents=par, seed=J + F, normalized=True, offset_scale=0.1)
        pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
        pos = pos.astype(np.float32)
        got = sk.from_root_positions(pos, par, off)
        name = _lib.last_kernel_name()
        took.add("order" if "from_root_positions_order_kernel" in name else "tile")
        assert order is None or ("from_root_positions_order_kernel" in name) == order, name
        ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
        err = np.minimum(np.abs(got - ref).max(-1), np.abs(got + ref).max(-1))
        # (these deep, narrow trees have many nearly straight bones: the sensitivity is the max over twelve draws -- with three the worst
        # record of win6_96 read 13.8 x, with twelve 1.5 x, on this kernel and on the tile kernel alike.  Beyond 128 joints -- win2_511:
        # depth 334 -- a twist is inherited by hundreds of joints below it, in the reference as here, and the kernel's fp32 steps weigh
        # like a few ulps of input each: 64 ulps there; and the bar counts the reference's discontinuities, see _record_bar)
        k = 8 if J <= 128 else 64
        sides = []
        bar, sens = _record_bar(pos, par, off, ref, k=k, draws=12, keep=sides)
        assert (err <= bar).all(), (F, name, float(err.max()), float(((err - 2e-5) / np.maximum(sens, 1e-12)).max()), int((err > bar).sum()))
        flipped = err > 2e-5 + k * sens   # records held by the direct k-ulp measure only: a switch of the reference within k ulps
        assert flipped.sum() <= max(1, 1e-4 * flipped.size), (F, name, int(flipped.sum()))
        if flipped.any():
            #  "anything within the distance the reference jumps" is not the statement: a record past a switch must be the
            # reference's answer on THE OTHER SIDE -- its float64 answer to one of the inputs k ulps away -- to the smooth part of the bar
            # (asked of the records AT a switch -- flipped, parent not flipped: what hangs below one inherits its parent's other pose on top
            # of its own movement and is held by the bar above)
            other = np.min([np.minimum(np.abs(got - r2).max(-1), np.abs(got + r2).max(-1)) for r2 in sides], axis=0)
            pf = flipped[:, par]
            pf[:, 0] = False
            top = flipped & ~pf
            assert (other[top] <= 2e-5 + 2 * k * sens[top]).all(), (F, name, float(other[top].max()), float(sens[top].max()))
        assert np.median(err) <= (1e-6 if J <= 128 else 1e-5), (F, float(np.median(err)))
        leaves = np.setdiff1d(np.arange(J), par[1:])
        assert (got[:, leaves] == np.array([1, 0, 0, 0], np.float32)).all()
        p2, _ = sk.fk(got, np.zeros_like(root), off, par)
        p_ref, _ = co.fk(ref, np.zeros((F, 3)), off.astype(np.float64), par)
        # (positions through fk: the rotations' errors add up along a chain -- beyond ~64 joints deep the bar of the long skeletons above;
        # frames with a record on the other side of one of the reference's switches are a different -- equally valid -- pose below it)
        ok = ~flipped.any(axis=1)
        assert not ok.any() or np.abs(p2 - p_ref)[ok].max() <= (2e-5 if depth <= 64 else (1e-4 if depth <= 128 else 1e-3)), float(np.abs(p2 - p_ref)[ok].max())
    print(kind, J, depth, sorted(took))


@pytest.mark.gpu
def test_gpu_mirror_positions_vs_reference_golden():
    import pymotion_amd.ops.skeleton as sk

    g = golden("ik.npz")
    i, want = g.get("mirror_positions_X", "in"), g.get("mirror_positions_X", "out64")
    r, gt, o, _ = sk.mirror(i["rot"], i["root"], i["parents"], i["off"], None, None, "positions", "X")
    assert _same_rotation_err(r, want["rot"]) <= 2e-5   # true mirror -> fk -> from_root_positions (measured 3.5e-6; round 2: 5e-4)
    assert_close(gt, want["gt"], 1e-7, "mirrored translation")
    assert_close(o, want["off"], 1e-7, "offsets unchanged in mode 'positions'")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,J", [("body", 22), ("smplh", 52), ("random", 40), ("random", 64), ("random", 96), ("random", 128)])
@pytest.mark.parametrize("chains", ["2", "4"])
def test_gpu_from_root_positions_both_list_schedules_give_the_same_bits(kind, J, chains, monkeypatch):
    """round 6: the tile kernels read a parent's world quaternion at the top of the step, so a joint may follow its parent in the very next step on
    any chain, and the host keeps the cheaper of two list schedules (ik.hip: ik_schedule).  A schedule only decides WHEN a joint is aligned --
    every alignment sees the same finished parent: the two rules (PM_IK_RELAXED = 0 / 1 on the tuning build), the production library's pick and the
    oracle agree, the first three bit for bit"""
    import pymotion_amd.ops.skeleton as sk
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn

    par = {"body": syn.PARENTS_22, "smplh": syn.PARENTS_52}.get(kind)
    if par is None:
        par = syn.random_parents(J, np.random.default_rng(J))
    F = 16 * 9 + 5
    rot, root, off, par = syn.fk_workload(F, parents=par, seed=3 * J, normalized=True, offset_scale=0.1)
    pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
    pos = pos.astype(np.float32)
    got = {}
    monkeypatch.setenv("PM_IK_ORDER", "0")
    monkeypatch.setenv("PM_IK_CHAINS", chains)
    with _lib.variant("tuning"):
        for rule in ("0", "1"):
            monkeypatch.setenv("PM_IK_RELAXED", rule)
            got[rule] = sk.from_root_positions(pos, par, off)
            assert "from_root_positions_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
        monkeypatch.delenv("PM_IK_RELAXED")
        got["both"] = sk.from_root_positions(pos, par, off)
    np.testing.assert_array_equal(got["0"], got["1"])
    np.testing.assert_array_equal(got["0"], got["both"])
    ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
    err = np.minimum(np.abs(got["both"] - ref).max(-1), np.abs(got["both"] + ref).max(-1))
    assert np.median(err) <= 1e-6 and np.quantile(err, 0.99) <= 2e-5, (float(np.median(err)), float(err.max()))
