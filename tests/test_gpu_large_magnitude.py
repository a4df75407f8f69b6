"""GPU: fk on large-magnitude data (centimetre mocap: bone offsets ~30, root positions ~200; far-away roots).

The reference accumulates in float64 (ops/skeleton.py:44) and an fp32 result can hold a coordinate of magnitude |p| only
to ulp(|p|) (3e-5 at |p| = 400): "1e-5 absolute" is a statement about metre-scale data.  The bar here (numerics):

    |pos error| <= max(1e-5, 2 ulp_fp32(largest |coordinate| of the batch))        rotmats: <= 1e-6 in this regime

which the kernels meet by giving such tiles float64 local rotations and a fixed-point (exactly additive) translation
chain, chosen per tile from the joint table and the tile's root positions (fk.hip: PREC_DYN).  Measured: 0.9 ulp at
J = 22, 1.6 ulp at J = 52 (what is left is the fp32 rotation chain times the bone lengths)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _ulp_of(x):
    return 2.0 ** (np.floor(np.log2(np.abs(x).max())) - 23)


def _cm_workload(F, parents, seed, off_scale=30.0, root_scale=200.0):
    rng = np.random.default_rng(seed)
    J = len(parents)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    root = rng.uniform(-root_scale, root_scale, (F, 3)).astype(np.float32)
    off = rng.uniform(-off_scale, off_scale, (J, 3)).astype(np.float32)
    off[0] = 0
    return rot, root, off


def _oracle(rot, root, off, parents):
    return co.fk(rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64), parents)


@pytest.mark.parametrize("J,F", [(22, 20_003), (52, 8_191), (24, 4_099), (31, 2_050), (64, 1_027), (96, 515), (130, 259), (3, 70_001)])
def test_fk_centimetre_scale_within_two_ulp_both_doors(J, F):
    import torch

    import pymotion_amd.ops.skeleton as sk
    import pymotion_amd.ops.skeleton_torch as skt

    parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if parents is None:
        parents = syn.random_parents(J, np.random.default_rng(J))
    rot, root, off = _cm_workload(F, parents, seed=J)
    p_o, r_o = _oracle(rot, root, off, parents)
    bar = max(1e-5, 2 * _ulp_of(p_o))
    pos, rm = sk.fk(rot, root, off, parents)
    assert np.abs(pos - p_o).max() <= bar, (np.abs(pos - p_o).max() / _ulp_of(p_o), "ulp")
    assert np.abs(rm - r_o).max() <= 1e-6
    # the root is the caller's value, bit for bit, on this path too (skeleton.py:49)
    np.testing.assert_array_equal(pos[:, 0].astype(np.float32), root)
    tp, tr = skt.fk(torch.from_numpy(rot).cuda(), torch.from_numpy(root).cuda(), torch.from_numpy(off).cuda(),
                    torch.from_numpy(np.asarray(parents)))
    assert np.abs(tp.cpu().numpy() - p_o).max() <= bar
    assert np.abs(tr.cpu().numpy() - r_o).max() <= 1e-6


def test_fk_metre_scale_keeps_the_absolute_bar_and_tighter_rotations():
    """human-scale data in metres stays on the fp32 path: 1e-5 absolute with a wide margin"""
    import pymotion_amd.ops.skeleton as sk

    rot, root, off, parents = syn.fk_workload(50_000, seed=21)
    p_o, r_o = _oracle(rot, root, off, parents)
    pos, rm = sk.fk(rot, root, off, parents)
    assert np.abs(pos - p_o).max() <= 2e-6
    assert np.abs(rm - r_o).max() <= 1.2e-6


def test_far_away_roots_with_metre_bones_and_mixed_tiles():
    """the decision is per tile: frames whose root is far from the origin get the fixed-point chain (one rounding of
    |p| instead of one per joint), their neighbours in other tiles stay on the fp32 path; both meet their bar"""
    import pymotion_amd.ops.skeleton as sk

    F = 4000
    rot, root, off, parents = syn.fk_workload(F, seed=22)
    root = root.copy()
    far = np.zeros(F, bool)
    far[1000:2000] = True            # a block of tiles
    far[2500::97] = True             # single frames inside otherwise near tiles
    root[far] += np.float32(1500.0)
    p_o, r_o = _oracle(rot, root, off, parents)
    pos, rm = sk.fk(rot, root, off, parents)
    err = np.abs(pos - p_o).max(axis=(1, 2))
    assert err[far].max() <= 1.01 * 2.0 ** (10 - 23), err[far].max()      # <= 1 ulp at |p| in [1024, 2048)
    near_only = np.ones(F, bool)
    for fpw in (20, 16, 12, 4):      # tiles without any far frame, for every tile size the library may pick
        for t0 in range(0, F, fpw):
            if far[t0:t0 + fpw].any():
                near_only[t0:t0 + fpw] = False
    assert near_only.sum() > F // 3
    assert err[near_only].max() <= 2e-6
    assert np.abs(rm - r_o).max() <= 1.2e-6
    np.testing.assert_array_equal(pos[:, 0].astype(np.float32), root)


@pytest.mark.parametrize("J", [22, 52])
def test_non_finite_inputs_propagate_like_the_reference_on_the_big_path(J):
    import pymotion_amd.ops.skeleton as sk

    parents = syn.PARENTS_22 if J == 22 else syn.PARENTS_52
    F = 200
    rot, root, off = _cm_workload(F, parents, seed=5)
    rot[7, 3, 1] = np.nan            # a NaN quaternion: its joint's rotation and everything below it
    rot[90, 4, 0] = np.inf          # (on the ROOT joint the kernels give an all-NaN root matrix where the reference keeps
                                     # the diagonal of to_matrix((nan,0,0,0)): the root is walked as e_r . L, and 0 * NaN = NaN)
    root[150, 2] = np.nan            # a NaN root coordinate: that row of every position of the frame
    root[33, 0] = np.inf
    with np.errstate(all="ignore"):
        p_o, r_o = _oracle(rot, root, off, parents)
    pos, rm = sk.fk(rot, root, off, parents)
    # joint positions depend on the PARENT's rotation: compare NaN patterns and the finite rest
    assert (np.isnan(pos) == np.isnan(p_o)).all()
    assert (np.isnan(rm) == np.isnan(r_o)).all()
    fin = np.isfinite(p_o)
    assert (np.isinf(pos) == np.isinf(p_o)).all()
    assert np.abs(pos[fin] - p_o[fin]).max() <= max(1e-5, 2 * _ulp_of(p_o[fin]))


@pytest.mark.parametrize("J,kind", [(6, "random"), (22, "body"), (36, "random"), (52, "smplh"), (80, "random"), (128, "chain"), (200, "random"), (300, "chain")])
def test_non_finite_translations_reach_the_rotation_rows_like_the_reference(J, kind):
    """the reference multiplies homogeneous 4 x 4 matrices (ops/skeleton.py:54-57): row r of a joint's rotation carries p_parent[r] * 0, so a
    NaN / Inf root coordinate (or offset) turns rotation rows below it into NaN and Inf positions into NaN one level further down -- on every
    walk (three lanes / quad / twelve lanes per frame, pipelined, streamed, wide), shared and per-frame offsets, quaternion and ortho6d source,
    metre and centimetre data; frames without such a value in the same tiles are untouched"""
    import pymotion_amd.ops.skeleton as sk
    from oracle import c_oracle as co

    rng = np.random.default_rng(J)
    if kind == "chain":
        parents = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
        parents[J // 2] = 0
        parents[3 * J // 4] = J // 4
    else:
        parents = {"body": syn.PARENTS_22, "smplh": syn.PARENTS_52}.get(kind)
        if parents is None:
            parents = syn.random_parents(J, rng)
    F = 131
    depth = int(syn.depth_of(parents).max())
    f64 = lambda x: x.astype(np.float64)  # noqa: E731
    for osc, rsc in ((0.2, 2.0), (20.0, 150.0)):
        rot = rng.standard_normal((F, J, 4)).astype(np.float32)
        root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
        off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
        off[0] = 0
        root[5, 1] = np.nan
        root[40, 0] = np.inf
        root[41, 2] = -np.inf
        root[100] = np.nan
        for offs in (off, np.broadcast_to(off, (F, J, 3)).copy()):
            if offs.ndim == 3 and J > 1:
                offs[77, J // 2, 1] = np.inf      # a per-frame offset: positions from that joint down, rotations from its children down
                offs[90, J - 1, 0] = np.nan       # a leaf: its own position only
            with np.errstate(all="ignore"):
                p_o, r_o = co.fk(f64(rot), f64(root), f64(offs), parents)
            pos, rm = sk.fk(rot, root, offs, parents)
            name = _lib.last_kernel_name()
            assert (np.isnan(rm) == np.isnan(r_o)).all(), (name, osc, offs.ndim, np.argwhere(np.isnan(rm) != np.isnan(r_o))[:4])
            assert (np.isnan(pos) == np.isnan(p_o)).all(), (name, osc, offs.ndim)
            assert (np.isinf(pos) == np.isinf(p_o)).all(), (name, osc, offs.ndim)
            fin = np.isfinite(r_o)
            assert np.abs(rm[fin] - r_o[fin]).max() <= 1e-5
            fin = np.isfinite(p_o)
            # (a tile that holds a NaN / Inf translation walks in fp32 -- its finite frames too: the fp32 chain's bound, as in test_gpu_deep.py)
            assert np.abs(pos[fin] - p_o[fin]).max() <= max(1e-5, 3 * _ulp_of(p_o[fin]), 4e-7 * depth * osc * 3), name
        if J <= 128:  # the fused ortho6d source shares the walks
            x = rng.standard_normal((F, J, 3, 2)).astype(np.float32)
            with np.errstate(all="ignore"):
                p_o, r_o = co.fk(co.o6d_to_quat(f64(x)), f64(root), f64(off), parents)
            pos, rm = sk.fk_from_ortho6d(x, root, off, parents)[:2]
            assert (np.isnan(rm) == np.isnan(r_o)).all() and (np.isnan(pos) == np.isnan(p_o)).all() and (np.isinf(pos) == np.isinf(p_o)).all(), _lib.last_kernel_name()
    # a NaN in the SHARED offsets table: every frame
    off2 = off.copy()
    off2[J // 2 if J > 1 else 0, 2] = np.nan
    if J > 2:
        with np.errstate(all="ignore"):
            p_o, r_o = co.fk(f64(rot), f64(root), f64(off2), parents)
        pos, rm = sk.fk(rot, root, off2, parents)
        assert (np.isnan(rm) == np.isnan(r_o)).all() and (np.isnan(pos) == np.isnan(p_o)).all(), _lib.last_kernel_name()


def test_per_frame_offsets_at_centimetre_scale():
    import pymotion_amd.ops.skeleton as sk

    for J, parents in ((22, syn.PARENTS_22), (52, syn.PARENTS_52)):
        F = 1501
        rot, root, off = _cm_workload(F, parents, seed=40 + J)
        offs = (off[None] * np.linspace(0.5, 1.5, F, dtype=np.float32)[:, None, None]).astype(np.float32)
        p_o, r_o = _oracle(rot, root, offs, parents)
        pos, rm = sk.fk(rot, root, offs, parents)
        assert np.abs(pos - p_o).max() <= 2 * _ulp_of(p_o)
        assert np.abs(rm - r_o).max() <= 1e-6


def test_fused_ortho6d_at_centimetre_scale():
    import pymotion_amd.ops.skeleton as sk

    parents = syn.PARENTS_52
    F = 8191
    rng = np.random.default_rng(77)
    x = rng.standard_normal((F, 52, 3, 2)).astype(np.float32)
    _, root, off = _cm_workload(F, parents, seed=52)  # the skeleton and roots of test_fk_centimetre_scale_within_two_ulp_both_doors[52]: the same bar means the same thing
    with np.errstate(all="ignore"):
        q_o = co.o6d_to_quat(x.astype(np.float64))
    p_o, r_o = _oracle(q_o, root, off, parents)
    for want_q in (True, False):
        out = sk.fk_from_ortho6d(x, root, off, parents, return_quat=want_q)
        # big tiles run Gram-Schmidt in float64 like the reference's chain (fk.hip: local_from_o6d): fk's own bar, on ALL frames
        assert np.abs(out[0] - p_o).max() <= max(1e-5, 2 * _ulp_of(p_o)), (np.abs(out[0] - p_o).max() / _ulp_of(p_o), "ulp")
        assert np.abs(out[1] - r_o).max() <= 1e-6, np.abs(out[1] - r_o).max()
        if want_q:
            assert np.minimum(np.abs(out[2] - q_o).max(-1), np.abs(out[2] + q_o).max(-1)).max() <= 1e-5


def test_threshold_edges_take_a_consistent_path():
    """bones of exactly 1.0 and roots of exactly 16.0 sit on the decision boundary: whichever side, the result is in the bar"""
    import pymotion_amd.ops.skeleton as sk

    rot, root, off, parents = syn.fk_workload(999, seed=9)
    for bone, far in ((1.0, 2.0), (np.nextafter(np.float32(1.0), np.float32(0)), 2.0), (0.3, 16.0), (0.3, np.nextafter(np.float32(16.0), np.float32(0)))):
        off2 = off.copy()
        off2[5, 1] = bone
        root2 = root.copy()
        root2[::50, 0] = far
        p_o, r_o = _oracle(rot, root2, off2, parents)
        pos, rm = sk.fk(rot, root2, off2, parents)
        assert np.abs(pos - p_o).max() <= 4e-6
        assert np.abs(rm - r_o).max() <= 1.2e-6


def test_translation_equivariance_at_full_size_centimetre_scale():
    """size-independent property at 2^20 frames: moving every root by d moves every joint by d (up to the final rounding),
    and rotations do not change at all"""
    import torch

    import pymotion_amd.ops.skeleton_torch as skt

    F = 1 << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    rot = torch.randn((F, 22, 4), generator=g, device="cuda")
    root = (torch.rand((F, 3), generator=g, device="cuda") * 2 - 1) * 200
    off_np = np.random.default_rng(3).uniform(-30, 30, (22, 3)).astype(np.float32)
    off_np[0] = 0
    off = torch.from_numpy(off_np).cuda()
    par = torch.from_numpy(syn.PARENTS_22)
    d = torch.tensor([64.0, -128.0, 32.0], device="cuda")     # powers of two: root + d is exact where no bit falls off
    p0, r0 = skt.fk(rot, root, off, par)
    p1, r1 = skt.fk(rot, root + d, off, par)
    assert bool(torch.equal(r0, r1))
    assert float(((p1 - p0) - d).abs().max()) <= 3 * 2.0 ** (9 - 23)   # a few ulp at |p| < 1024
    n = 1 << 14
    sl = slice(F // 2, F // 2 + n)
    p_o, r_o = _oracle(rot[sl].cpu().numpy(), root[sl].cpu().numpy(), off_np, syn.PARENTS_22)
    assert np.abs(p0[sl].cpu().numpy() - p_o).max() <= 2 * _ulp_of(p_o)


def _chain(J):
    p = np.arange(-1, J - 1)
    p[0] = 0
    return p


@pytest.mark.parametrize("J,kind", [(22, "body"), (52, "body"), (12, "chain"), (31, "random"), (96, "random"), (130, "random"), (128, "chain")])
def test_root_dual_quaternions_within_two_ulp_at_centimetre_scale(J, kind):
    """to_root_dual_quat's dual part is the running translation TIMES the running quaternion, 0.5 (0, T_j) (x) Q_j: the error of
    an fp32 quaternion chain (4e-7 after ten joints) times |T| = 400 read 4-5 ulp of the largest component at J = 52 (round 2's
    bar here: 8 ulp).  The reference composes in float64 (skeleton.py:230-241, dual_quat.py:32); big-magnitude tiles now do too
    (dq.hip: float64 quaternion chain across the quad, fixed-point translations), and the bar is fk's:

        |error| <= max(1e-6, 2 ulp_fp32(largest |component| of the batch))          (measured 0.7 - 1.1 ulp)

    Metre-scale data keeps the fp32 step: its error is 3-6e-7 ABSOLUTE (2.7 / 4 ulp of components of ~3), under the 1e-6 floor.
    The one exception is stated, not hidden: a 128-joint CHAIN of 30-unit bones reads 2.3 ulp -- there the fp32 rotation of
    each offset (5e-6 per joint, a random walk over 127 joints) is what is left."""
    import pymotion_amd.ops.skeleton as sk

    parents = {"body": syn.PARENTS_22 if J == 22 else syn.PARENTS_52, "chain": _chain(J)}.get(kind)
    if parents is None:
        parents = syn.random_parents(J, np.random.default_rng(J))
    for osc, rsc in ((0.3, 2.0), (30.0, 200.0)):
        rng = np.random.default_rng(J)
        F = 6001 if J <= 64 else 1501
        rot = rng.standard_normal((F, J, 4))
        rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
        root = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
        off = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
        off[0] = 0
        d = sk.to_root_dual_quat(rot, root, parents, off)
        d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
        ulps = 3.0 if (kind == "chain" and J > 64) else 2.0
        floor = 1e-6 if J <= 64 else 1e-5   # (deep metre-scale trees on the fp32 step: north_star's absolute bar)
        err = np.abs(d - d_o).max()
        assert err <= max(floor, ulps * _ulp_of(d_o)), (err, err / _ulp_of(d_o), "ulp", osc)
        assert np.abs(d[..., :4] - d_o[..., :4]).max() <= (1e-7 if osc > 1 else 2e-6)  # the float64 chain: real part to fp32 rounding
        t, q = sk.from_root_dual_quat(d, parents)
        assert np.abs(q - rot).max() <= (2e-6 if J <= 64 else 4e-6)
        assert np.abs(t[:, 1:] - off[1:]).max() <= 4e-6 * max(1.0, np.abs(d_o).max())  # decode = 2 qd (x) conj(qr): relative to |dq|


def test_root_dual_quaternions_mixed_tiles_far_roots_and_non_finite_inputs():
    """the arithmetic is chosen per tile: a block of frames with far-away roots takes the precise step, its neighbours the
    fp32 one; NaN / Inf inputs come out in the reference's pattern on both"""
    import pymotion_amd.ops.skeleton as sk

    for J, parents in ((22, syn.PARENTS_22), (52, syn.PARENTS_52), (12, _chain(12))):
        F = 3000
        rng = np.random.default_rng(100 + J)
        rot = rng.standard_normal((F, J, 4))
        rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
        root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
        off = rng.uniform(-0.3, 0.3, (J, 3)).astype(np.float32)
        off[0] = 0
        far = np.zeros(F, bool)
        far[1000:1800] = True
        far[2200::131] = True
        root[far] += np.float32(700.0)
        # (joints at depth >= 2: the root and its children are composed with an identity here where the reference copies them,
        # so a NaN in ONE component of theirs comes out in all four -- same joints, wider pattern)
        deep = [j for j in range(J) if syn.depth_of(parents)[j] >= 2]
        rot[1100, deep[1], 2] = np.nan  # inside a precise tile
        rot[1200, deep[0], 0] = np.inf  # Inf in a precise tile
        rot[50, deep[2], 1] = np.nan    # inside an fp32 tile
        root[1300, 1] = np.nan          # a NaN root: its tile falls back to the fp32 step, which propagates it
        skip = np.zeros(F, bool)
        skip[1296:1312] = True          # (that tile's other frames are fp32-step frames with far roots: judged by neither bar)
        with np.errstate(all="ignore"):
            d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
        d = sk.to_root_dual_quat(rot, root, parents, off)
        assert (np.isnan(d) == np.isnan(d_o)).all()
        fin = np.isfinite(d_o)
        assert (np.isinf(d) == np.isinf(d_o)).all()
        err = np.abs(d[fin] - d_o[fin])
        frame_of = np.broadcast_to(np.arange(F)[:, None, None], d.shape)[fin]
        assert err[(far & ~skip)[frame_of]].max() <= 2 * 2.0 ** (9 - 23)        # components up to ~600: 2 ulp there
        near_only = np.ones(F, bool)
        for fpw in (16, 8, 4):
            for t0 in range(0, F, fpw):
                if far[t0:t0 + fpw].any():
                    near_only[t0:t0 + fpw] = False
        assert err[(near_only & ~skip)[frame_of]].max() <= 1e-6


@pytest.mark.parametrize("J,kind", [(22, "body"), (31, "random"), (52, "body"), (12, "chain")])
def test_root_dual_quaternions_of_non_unit_rotations_at_centimetre_scale(J, kind):
    """to_root_dual_quat does NOT normalise its inputs (skeleton.py:207-244), and with |q| != 1 a rotated offset grows by |Q_parent|^2
    down the chain: the precise step's fixed-point bound (unit rotations) does not hold, so tiles holding an off-unit quaternion keep the
    fp32 step (round-3 ADVICE: norm 1.05 is 2.2x after eight ancestors, past the word's headroom -- garbage, not rounding).  Bar: the
    fp32 step's, 8 ulp of the largest component (round 2's bar for this op), on norms 0.8 ... 1.2; and unit tiles NEXT to them still
    take the precise step (2 ulp)."""
    import pymotion_amd.ops.skeleton as sk

    parents = {"body": syn.PARENTS_22 if J == 22 else syn.PARENTS_52, "chain": _chain(J)}.get(kind)
    if parents is None:
        parents = syn.random_parents(J, np.random.default_rng(J))
    rng = np.random.default_rng(300 + J)
    F = 4096
    rot = rng.standard_normal((F, J, 4))
    rot = rot / np.linalg.norm(rot, axis=-1, keepdims=True)
    scaled = np.zeros(F, bool)
    scaled[: F // 2] = True                                   # first half: every quaternion off unit length
    rot[scaled] *= rng.uniform(0.8, 1.2, (int(scaled.sum()), J, 1))
    rot = rot.astype(np.float32)
    root = rng.uniform(-200, 200, (F, 3)).astype(np.float32)
    off = rng.uniform(-30, 30, (J, 3)).astype(np.float32)
    off[0] = 0
    d = sk.to_root_dual_quat(rot, root, parents, off)
    d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
    assert np.isfinite(d).all()
    for sel, ulps in ((scaled, 8.0), (~scaled, 2.0)):
        err = np.abs(d[sel] - d_o[sel]).max()
        assert err <= ulps * _ulp_of(d_o[sel]), (err / _ulp_of(d_o[sel]), "ulp", "scaled" if ulps == 8.0 else "unit")


@pytest.mark.usefixtures("lane_per_frame_at_test_sizes")
@pytest.mark.parametrize("J,kind", [(22, "body"), (31, "random"), (20, "random"), (36, "random")])
def test_root_dual_quaternions_raw_abi_with_and_without_the_scale_hint(J, kind):
    """The front doors pass max |offsets| as a host-side hint (pm_to_root_dq_hint_f32): big-bone skeletons of 20 joints or more then
    take the lane-per-frame kernel (float64 state, scale-blind).  The raw ABI without the hint cannot see the scale and keeps the
    per-tile precise step.  Same bar for both: max(1e-6, 2 ulp of the largest component)."""
    import ctypes as C

    import torch

    from pymotion_amd import _lib

    parents = syn.PARENTS_22 if kind == "body" else syn.random_parents(J, np.random.default_rng(J))
    rng = np.random.default_rng(500 + J)
    F = 5003
    rot = rng.standard_normal((F, J, 4))
    rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
    root = rng.uniform(-200, 200, (F, 3)).astype(np.float32)
    off = rng.uniform(-30, 30, (J, 3)).astype(np.float32)
    off[0] = 0
    d_o = co.to_root_dual_quat(rot.astype(np.float64), root.astype(np.float64), parents, off.astype(np.float64))
    bar = max(1e-6, 2 * _ulp_of(d_o))
    rt, gt, ot = (torch.from_numpy(x).cuda() for x in (rot, root, off))
    out = torch.empty((F, J, 8), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pp = np.asarray(parents, dtype=np.int32).ctypes.data_as(C.c_void_p)
    names = []
    for hint in (None, float(np.abs(off).max()), -1.0, float("nan")):
        out.zero_()
        if hint is None:
            _lib.call("pm_to_root_dq_f32", p(rt), p(gt), pp, p(ot), F, J, p(out), None)
        else:
            _lib.call("pm_to_root_dq_hint_f32", p(rt), p(gt), pp, p(ot), F, J, p(out), C.c_float(hint), None)
        names.append(_lib.last_kernel_name())
        err = np.abs(out.cpu().numpy() - d_o).max()
        assert err <= bar, (hint, err / _ulp_of(d_o), "ulp", names[-1])
    if kind == "body":  # (random trees may hold more open branch points than the lane-per-frame kernel's register slots: they stay on the tile kernels)
        assert "ring_kernel" in names[1] or "deep_kernel" in names[1], names  # the hint routes big bones to the lane-per-frame kernel
    assert names[0] == names[2] == names[3] and "ring_kernel" not in names[0] and "deep_kernel" not in names[0], names


@pytest.mark.parametrize("J", [64, 110, 130])
def test_root_dual_quaternions_on_the_deep_chains_the_fuzz_run_found(J):
    """round 5's randomised fuzz runs read 4.1 ulp on a 110-joint skeleton of two 55-deep chains of 30-unit bones and 3.6 ulp on a 65-deep one
    (gpurun_out/dq_fuzz_fail.txt), round 6's 3.4 ulp on a 32-deep one: the tile kernels' precise step rotated each bone in fp32, one rounding a joint, a
    random walk down the chain -- and its fixed-point words had a resolution of a whole fp32 ulp of the result there.  From twelve levels on the step
    now rotates the bones in float64 and scales the words to the range they have (dq.hip: dq_step_rot_f64, fx_scale_exact).  Pinned here, seeded, batch
    shapes that fill one tile and several:

        |error| <= max(1e-5, 2 ulp_fp32(largest |component|))            (measured 1.8 ulp on both; 4.1 / 3.6 before)

    and the measured worst case in ulps printed beside it."""
    import pymotion_amd.ops.skeleton as sk

    par = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    par[J // 2] = 0                                   # two chains off the root, as tests/test_gpu_fuzz.py::wide_skeletons draws them
    depth = J // 2
    worst = 0.0
    for seed in range(8):
        for lead in ((1,), (17,), (130,)):
            rng = np.random.default_rng(1000 * J + seed)
            rot = rng.standard_normal(lead + (J, 4))
            rot = (rot / np.linalg.norm(rot, axis=-1, keepdims=True)).astype(np.float32)
            gpos = (rng.uniform(-7, 7, lead + (3,)) * 30.0).astype(np.float32)
            off = (rng.uniform(-1, 1, (J, 3)) * 30.0).astype(np.float32)
            off[0] = 0
            d = sk.to_root_dual_quat(rot, gpos, par, off)
            d_o = co.to_root_dual_quat(rot.astype(np.float64), gpos.astype(np.float64), par, off.astype(np.float64))
            ulp = _ulp_of(d_o)
            err = np.abs(d - d_o).max()
            worst = max(worst, err / ulp)
            assert err <= max(1e-5, 2 * ulp), (seed, lead, err, err / ulp, "ulp")
            assert np.abs(d[..., :4] - d_o[..., :4]).max() <= 2e-7   # the float64 quaternion chain: the real part to fp32 rounding
    print(f"J={J} depth={depth}: worst {worst:.2f} ulp of the largest component (bar 2)")


# which walks give an all-NaN ROOT matrix for a root quaternion with an infinite component ("Which kernel a call takes"):
# the reference's to_matrix((nan, 0, 0, 0)) keeps 1 on the diagonal (quat.py:293-315: 1 - (0 + 0)) and copies that matrix into the root's
# transform as it is (skeleton.py:46-49).  Walks in which the root takes NO step (four joints of a frame at a time, a wave per frame) do the
# same; walks that run the root through the common step multiply it with the seed row e_r, and 0 x NaN = NaN fills the row.
_INF_ROOT_CASES = [  # (J, kind, F, kernel-name fragment, root matrix like the reference?)
    (6, "random", 4000, "fk_kernel<20", False),
    (10, "random", 4000, "fk_kernel<16", False),
    (22, "body", 4000, "fk_kernel<16", False),
    (36, "chain", 4000, "fk_kernel<8", False),
    (52, "smplh", 4000, "fk_pipe_kernel<4, 4", True),     # tree_walk_w4: the root's slot holds L_0 as parked
    (60, "chain", 4000, "fk_pipe_kernel<4, 4", False),    # a chain keeps the twelve-lane walk (seed row)
    (200, "random", 600, "fk_wide_kernel", True),
    (256, "chain", 4200, "fk_stream_kernel", False),
]


@pytest.mark.parametrize("J,kind,F,kernel,like_reference", _INF_ROOT_CASES)
def test_infinite_root_quaternion_which_kernels_keep_the_reference_diagonal(J, kind, F, kernel, like_reference):
    import pymotion_amd.ops.skeleton as sk

    parents = {"body": syn.PARENTS_22, "smplh": syn.PARENTS_52, "chain": _chain(J)}.get(kind)
    if parents is None:
        parents = syn.random_parents(J, np.random.default_rng(J))
    rng = np.random.default_rng(J)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)
    root = rng.uniform(-2, 2, (F, 3)).astype(np.float32)
    off = rng.uniform(-0.3, 0.3, (J, 3)).astype(np.float32)
    off[0] = 0
    f = F // 3
    rot[f, 0] = (np.inf, 0.0, 0.0, 0.0)
    with np.errstate(all="ignore"):
        p_o, r_o = _oracle(rot, root, off, parents)
    pos, rm = sk.fk(rot, root, off, parents)
    assert kernel in _lib.last_kernel_name(), _lib.last_kernel_name()
    want = r_o[f, 0]
    assert np.isnan(want).sum() == 6 and (np.diag(want) == 1).all()          # the reference: NaN off the diagonal, 1 on it
    if like_reference:
        assert (np.isnan(rm[f, 0]) == np.isnan(want)).all() and (np.diag(rm[f, 0]) == 1).all()
    else:
        assert np.isnan(rm[f, 0]).all()                                        # the stated divergence, on these kernels only
    # everything else is the reference's either way: the root's position, every joint below it (NaN x anything), the other frames
    np.testing.assert_array_equal(pos[f, 0].astype(np.float32), root[f])
    assert np.isnan(rm[f, 1:]).all() and np.isnan(r_o[f, 1:]).all()
    others = np.arange(F) != f
    assert np.abs(rm[others] - r_o[others]).max() <= 2e-6 * max(1.0, J / 12.0)
    assert (np.isnan(pos[others]) == np.isnan(p_o[others])).all()
