"""mirror's step-list walk (mirror_wide_kernel, pymotion_amd/csrc/mirror.hip; reference: pymotion/ops/skeleton.py:247-344): 1 / 2 / 4 / 8 frames a wave and
16 / 8 / 4 / 2 joints of a frame a step from a host-made list held in registers.  Against the reference's chain rebuilt from oracle pieces (fk -> from_matrix ->
joint permutation -> negate -> from_global_rotations) at every tile shape, with and without a joint mapping; bit for bit against the tile kernels (same products
in the same order); a NaN stays in its frame; which instance the production dispatch picks.  The forced instances run on the -DPM_TUNING compilation
(PM_MIRROR_WIDE = frames a wave): same kernels as libpmhip.so."""
import numpy as np
import pytest

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from test_gpu_deep import chain_like, humanoid_with_hands
from test_gpu_parity import _mirror_tie_elements

pytestmark = pytest.mark.gpu
ATOL = 1e-5
AXES = {"X": (2, 3), "Y": (1, 3), "Z": (1, 2)}  # skeleton.py:310-318


def _tree(kind, J):
    if kind == "smplh":
        return np.asarray(syn.PARENTS_52, dtype=np.int32)
    if kind == "chain":
        return chain_like(J)
    if kind == "humanoid":
        return humanoid_with_hands(J)
    if kind == "star":
        return np.zeros(J, dtype=np.int32)
    return syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)


def _inputs(F, J, seed):
    rng = np.random.default_rng(seed)
    rot = rng.standard_normal((F, J, 4)).astype(np.float32)  # mirror normalises like fk does (skeleton.py:45): not unit length on purpose
    root = rng.uniform(-1, 1, (F, 3)).astype(np.float32)
    off = syn.make_offsets(J, rng, 0.1)
    return rot, root, off


def _want(rot, off, parents, mapping, axis):
    _, rm = co.fk(rot.astype(np.float64), np.zeros((len(rot), 3)), off.astype(np.float64), parents)
    g = co.quat_from_matrix(rm)
    if mapping is not None:
        g = g[:, mapping]
    for comp in AXES[axis]:
        g[..., comp] *= -1
    return co.from_global_rotations(g, parents)


SHAPES = [(17, "star"), (22, "bushy"), (44, "bushy"), (52, "smplh"), (64, "humanoid"), (65, "bushy"), (96, "chain"), (100, "bushy"), (128, "bushy"), (129, "humanoid"),
          (250, "bushy"), (300, "humanoid"), (512, "bushy")]


@pytest.mark.parametrize("fpw", [1, 2, 4, 8])
@pytest.mark.parametrize("J,kind", SHAPES)
def test_step_list_walk_against_the_oracle_composition(J, kind, fpw, monkeypatch):
    import pymotion_amd.ops.skeleton as sk

    if fpw * J > 512:
        pytest.skip("a tile is at most eight batches of 64 records")
    parents = _tree(kind, J)
    depth = int(syn.depth_of(parents).max())
    if depth > 48:
        pytest.skip("more levels than the step list holds")
    monkeypatch.setenv("PM_MIRROR_WIDE", str(fpw))
    rng = np.random.default_rng(J)
    mapping = np.arange(J)
    pairs = rng.permutation(np.arange(1, J))[: 2 * ((J - 1) // 3)].reshape(-1, 2)
    mapping[pairs[:, 0]], mapping[pairs[:, 1]] = pairs[:, 1], pairs[:, 0]  # an involution, root fixed
    with _lib.variant("tuning"):
        for F, nt, axis, mp in ((1, 1, "X", None), (fpw + 1, 2, "Y", mapping), (257, 1, "Z", None), (7 * fpw + 3, 3, "X", mapping), (1000, 3, "Y", None)):
            monkeypatch.setenv("PM_MW_NT", str(nt))
            rot, root, off = _inputs(F, J, 10 * J + F)
            got, gt, o2, _ = sk.mirror(rot, root, parents, off, None, mp, "all" if mp is None else "symmetry", axis)
            name = _lib.last_kernel_name()
            if "wide_kernel" not in name:  # a deep, narrow tree at many joints a step: more steps than the list holds
                assert kind == "chain" and fpw < 4, (name, fpw)
                continue
            assert "mirror_wide_kernel<%d," % fpw in name, name
            want = _want(rot, off, parents, mp, axis)
            err = np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max()
            assert err <= ATOL * max(1.0, depth / 32.0), (F, axis, err)
            tie_el = _mirror_tie_elements(rot, off, parents, mp)
            same = np.abs(got - want).max(-1) <= ATOL * max(1.0, depth / 32.0)
            assert same[~tie_el].all() and tie_el.mean() < 0.02, (int((~same[~tie_el]).sum()), float(tie_el.mean()))


@pytest.mark.parametrize("J,kind", [(22, "bushy"), (52, "smplh"), (100, "bushy"), (200, "bushy"), (64, "chain")])
def test_step_list_walk_gives_the_tile_kernels_bits(J, kind, monkeypatch):
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    rot, root, off = _inputs(3001, J, J)
    with _lib.variant("tuning"):
        monkeypatch.setenv("PM_MIRROR_WIDE", "0")
        monkeypatch.setenv("PM_MIRROR_DEEP", "0")
        ref = sk.mirror(rot, root, parents, off, None, None, "all", "X")[0]
        assert "mirror_kernel<" in _lib.last_kernel_name(), _lib.last_kernel_name()
        monkeypatch.delenv("PM_MIRROR_DEEP")
        ran = 0
        for fpw in (1, 2, 4, 8):
            if fpw * J > 512:
                continue
            monkeypatch.setenv("PM_MIRROR_WIDE", str(fpw))
            d = sk.mirror(rot, root, parents, off, None, None, "all", "X")[0]
            if "wide_kernel" not in _lib.last_kernel_name():
                continue
            ran += 1
            np.testing.assert_array_equal(d.view(np.int32), ref.view(np.int32), err_msg=f"fpw {fpw}")
        assert ran >= 2


@pytest.mark.parametrize("fpw", [1, 2, 4, 8])
def test_step_list_walk_keeps_a_nan_in_its_frame(fpw, monkeypatch):
    import pymotion_amd.ops.skeleton as sk

    J, F = 52, 300
    parents = _tree("smplh", J)
    rot, root, off = _inputs(F, J, 3)
    rot[70, 9, 2] = np.nan
    rot[171, 0, 0] = np.inf
    monkeypatch.setenv("PM_MIRROR_WIDE", str(fpw))
    with _lib.variant("tuning"):
        got = sk.mirror(rot, root, parents, off, None, None, "all", "X")[0]
        assert "mirror_wide_kernel<%d," % fpw in _lib.last_kernel_name()
    clean = np.ones(F, bool)
    clean[[70, 171]] = False
    assert np.isfinite(got[clean]).all()
    assert np.isnan(got[70, 9]).all() and np.isnan(got[171, 0]).all()
    want = _want(rot[clean], off, parents, None, "X")
    assert np.minimum(np.abs(got[clean] - want).max(-1), np.abs(got[clean] + want).max(-1)).max() <= ATOL


PICKS = [(39, "bushy", None), (40, "bushy", 4), (44, "bushy", 4), (52, "smplh", 4), (64, "humanoid", 4), (72, "bushy", 4), (96, "chain", 4), (100, "bushy", 4), (128, "humanoid", 2),
         (300, "bushy", 1), (130, "chain", None)]


@pytest.mark.parametrize("J,kind,fpw", PICKS)
def test_production_dispatch_of_the_step_list_walk(J, kind, fpw):
    import pymotion_amd.ops.skeleton as sk

    parents = _tree(kind, J)
    rot, root, off = _inputs(257, J, 5 * J)
    assert _lib.lib() is _lib._handles.get("prod")
    got = sk.mirror(rot, root, parents, off, None, None, "all", "Z")[0]
    name = _lib.last_kernel_name()
    if fpw is None:
        assert "wide_kernel" not in name, name
    else:
        assert "mirror_wide_kernel<%d," % fpw in name, name
    want = _want(rot, off, parents, None, "Z")
    depth = int(syn.depth_of(parents).max())
    assert np.minimum(np.abs(got - want).max(-1), np.abs(got + want).max(-1)).max() <= ATOL * max(1.0, depth / 32.0)
