"""ops/time.py interpolate_positions: the C oracle on the CPU and the HIP kernel on the
GPU, both against vectors produced by the reference (make_golden.py: gen_time), plus the reference's own
literal (ops/tests/test_time.py:12-66, atol 1e-6)."""
import numpy as np
import pytest

from conftest import assert_close, golden, up64
from oracle import c_oracle as co

CASES = ["interp_lit", "interp_clip", "interp_lead", "interp_two"]


def _axis(i):
    return i["pos"].ndim - 2  # the reference's broadcast (time.py:61-64) only supports this layout


@pytest.mark.parametrize("case", CASES)
def test_oracle_interpolate_vs_reference_golden(case):
    g = golden("time.npz")
    i, want = up64(g.get(case, "in")), g.get(case, "out64")["out"]
    assert_close(co.interpolate_positions(i["sample"], i["orig"], i["pos"], _axis(i)), want, 1e-12, case)


def test_oracle_reference_literal():
    """the numbers of ops/tests/test_time.py:31-61 themselves"""
    g = golden("time.npz")
    i = g.get("interp_lit", "in")
    got = co.interpolate_positions(i["sample"], i["orig"], i["pos"], 3)
    assert got.shape == (1, 1, 2, 9, 3)
    assert_close(got[0, 0, 0, :4], [[0.25, 0.25, 0.0], [0.875, 0.875, 0.0], [1.25, 0.75, 0.0], [6.5, 0.0, 0.75]], 1e-6)
    assert_close(got[0, 0, 1, :2], [[1.0, 1.0, 0.75], [1.0, 1.0, 0.125]], 1e-6)
    assert_close(got[0, 0, 0, 4:], [[8, 0, 1], [20, 0, 0], [32, 0, -1], [44, 0, -2], [56, 0, -3]], 1e-6)  # extrapolation


def test_oracle_any_time_axis_is_the_same_gather():
    rng = np.random.default_rng(3)
    orig = np.cumsum(rng.uniform(0.1, 1.0, 11))
    sample = rng.uniform(orig[0] - 1, orig[-1] + 1, 23)
    p = rng.standard_normal((4, 11, 5, 3))
    base = co.interpolate_positions(sample, orig, np.moveaxis(p, 1, 2), 2)   # time second to last
    assert_close(np.moveaxis(co.interpolate_positions(sample, orig, p, 1), 1, 2), base, 1e-14)
    assert_close(np.moveaxis(co.interpolate_positions(sample, orig, np.moveaxis(p, 1, 0), 0), 0, 2), base, 1e-14)


# ---- GPU ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_interpolate_vs_reference_golden(case):
    import torch

    import pymotion_amd.ops.time as tm
    import pymotion_amd.ops.time_torch as tmt

    g = golden("time.npz")
    i = g.get(case, "in")
    ax = _axis(i)
    tol = 2e-5 if case == "interp_lit" else 1e-5  # fp32 on values up to 56 (extrapolated) in the literal
    got = tm.interpolate_positions(i["sample"], i["orig"], i["pos"], ax)
    assert_close(got, g.get(case, "out_np")["out"], tol, case)
    assert got.dtype == g.get(case, "out_np")["out"].dtype
    i64 = up64(i)
    got64 = tm.interpolate_positions(i64["sample"], i64["orig"], i64["pos"], ax)
    assert got64.dtype == np.float64
    assert_close(got64, g.get(case, "out64")["out"], tol, case + " f64 door")
    for dev in ("cuda", "cpu"):
        t = [torch.from_numpy(i[k]).to(dev) for k in ("sample", "orig", "pos")]
        out = tmt.interpolate_positions(t[0], t[1], t[2], ax)
        assert out.device.type == dev and out.dtype == torch.float32
        assert_close(out.cpu().numpy(), g.get(case, "out_t")["out"], tol, case + " torch " + dev)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,axis", [((1000, 22, 3), 0), ((7, 1000, 22, 3), 1), ((22, 257, 3), -2), ((3, 64, 5), 1),
                                        ((100_003, 4), 0), ((2, 2, 7), 1), ((5, 0, 3), 0),
                                        # rows of 63 / 6 / 9 floats: dword-aligned dwordx4 per row with a 3 / 2 / 1-float tail
                                        ((333, 21, 3), 0), ((3, 500, 21, 3), 1), ((5000, 6), 0), ((4, 77, 3, 3), 1)])
def test_gpu_interpolate_vs_oracle_shapes(shape, axis):
    """every time axis, row lengths that are / are not multiples of 4 floats, an empty axis after the time axis"""
    import pymotion_amd.ops.time as tm

    rng = np.random.default_rng(abs(hash(shape)) % 1000)
    Tn = shape[axis]
    if Tn < 2:
        pytest.skip("needs two frames")
    orig = np.cumsum(rng.uniform(0.01, 0.05, Tn)).astype(np.float32)
    sample = rng.uniform(orig[0] - 0.05, orig[-1] + 0.05, 2 * Tn + 1).astype(np.float32)
    p = rng.uniform(-2, 2, shape).astype(np.float32)
    got = tm.interpolate_positions(sample, orig, p, axis)
    want = co.interpolate_positions(sample.astype(np.float64), orig.astype(np.float64), p.astype(np.float64), axis)
    assert got.shape == want.shape
    assert_close(got, want, 3e-5, str(shape))


@pytest.mark.gpu
def test_gpu_interpolate_argument_errors():
    import pymotion_amd.ops.time as tm

    p = np.zeros((5, 3), np.float32)
    t = np.arange(5, dtype=np.float32)
    with pytest.raises(ValueError):
        tm.interpolate_positions(t, t, p, 0, method="cubic")
    with pytest.raises(ValueError):
        tm.interpolate_positions(t, t[:4], p, 0)
    with pytest.raises(ValueError):
        tm.interpolate_positions(t, t[:1], p[:1], 0)
    assert tm.interpolate_positions(t[:0], t, p, 0).shape == (0, 3)
