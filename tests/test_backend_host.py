"""CPU-only: the host-side helpers of the NumPy door (threaded dtype conversion into fresh / reused memory,
staging-buffer reuse, interpolation coefficients) -- no device calls."""
import numpy as np

from pymotion_amd import _backend as B


def test_parallel_copyto_matches_astype_for_every_dtype_pair():
    rng = np.random.default_rng(0)
    n = (B._PAR_MIN_BYTES // 4) * 3 + 12345  # above the threshold, not a multiple of the per-thread step
    src32 = rng.standard_normal(n).astype(np.float32)
    for src, dt in ((src32, np.float64), (src32.astype(np.float64), np.float32), (src32, np.float32),
                    ((src32 * 100).astype(np.int64), np.float32), (src32.reshape(-1, 3)[: n // 3], np.float64)):
        dst = np.empty(src.shape, dtype=dt)
        B._parallel_copyto(dst, src)
        np.testing.assert_array_equal(dst, src.astype(dt))
    small = np.arange(7, dtype=np.float32)
    out = np.empty(7, np.float64)
    B._parallel_copyto(out, small)  # below the threshold: the plain path
    np.testing.assert_array_equal(out, small)
    z = np.empty((0, 3), np.float64)
    B._parallel_copyto(z, np.empty((0, 3), np.float32))


def test_host_staging_buffers_are_reused_by_size_class():
    st = B._HostStage(cap_bytes=1 << 26)
    a = st.get(5 << 20)
    assert a.dtype == np.float32 and a.nbytes == 8 << 20  # next power of two
    st.put(a)
    b = st.get(6 << 20)
    assert b is a  # same class -> the same, already touched, memory
    c = st.get(6 << 20)
    assert c is not a
    st.put(b)
    st.put(c)
    big = st.get(1 << 27)
    st.put(big)  # over the cap: dropped, not cached
    assert st.cached <= st.cap


def test_interp_coefficients_follow_the_reference_formula():
    """ops/time.py:49-54: searchsorted(left) - 1, clamped to [0, T-2]; weights may leave [0, 1] (extrapolation)"""
    orig = np.array([0.0, 2.0, 3.0, 4.0, 5.0])
    sample = np.array([-1.0, 0.0, 0.5, 2.0, 2.5, 5.0, 8.0])
    idx, w = B.NumpyBackend.interp_coefficients(sample, orig)
    assert idx.dtype == np.int32
    np.testing.assert_array_equal(idx, [0, 0, 0, 0, 1, 3, 3])
    np.testing.assert_allclose(w, [-0.5, 0.0, 0.25, 1.0, 0.5, 1.0, 4.0])
