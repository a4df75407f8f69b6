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


def test_euler_order_arrays_are_compressed_to_what_repeats():
    """quat.from_euler / to_euler take an order array shaped like euler (quat.py:51-62); the host encodes it as
    uint8 and keeps a constant array as one triple, a per-joint pattern tiled over the frames as a [J, 3] table."""
    import pytest

    from pymotion_amd import _ops

    rng = np.random.default_rng(1)
    F, J = 50, 7
    per_joint = np.array(list("xyz"))[rng.permuted(np.tile(np.arange(3), (J, 1)), axis=1)]
    tiled = np.tile(per_joint, (F, 1, 1))
    codes, mode = _ops._order_codes(tiled, (F, J))
    assert mode == J and codes.shape == (J, 3) and codes.dtype == np.uint8
    np.testing.assert_array_equal(np.array(list("xyz"))[codes], per_joint)
    c2, m2 = _ops._order_codes(tiled.astype("S1"), (F, J))  # bytes strings take the same fast path
    assert m2 == J and np.array_equal(c2, codes)
    const = np.tile(np.array(["z", "x", "y"]), (F, J, 1))
    c0, m0 = _ops._order_codes(const, (F, J))
    assert m0 == 0 and c0.tolist() == [2, 0, 1]
    odd = tiled.copy()
    odd[17, 3] = odd[17, 3][::-1]  # one frame differs: an order per element
    c1, m1 = _ops._order_codes(odd, (F, J))
    assert m1 == 1 and c1.shape == (F * J, 3)
    assert _ops._order_codes(per_joint, (J,))[1] == 1          # 2-D euler [J, 3]: no frame axis to repeat over
    assert _ops._order_table(per_joint, (F, J)) [1] == J
    assert _ops._order_table(np.tile(np.array(["x", "y", "z"]), (J, 1)), (F, J))[1] == 0
    with pytest.raises(ValueError):
        _ops._order_codes(np.array([["x", "y", "w"]]), (1,))
    with pytest.raises(AssertionError):
        _ops._order_codes(tiled, (F, J + 1))
    with pytest.raises(ValueError):
        _ops._order_table(per_joint, (F, J + 1))


def test_torch_door_memo_is_keyed_on_the_tensor_object_not_its_address():
    """The memo of device-resident `parents` must miss for a different tensor object even when address, version,
    shape and dtype all coincide (caching-allocator reuse), hit for the same object, and miss after an in-place edit."""
    from pymotion_amd._backend import TorchBackend

    class FakeTensor:  # only what the memo looks at
        def __init__(self, version=0):
            self._version = version

    table = {}
    a, b = FakeTensor(), FakeTensor()
    assert TorchBackend._memo_get(table, "parents", a) is None
    TorchBackend._memo_put(table, "parents", a, "topology-a")
    assert TorchBackend._memo_get(table, "parents", a) == "topology-a"
    assert TorchBackend._memo_get(table, "parents", b) is None          # another object: never a hit
    a._version = 1                                                       # in-place edit
    assert TorchBackend._memo_get(table, "parents", a) is None
    TorchBackend._memo_put(table, "parents", a, "topology-a1")
    TorchBackend._memo_put(table, "joints_mapping", b, "mapping-b")      # independent slots do not evict each other
    assert TorchBackend._memo_get(table, "parents", a) == "topology-a1"
    assert TorchBackend._memo_get(table, "joints_mapping", b) == "mapping-b"
    del a                                                                # the entry must not keep the tensor alive
    import gc

    gc.collect()
    assert table["parents"][0]() is None
    assert TorchBackend._memo_get(table, "parents", FakeTensor(1)) is None


def test_device_pool_never_hands_a_block_to_another_device(monkeypatch):
    """cached device blocks are keyed by the HIP device they were allocated on."""
    import ctypes as C

    from pymotion_amd import _backend, _lib

    next_ptr = [0x1000]
    freed = []

    def fake_call(name, *args):
        if name == "pm_malloc":
            C.cast(args[0], C.POINTER(C.c_void_p))[0] = next_ptr[0]
            next_ptr[0] += 0x1000
        elif name == "pm_free":
            freed.append(args[0].value)

    monkeypatch.setattr(_lib, "call", fake_call)
    pool = _backend._DevPool(cap_bytes=1 << 20)
    p0, c0 = pool.get(1000, dev=0)
    pool.put(p0, c0, dev=0)
    p1, c1 = pool.get(1000, dev=1)          # same size class, other device: must be a fresh allocation
    assert p1 != p0 and c1 == c0
    p0b, _ = pool.get(1000, dev=0)          # same device: the cached block comes back
    assert p0b == p0
    pool.put(p0b, c0, dev=0)
    pool.put(p1, c1, dev=1)
    pool.trim()
    assert sorted(freed) == sorted([p0, p1]) and pool.cached == 0


def _fake_allocator(monkeypatch):
    """_lib.call with pm_malloc / pm_host_alloc / pm_free / pm_host_free backed by real host memory and counted"""
    import ctypes as C

    from pymotion_amd import _lib

    live = {}
    stats = {"dev": 0, "host": 0, "freed": []}

    def fake_call(name, *args):
        if name in ("pm_malloc", "pm_host_alloc"):
            buf = (C.c_uint8 * int(args[1]))()
            addr = C.addressof(buf)
            live[addr] = buf
            C.cast(args[0], C.POINTER(C.c_void_p))[0] = addr
            stats["dev" if name == "pm_malloc" else "host"] += 1
        elif name in ("pm_free", "pm_host_free"):
            a = args[0].value if hasattr(args[0], "value") else args[0]
            assert a in live, "freed twice or never allocated"
            del live[a]
            stats["freed"].append(a)

    monkeypatch.setattr(_lib, "call", fake_call)
    return live, stats


def test_small_call_arenas_are_shared_by_threads_that_come_and_go(monkeypatch):
    """an arena (4 MB device + 4 MB page-locked) is checked out per op and handed back, not kept per thread: fifty
    short-lived threads leave one arena behind, and trim() frees it"""
    import threading

    from pymotion_amd import _backend

    live, stats = _fake_allocator(monkeypatch)
    pool = _backend._ArenaPool()
    monkeypatch.setattr(_backend, "_arenas", pool)

    def op():
        a = pool.get(0)
        assert a.host.shape == (_backend._ARENA_BYTES,)
        pool.put(a)

    for _ in range(50):
        t = threading.Thread(target=op)
        t.start()
        t.join()
    assert stats["dev"] == 1 and stats["host"] == 1
    a, b = pool.get(0), pool.get(0)          # two ops in flight at once: two arenas
    assert a is not b and stats["dev"] == 2
    c = pool.get(1)                          # another device never takes device 0's
    assert c.dev == 1 and stats["dev"] == 3
    for x in (a, b, c):
        pool.put(x)
    pool.trim()
    assert not live and len(stats["freed"]) == 6
    d = pool.get(0)                          # one that is never handed back frees itself
    del d
    import gc

    gc.collect()
    assert not live


def test_numpy_door_unroll_pairs_are_pooled_and_freed_on_failure(monkeypatch):
    """the one-pass scans' workspace pair of the NumPy door is checked out of a process-wide list per call; a failed
    call frees its blocks instead of dropping them"""
    import threading

    from pymotion_amd import _backend

    live, stats = _fake_allocator(monkeypatch)
    monkeypatch.setattr(_backend, "_np_pairs_idle", {})

    def make():
        import ctypes as C

        ptrs = []
        for _ in range(2):
            d = C.c_void_p()
            _backend._lib.call("pm_malloc", C.byref(d), 1 << 10)
            ptrs.append(d.value)
        return _backend._UnrollPair(ptrs[0], ptrs[1], None)

    key = ("numpy", 0)

    def scan():
        p = _backend._np_pair_get(0, 100, make)
        use, other, words = p.take()
        p.done(7)
        _backend._unroll_pair_release(key, p)

    for _ in range(20):
        t = threading.Thread(target=scan)
        t.start()
        t.join()
    assert stats["dev"] == 2                          # one pair for all of them
    p = _backend._np_pair_get(0, 100, make)
    assert p.dirty[1 - p.cur] in (0, 7) and stats["dev"] == 2   # and its alternation state travelled with it
    _backend._unroll_pair_drop(key, p)                # a failed scan
    assert len(live) == 0 and len(stats["freed"]) == 2
    assert _backend._np_pair_get(0, _backend._UNROLL_WS_BYTES + 1, make) is None   # too big for a pair: the plain entry point
    q = _backend._np_pair_get(0, 100, make)
    _backend._unroll_pair_release(key, q)
    _backend._np_pairs_trim()
    assert not live
