"""Array adapters between the reference-style Python API and the device-pointer C ABI.

Two front doors, one engine:

* ``NUMPY``  -- ``np.ndarray`` in / out.  Host arrays are cast to contiguous fp32, copied to a
  pooled device buffer (``pm_malloc`` / ``pm_memcpy_h2d``), the kernel runs on the default
  stream, results are copied back and cast to the dtype the reference would return.
  No torch import.
* ``TORCH``  -- ``torch.Tensor`` in / out.  HIP tensors are used zero-copy (``data_ptr()`` on
  torch's current stream); CPU tensors are moved to ``cuda:current`` and the result moved back,
  mirroring "result lives where the input lives" of the reference's eager twins.

Both compute in fp32 on the GPU (BASELINE.json north_star).  There is no autograd support:
the reference gets it for free from eager torch ops, hand-written kernels do not (out of
scope, stated in DESIGN.md); tensors that require grad are rejected instead of silently
detached.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib


def _prod(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


# ---------------------------------------------------------------------------------------------
# NumPy front door
# ---------------------------------------------------------------------------------------------
def _current_device():
    d = C.c_int(0)
    _lib.call("pm_get_device", C.byref(d))
    return d.value


class _DevPool:
    """Tiny size-class pool over pm_malloc so repeated NumPy-path calls do not pay hipMalloc.

    Free lists are keyed by HIP device: pm_malloc allocates on the calling thread's current device, and a block
    cached while device 0 was current must never be handed to a call that runs on device 1
    (``torch.cuda.set_device(local_rank)`` in the same process, threads on different GPUs)."""

    def __init__(self, cap_bytes=4 << 30):
        self.free = {}      # (device, size class) -> [ptr]
        self.cached = 0
        self.cap = cap_bytes
        self.lock = threading.Lock()

    @staticmethod
    def _cls(nbytes):
        n = max(int(nbytes), 256)
        return 1 << (n - 1).bit_length()

    def get(self, nbytes, dev):
        c = self._cls(nbytes)
        with self.lock:
            lst = self.free.get((dev, c))
            if lst:
                self.cached -= c
                return lst.pop(), c
        p = C.c_void_p()
        try:
            _lib.call("pm_malloc", C.byref(p), c)
        except _lib.PmhipError:
            self.trim()
            _lib.call("pm_malloc", C.byref(p), c)
        return p.value, c

    def put(self, ptr, c, dev):
        with self.lock:
            if self.cached + c <= self.cap:
                self.free.setdefault((dev, c), []).append(ptr)
                self.cached += c
                return
        _lib.call("pm_free", C.c_void_p(ptr))

    def trim(self):
        with self.lock:
            blocks = [p for lst in self.free.values() for p in lst]
            self.free.clear()
            self.cached = 0
        for p in blocks:
            _lib.call("pm_free", C.c_void_p(p))  # hipFree takes a pointer of any device


_pool = _DevPool()


class _DevBuf:
    def __init__(self, nbytes, dev):
        self.dev = dev
        self.ptr, self.cls = _pool.get(nbytes, dev)
        self.nbytes = nbytes

    def release(self):
        if self.ptr is not None:
            _pool.put(self.ptr, self.cls, self.dev)
            self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def trim():
    """Give the NumPy door's cached device blocks and host staging buffers back (public: ``pymotion_amd.trim()``)."""
    _pool.trim()
    _stage.trim()
    _pinned.trim()
    _arenas.trim()      # every idle small-call arena of the process (one in use by an op elsewhere is handed back later and kept)
    _np_pairs_trim()    # and every idle workspace pair of the one-pass scans


# ---- host side of the NumPy door: staging buffers that are reused (no page faults on the copy path) and
# dtype conversions split over a few threads (NumPy releases the GIL inside copyto).  Measured at 2^20 x 22 on the
# GPU box: a fresh 830 MB array costs 68 ms of first-touch page faults single-threaded, `astype(float64)` 95 ms;
# eight threads do cast + first touch in 15 ms.  Small arrays take the plain path.
_PAR_MIN_BYTES = 4 << 20
_PAR_THREADS = max(1, min(int(os.environ.get("PM_HOST_THREADS", "8")), (os.cpu_count() or 1)))
_executor = None
_exec_lock = threading.Lock()


def _pool_exec():
    global _executor
    with _exec_lock:
        if _executor is None:
            from concurrent.futures import ThreadPoolExecutor

            _executor = ThreadPoolExecutor(_PAR_THREADS, thread_name_prefix="pm-cast")
    return _executor


def _parallel_copyto(dst, src):
    """dst[...] = src with dtype conversion; both C-contiguous and of the same shape."""
    if dst.nbytes < _PAR_MIN_BYTES or _PAR_THREADS == 1 or dst.ndim == 0:
        np.copyto(dst, src, casting="unsafe")
        return
    d, s_ = dst.reshape(-1), src.reshape(-1)
    n = d.shape[0]
    step = -(-n // _PAR_THREADS)
    step = (step + 4095) & ~4095  # whole pages of output per thread
    list(_pool_exec().map(lambda i: np.copyto(d[i:i + step], s_[i:i + step], casting="unsafe"), range(0, n, step)))


class _HostStage:
    """Reusable fp32 host buffers (size classes, a few kept): already-touched memory for H2D / D2H staging."""

    def __init__(self, cap_bytes=4 << 30):
        self.free = {}
        self.cached = 0
        self.cap = cap_bytes
        self.lock = threading.Lock()

    def get(self, nbytes):
        c = 1 << (max(int(nbytes), 4096) - 1).bit_length()
        with self.lock:
            lst = self.free.get(c)
            if lst:
                self.cached -= c
                return lst.pop()
        return np.empty(c // 4, dtype=np.float32)

    def put(self, buf):
        with self.lock:
            if self.cached + buf.nbytes <= self.cap:
                self.free.setdefault(buf.nbytes, []).append(buf)
                self.cached += buf.nbytes

    def trim(self):
        with self.lock:
            self.free.clear()
            self.cached = 0


_stage = _HostStage()


# ---- chunked, double-buffered pipeline of the NumPy door ------------------------------------------------------
# A big frame batch that arrives in host memory costs PCIe time both ways (2^20 x 22 fk: 0.38 GB in, 1.1 GB out), a cast
# of the inputs to fp32 and of the outputs to the dtype the reference returns.  Done one after the other that was 47.7 ms
# per 2^20 frames against 0.26 ms of kernel.  Here the batch is cut into frame chunks; chunk k's inputs are staged into
# PAGE-LOCKED memory (threads), copied, computed and copied back on stream k % 2 while the host casts chunk k-1's
# results into the final arrays and the other stream's transfer runs the opposite way on the bus: the call costs about
# what the larger direction of the bus does.
_PIPE_MIN_BYTES = 48 << 20      # below this the plain path wins (fixed costs of a second stream, events, pinned staging)
_PIPE_CHUNK_BYTES = 160 << 20   # payload (in + out, fp32) per chunk (measured 48 / 96 / 192 MB: 55 / 48 / 46 ms at 2^20 x 22)


class _PinnedPool:
    """page-locked staging buffers (hipHostMalloc is expensive: cached by size class, a few kept)"""

    def __init__(self, cap_bytes=2 << 30):
        self.free = {}
        self.cached = 0
        self.cap = cap_bytes
        self.lock = threading.Lock()

    def get(self, nbytes):
        c = 1 << (max(int(nbytes), 1 << 16) - 1).bit_length()
        with self.lock:
            lst = self.free.get(c)
            if lst:
                self.cached -= c
                return lst.pop()
        p = C.c_void_p()
        _lib.call("pm_host_alloc", C.byref(p), c)
        arr = np.ctypeslib.as_array(C.cast(p.value, C.POINTER(C.c_float)), shape=(c // 4,))
        return (p.value, c, arr)

    def put(self, item):
        with self.lock:
            if self.cached + item[1] <= self.cap:
                self.free.setdefault(item[1], []).append(item)
                self.cached += item[1]
                return
        _lib.call("pm_host_free", C.c_void_p(item[0]))

    def trim(self):
        with self.lock:
            items = [i for lst in self.free.values() for i in lst]
            self.free.clear()
            self.cached = 0
        for i in items:
            _lib.call("pm_host_free", C.c_void_p(i[0]))


_pinned = _PinnedPool()
_pipe_streams = {}  # device -> ([stream0, stream1], [event0, event1, event2], lock): created once per device
_pipe_ctx_lock = threading.Lock()


def _pipe_ctx(dev):
    """streams, events and the lock that serialises pipelined calls ON ONE DEVICE (they share its streams / events and
    saturate its bus anyway); calls on different GPUs of one process do not wait for each other"""
    with _pipe_ctx_lock:
        ctx = _pipe_streams.get(dev)
        if ctx is None:
            hs = [C.c_void_p(), C.c_void_p()]
            ev = [C.c_void_p(), C.c_void_p(), C.c_void_p()]  # one per staging slot
            for h in hs:
                _lib.call("pm_stream_create", C.byref(h))
            for e in ev:
                _lib.call("pm_event_create", C.byref(e))
            ctx = _pipe_streams[dev] = (hs, ev, threading.Lock())
    return ctx


_libc = None


def _big_empty(shape, dtype):
    """np.empty + a transparent-huge-page hint: a fresh multi-GB result is ~500 k first-touch page faults with 4 KiB pages
    (the largest host cost of the door); with 2 MiB pages it is a thousand.  No effect where THP is disabled."""
    global _libc
    out = np.empty(shape, dtype=dtype)
    if out.nbytes >= (32 << 20) and os.environ.get("PM_NO_THP") != "1":
        try:
            if _libc is None:
                _libc = C.CDLL(None, use_errno=True)
            a0 = (out.ctypes.data + (1 << 21) - 1) & ~((1 << 21) - 1)
            n = (out.ctypes.data + out.nbytes - a0) & ~((1 << 21) - 1)
            if n > 0:
                _libc.madvise(C.c_void_p(a0), C.c_size_t(n), 14)  # MADV_HUGEPAGE
            _prefault(out)
        except Exception:  # noqa: BLE001  (a hint, never an error)
            pass
    return out


_prefault_pool = None


def _prefault(arr):
    """First-touch the pages of a fresh result array from a few helper threads (MADV_POPULATE_WRITE, Linux >= 5.14; EINVAL on an
    older kernel is ignored) while the pipeline stages and transfers its first chunks: the finisher's casts then write
    into pages that exist instead of faulting them in one by one on the critical path."""
    global _prefault_pool
    if os.environ.get("PM_NO_PREFAULT") == "1":
        return
    if _prefault_pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _prefault_pool = ThreadPoolExecutor(4, thread_name_prefix="pm-prefault")
    page = 4096
    a0 = (arr.ctypes.data + page - 1) & ~(page - 1)
    n = (arr.ctypes.data + arr.nbytes - a0) & ~(page - 1)
    step = max(64 << 20, ((n // 8) + page - 1) & ~(page - 1))
    def populate(keep, addr, nbytes):  # `keep` pins the buffer: a task that runs after the caller dropped the result must not
        _libc.madvise(C.c_void_p(addr), C.c_size_t(nbytes), 23)  # MADV_POPULATE_WRITE      # populate someone else's mapping
        del keep

    for off in range(0, n, step):
        _prefault_pool.submit(populate, arr, a0 + off, min(step, n - off))


def pipelined_frames(F, ins, outs, launch):
    """Run ``launch`` over frame chunks with H2D / kernel / D2H / host casts overlapped (NumPy door).

    ins   list of (array, per_frame): per-frame arrays have F on axis 0 (any float dtype / layout), the others are
          uploaded once as they are (already contiguous fp32 / int arrays)
    outs  list of (trailing shape, result dtype)
    launch(in_ptrs, out_ptrs, n_frames, stream)  -> enqueues the kernel for one chunk
    Returns the list of result arrays [F, *trailing].
    """
    _lib.require_device()
    dev = _current_device()
    ctx = _pipe_ctx(dev)
    with ctx[2]:
        return _pipelined_frames_locked(F, ins, outs, launch, dev, ctx)


def _pipelined_frames_locked(F, ins, outs, launch, dev, ctx):
    streams, events = ctx[0], ctx[1]
    per_in = [int(np.prod(a.shape[1:])) * 4 if pf else 0 for a, pf in ins]
    per_out = [_prod(t) * 4 for t, _ in outs]
    per_frame = max(1, sum(per_in) + sum(per_out))
    chunk = max(1, min(F, _PIPE_CHUNK_BYTES // per_frame))
    # Tapered schedule: the bus is the bottleneck (H2D and D2H share it on this platform: 57 GB/s in all), so what the pipeline
    # adds to the transfer time is its two ends -- staging the first chunk before anything moves, casting the last one after
    # everything has.  Small chunks there, full ones in between (measured at 2^20 x 22, alternating on one box: 40.4 / 35.0 ms
    # uniform, 37.4 / 34.7 ms tapered -- the host side of this path varies by more than the taper gains).
    bounds, f = [], 0
    ramp = [max(1, chunk // 8), max(1, chunk // 4), max(1, chunk // 2)] if F >= 3 * chunk else []
    tail = sum(ramp)
    for c in ramp:
        bounds.append((f, f + c)); f += c
    while F - tail - f > 0:
        c = min(chunk, F - tail - f)
        bounds.append((f, f + c)); f += c
    for c in reversed(ramp):
        c = min(c, F - f)
        if c > 0:
            bounds.append((f, f + c)); f += c
    assert f == F and all(b > a_ for a_, b in bounds)
    nchunks = len(bounds)
    results = [_big_empty((F,) + tuple(t), dt) for t, dt in outs]
    live, consts, slots = [], [], []
    try:
        # constants: one plain upload, visible to both streams (default-stream sync before the pipeline starts)
        const_ptrs = {}
        for i, (a, pf) in enumerate(ins):
            if not pf:
                buf = _DevBuf(max(a.nbytes, 4), dev)
                live.append(buf)
                _lib.call("pm_memcpy_h2d", C.c_void_p(buf.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, None)
                consts.append(a)
                const_ptrs[i] = C.c_void_p(buf.ptr)
        _lib.call("pm_stream_synchronize", None)
        for _ in range(min(3, nchunks)):
            slot = {"din": [], "dout": [], "hin": [], "hout": []}
            for i, (a, pf) in enumerate(ins):
                if pf:
                    slot["din"].append(_DevBuf(chunk * per_in[i], dev))
                    slot["hin"].append(_pinned.get(chunk * per_in[i]))
            for nb in per_out:
                slot["dout"].append(_DevBuf(chunk * nb, dev))
                slot["hout"].append(_pinned.get(chunk * nb))
            slots.append(slot)

        def submit(k):
            sl, st = slots[k % len(slots)], streams[k % 2]
            f0, f1 = bounds[k]
            n = f1 - f0
            in_ptrs, j = [], 0
            for i, (a, pf) in enumerate(ins):
                if not pf:
                    in_ptrs.append(const_ptrs[i])
                    continue
                cnt = n * (per_in[i] // 4)
                stage = sl["hin"][j][2][:cnt].reshape((n,) + a.shape[1:])
                _parallel_copyto(stage, a[f0:f1])          # cast / gather / plain copy into page-locked memory, threaded
                _lib.call("pm_memcpy_h2d", C.c_void_p(sl["din"][j].ptr), C.c_void_p(sl["hin"][j][0]), cnt * 4, st)
                in_ptrs.append(C.c_void_p(sl["din"][j].ptr))
                j += 1
            out_ptrs = [C.c_void_p(b.ptr) for b in sl["dout"]]
            launch(in_ptrs, out_ptrs, n, st)
            for b, h, nb in zip(sl["dout"], sl["hout"], per_out):
                _lib.call("pm_memcpy_d2h", C.c_void_p(h[0]), C.c_void_p(b.ptr), n * nb, st)
            _lib.call("pm_event_record", events[k % len(events)], st)

        def finish(k):
            sl = slots[k % len(slots)]
            f0, f1 = bounds[k]
            n = f1 - f0
            _lib.call("pm_event_synchronize", events[k % len(events)])
            for res, h, (t, _) in zip(results, sl["hout"], outs):
                cnt = n * _prod(t)
                _parallel_copyto(res[f0:f1], h[2][:cnt].reshape((n,) + tuple(t)))

        # The main thread stages and submits; a helper thread waits for each chunk's event and casts its results into the
        # final arrays (both sides fan their copies out over the cast pool, NumPy releases the GIL inside them): staging
        # chunk k+1 and un-staging chunk k-1 overlap each other as well as the transfers.  A slot is reused only after
        # its previous occupant has been finished.
        import queue

        done = [threading.Event() for _ in range(nchunks)]
        todo = queue.Queue()
        err = []

        def finisher():
            while True:
                k = todo.get()
                if k is None:
                    return
                try:
                    if not err:
                        finish(k)
                except Exception as exc:  # noqa: BLE001
                    err.append(exc)
                finally:
                    done[k].set()

        th = threading.Thread(target=finisher, name="pm-door-finish", daemon=True)
        th.start()
        try:
            for k in range(nchunks):
                if k >= len(slots):
                    done[k - len(slots)].wait()  # the slot's previous chunk has left its staging buffers
                if err:
                    break
                submit(k)
                todo.put(k)
        finally:
            todo.put(None)
            th.join()
        if err:
            raise err[0]
    finally:
        for st in streams:
            try:
                _lib.call("pm_stream_synchronize", st)
            except Exception:  # noqa: BLE001  (already failing: release what we hold)
                pass
        for sl in slots:
            for b in sl["din"] + sl["dout"]:
                b.release()
            for h in sl["hin"] + sl["hout"]:
                _pinned.put(h)
        for b in live:
            b.release()
    return results


# ---- small calls (a clip of real length: config 1's 1000 frames x 22 joints is 0.35 MB in, 1.06 MB out) ---------------------------------
# The plain path pays one hipMemcpy from PAGEABLE memory per operand and per result (each a synchronous staging inside the runtime: ~10-20 us
# apiece, bench.py's config1 line: 150-175 us per fk call of which ~100 are those five copies and their waits).  Here every operand of the call
# is cast straight into one PAGE-LOCKED arena, goes over the bus as ONE asynchronous copy right before the launch, the results come back as ONE
# copy into the same arena, and the cast to the result dtype reads from there.  One arena per thread and device, allocated on first use.
_ARENA_BYTES = 4 << 20


class _Arena:
    def __init__(self, dev):
        self.dev = dev
        d = C.c_void_p()
        _lib.call("pm_malloc", C.byref(d), _ARENA_BYTES)
        h = C.c_void_p()
        try:
            _lib.call("pm_host_alloc", C.byref(h), _ARENA_BYTES)
        except Exception:
            _lib.call("pm_free", d)
            raise
        self.dptr, self.hptr = d.value, h.value
        self.host = np.ctypeslib.as_array(C.cast(h.value, C.POINTER(C.c_uint8)), shape=(_ARENA_BYTES,))

    def free(self):
        if self.dptr is not None:
            d, h = self.dptr, self.hptr
            self.dptr = self.hptr = self.host = None
            _lib.call("pm_free", C.c_void_p(d))
            _lib.call("pm_host_free", C.c_void_p(h))

    def __del__(self):  # an arena that never came back (an op that died between begin() and end()): the memory does
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class _ArenaPool:
    """Process-wide, per device: an op checks an arena out in begin() and hands it back in end(), so threads that come and go reuse
    the same few arenas (as many as ops were ever in flight at once) instead of leaving 8 MB of device + page-locked memory each."""

    def __init__(self):
        self.lock = threading.Lock()
        self.idle = {}

    def get(self, dev):
        with self.lock:
            lst = self.idle.get(dev)
            if lst:
                return lst.pop()
        return _Arena(dev)

    def put(self, a):
        if a is None or a.dptr is None:
            return
        with self.lock:
            lst = self.idle.setdefault(a.dev, [])
            if len(lst) < 8:
                lst.append(a)
                return
        a.free()

    def trim(self):
        with self.lock:
            arenas = [a for lst in self.idle.values() for a in lst]
            self.idle.clear()
        for a in arenas:
            a.free()


_arenas = _ArenaPool()


# ---- the one-pass scans' workspace pair (pm_unroll_onepass_f32): two zeroed device blocks per thread, device and stream, alternated call by
# call -- the scan then needs no reset launch in front of it (~3 us of the 7-19 us a clip of real length takes).  A pair is private to the
# thread that made it and to one stream: launches of one thread on one stream run in the order they were made, which is all the alternation needs.
_UNROLL_WS_BYTES = 1 << 20   # covers clips of 2^21 frames x 22 series; bigger calls keep the plain entry points (their reset launch is noise there)
_unroll_pairs = threading.local()   # torch door: pairs are torch tensors keyed by (device, stream), dropped with the thread's table
_np_pairs_lock = threading.Lock()
_np_pairs_idle = {}                 # NumPy door: device -> idle pairs (pm_malloc'd; checked out per call, so short-lived threads do not leak them)


class _UnrollPair:
    def __init__(self, ptr0, ptr1, keep):
        self.ptr = (ptr0, ptr1)
        self.dirty = [0, 0]   # 8-byte words each block holds non-zero
        self.cur = 0          # the clean one
        self.keep = keep      # the torch tensor that owns the blocks, or None: pm_malloc'd, freed by free()

    def take(self):
        o = 1 - self.cur
        return C.c_void_p(self.ptr[self.cur]), C.c_void_p(self.ptr[o]), self.dirty[o]

    def done(self, words):
        self.dirty[1 - self.cur] = 0
        self.dirty[self.cur] = int(words)
        self.cur = 1 - self.cur

    def free(self):
        if self.keep is None and self.ptr is not None:
            ptrs, self.ptr = self.ptr, None
            for p in ptrs:
                _lib.call("pm_free", C.c_void_p(p))
        self.keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def _unroll_pair_wanted(nbytes):
    # (PM_UNROLL_ONEPASS=0: the tuning build's three-pass scan for few series -- it has no reset to save)
    return not (nbytes > _UNROLL_WS_BYTES or os.environ.get("PM_NO_UNROLL_PAIR") == "1" or os.environ.get("PM_UNROLL_ONEPASS") == "0")


def _unroll_pair(key, nbytes, make):
    """torch door: the calling thread's pair for `key`"""
    if not _unroll_pair_wanted(nbytes):
        return None
    table = getattr(_unroll_pairs, "table", None)
    if table is None:
        table = _unroll_pairs.table = {}
    p = table.get(key)
    if p is None:
        if len(table) >= 16:
            table.clear()  # (streams come and go: start over rather than grow)
        p = table[key] = make()
    return p


def _np_pair_get(dev, nbytes, make):
    """NumPy door: check a pair out of the process-wide list (every launch of this door is on the null stream: in order whoever made it)"""
    if not _unroll_pair_wanted(nbytes):
        return None
    with _np_pairs_lock:
        lst = _np_pairs_idle.get(dev)
        if lst:
            return lst.pop()
    return make()


def _unroll_pair_release(key, pair):
    """after a successful scan: the NumPy door's pair goes back to the list (the torch door's stays in its thread's table)"""
    if key is not None and key[0] == "numpy" and pair is not None:
        with _np_pairs_lock:
            lst = _np_pairs_idle.setdefault(key[1], [])
            if len(lst) < 8:
                lst.append(pair)
                return
        pair.free()


def _unroll_pair_drop(key, pair=None):
    """after a failed scan the blocks' state is unknown: the torch door forgets its pair (the tensor goes with it), the NumPy door frees it"""
    table = getattr(_unroll_pairs, "table", None)
    if table is not None:
        table.pop(key, None)
    if pair is not None and key is not None and key[0] == "numpy":
        try:
            _lib.call("pm_stream_synchronize", None)  # nothing of the failed call may still be writing the blocks
        except Exception:  # noqa: BLE001
            pass
        pair.free()


def _np_pairs_trim():
    with _np_pairs_lock:
        pairs = [p for lst in _np_pairs_idle.values() for p in lst]
        _np_pairs_idle.clear()
    for p in pairs:
        p.free()


class NumpyBackend:
    name = "numpy"

    def __init__(self):
        self._live = []
        self._stages = []
        self._ar = None
        self._ar_top = self._ar_in = self._ar_sent = 0
        self._ar_out0 = self._ar_out1 = self._ar_got = 0

    # -- introspection
    @staticmethod
    def shape(x):
        return tuple(np.shape(x))

    @staticmethod
    def result_dtype(*xs):
        dt = np.result_type(*[np.asarray(x).dtype for x in xs])
        return dt if dt.kind == "f" else np.dtype(np.float64)

    @property
    def always64(self):  # what the reference hard-codes for some outputs (opt-out: config.numpy_float64_outputs)
        from . import config

        return np.dtype(np.float64) if config.numpy_float64_outputs else np.dtype(np.float32)

    def stream(self):
        self._ar_flush()  # (every launch takes be.stream() as its last argument: the staged operands go over the bus right before it)
        return None

    @staticmethod
    def wants_pipeline(F, bytes_per_frame):
        """big host-resident frame batches go through pipelined_frames (chunked, double-buffered)"""
        return F >= 2 and F * bytes_per_frame >= _PIPE_MIN_BYTES

    def begin(self, *_):
        _lib.require_device()
        self._dev = _current_device()  # device blocks are pooled per device
        self._live = []
        self._stages = []
        # the small-call arena: [0, _ar_top) is in use, [_ar_sent, _ar_in) holds staged operands not yet sent, [_ar_out0, _ar_out1) results
        if self._ar is not None:  # (a begin() without its end(): hand the old arena back first)
            _arenas.put(self._ar)
        self._ar = _arenas.get(self._dev) if os.environ.get("PM_NO_ARENA") != "1" else None
        self._ar_top = self._ar_in = self._ar_sent = 0
        self._ar_out0 = self._ar_out1 = self._ar_got = 0

    # -- the arena (see _Arena)
    def _ar_take(self, nbytes):
        """offset of `nbytes` of arena (256-byte aligned), or -1"""
        if self._ar is None:
            return -1
        off = (self._ar_top + 255) & ~255
        if off + nbytes > _ARENA_BYTES:
            return -1
        self._ar_top = off + nbytes
        return off

    def _ar_flush(self):
        if self._ar is not None and self._ar_in > self._ar_sent:
            a = self._ar
            _lib.call("pm_memcpy_h2d", C.c_void_p(a.dptr + self._ar_sent), C.c_void_p(a.hptr + self._ar_sent), self._ar_in - self._ar_sent, None)
            self._ar_sent = self._ar_in

    # -- data movement
    def dev_in(self, x, shape=None, dtype=np.float32):
        a = np.asarray(x)
        if shape is not None and a.shape != tuple(shape):
            a = np.broadcast_to(a, shape)
        nb = a.size * np.dtype(dtype).itemsize
        if 0 < nb <= _ARENA_BYTES // 2 and self._ar_out1 == 0:  # (operands come before results: one range each)
            off = self._ar_take(nb)
            if off >= 0:
                np.copyto(self._ar.host[off:off + nb].view(dtype).reshape(a.shape), a, casting="unsafe")  # cast + gather in one pass
                self._ar_in = off + nb
                return C.c_void_p(self._ar.dptr + off)
        keep = None
        if a.dtype == dtype and a.flags.c_contiguous:
            src = a                                    # straight from the caller's memory
        elif dtype == np.float32 and a.size * 4 >= _PAR_MIN_BYTES:
            keep = _stage.get(a.size * 4)              # big and needs a cast / gather: threaded, into reused memory
            src = keep[:a.size].reshape(a.shape)
            _parallel_copyto(src, a if a.flags.c_contiguous else np.ascontiguousarray(a))
        else:
            src = np.ascontiguousarray(a, dtype=dtype)
        buf = _DevBuf(src.nbytes, self._dev)
        _lib.call("pm_memcpy_h2d", C.c_void_p(buf.ptr), src.ctypes.data_as(C.c_void_p), src.nbytes, None)
        self._live.append((buf, src))
        if keep is not None:
            self._stages.append(keep)
        return C.c_void_p(buf.ptr)

    def dev_out(self, shape):
        # (operands staged in the arena go over the bus no later than here: after the first result range dev_in takes the plain path, so a
        # launch that got its stream from somewhere else than be.stream() still finds every operand on the device)
        self._ar_flush()
        nb = _prod(shape) * 4
        if 0 < nb and self._ar_got == 0:
            off = self._ar_take(nb)
            if off >= 0:
                if self._ar_out1 == 0:
                    self._ar_out0 = off
                self._ar_out1 = off + nb
                return C.c_void_p(self._ar.dptr + off), (None, tuple(shape), off)
        buf = _DevBuf(nb, self._dev)
        self._live.append((buf, None))
        return C.c_void_p(buf.ptr), (buf, tuple(shape))

    def result(self, handle, dtype):
        if handle[0] is None:  # in the arena: every result of the call in ONE copy, then the cast from page-locked memory
            _, shape, off = handle
            if self._ar_got == 0:
                self._ar_flush()
                a = self._ar
                _lib.call("pm_memcpy_d2h", C.c_void_p(a.hptr + self._ar_out0), C.c_void_p(a.dptr + self._ar_out0), self._ar_out1 - self._ar_out0, None)
                _lib.call("pm_stream_synchronize", None)
                self._ar_got = 1
            n = _prod(shape)
            return self._ar.host[off:off + 4 * n].view(np.float32).reshape(shape).astype(dtype)  # (always a copy: the arena is reused)
        buf, shape = handle
        n = _prod(shape)
        if n * 4 < _PAR_MIN_BYTES:
            out = np.empty(shape, dtype=np.float32)
            _lib.call("pm_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr), out.nbytes, None)
            _lib.call("pm_stream_synchronize", None)
            return out if np.dtype(dtype) == np.float32 else out.astype(dtype)
        # big: D2H into reused (already touched) memory, then the threaded copy / cast into the fresh result
        st = _stage.get(n * 4)
        try:
            _lib.call("pm_memcpy_d2h", st.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr), n * 4, None)
            _lib.call("pm_stream_synchronize", None)
            out = np.empty(shape, dtype=dtype)
            _parallel_copyto(out, st[:n].reshape(shape))
        finally:
            _stage.put(st)
        return out

    def unroll_pair(self, nbytes):
        """(pair, key) for pm_unroll_onepass_f32 on this backend's stream, or (None, None)"""
        key = ("numpy", self._dev)

        def make():
            ptrs = []
            try:
                for _ in range(2):
                    d = C.c_void_p()
                    _lib.call("pm_malloc", C.byref(d), _UNROLL_WS_BYTES)
                    ptrs.append(d.value)
                    _lib.call("pm_memset", d, 0, _UNROLL_WS_BYTES, None)
            except Exception:
                for p in ptrs:
                    _lib.call("pm_free", C.c_void_p(p))
                raise
            return _UnrollPair(ptrs[0], ptrs[1], None)  # (checked out for this call; _unroll_pair_release puts it back)

        return _np_pair_get(self._dev, nbytes, make), key

    def scratch(self, nbytes):
        self._ar_flush()
        buf = _DevBuf(max(int(nbytes), 4), self._dev)
        self._live.append((buf, None))
        return C.c_void_p(buf.ptr)

    @staticmethod
    def moveaxis(x, src, dst):
        return np.moveaxis(np.asarray(x), src, dst)

    @staticmethod
    def radians(x):
        return np.radians(np.asarray(x))

    def flags_alloc(self, n=3):
        """Zeroed device int32[n] for kernels that report batch-wide predicates."""
        self._ar_flush()
        buf = _DevBuf(4 * n, self._dev)
        _lib.call("pm_memset", C.c_void_p(buf.ptr), 0, 4 * n, None)
        self._live.append((buf, None))
        return C.c_void_p(buf.ptr), (buf, n)

    def flags_read(self, handle):
        buf, n = handle
        out = np.empty(n, dtype=np.int32)
        _lib.call("pm_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(buf.ptr), out.nbytes, None)
        _lib.call("pm_stream_synchronize", None)
        return [int(v) for v in out]

    def end(self):
        try:
            _lib.call("pm_stream_synchronize", None)
        finally:
            for buf, _ in self._live:
                buf.release()
            self._live = []
            for st in self._stages:
                _stage.put(st)
            self._stages = []
            ar, self._ar = self._ar, None
            _arenas.put(ar)  # (results were copied out of it by result(): free for the next op of any thread)

    @staticmethod
    def host_ints(x, slot="parents"):
        return np.ascontiguousarray(np.asarray(x), dtype=np.int32)

    @staticmethod
    def interp_coefficients(sample_times, original_times):
        """ops/time.py:49-54 on the host (two 1-D arrays): -> (idx int32[S], weights [S] in the times' dtype)."""
        st, ot = np.asarray(sample_times), np.asarray(original_times)
        idxs = np.minimum(np.maximum(np.searchsorted(ot, st) - 1, 0), ot.shape[0] - 2)
        weights = (st - ot[idxs]) / (ot[idxs + 1] - ot[idxs])
        return idxs.astype(np.int32), weights

    i32 = np.int32


# ---------------------------------------------------------------------------------------------
# torch front door
# ---------------------------------------------------------------------------------------------
class TorchBackend:
    name = "torch"
    _device_seen = False
    _raw_stream = None

    def __init__(self):
        import torch

        self.torch = torch
        self.dev = None
        self.home = None
        self._keep = []

    @staticmethod
    def shape(x):
        return tuple(x.shape)

    def result_dtype(self, *xs):
        torch = self.torch
        dt = xs[0].dtype
        for x in xs[1:]:
            dt = torch.promote_types(dt, x.dtype)
        return dt if dt.is_floating_point else torch.get_default_dtype()

    @property
    def always64(self):  # the torch twins allocate default-dtype outputs where NumPy hard-codes f64
        return self.torch.get_default_dtype()

    def begin(self, *tensors):
        torch = self.torch
        if not TorchBackend._device_seen:  # (asked once per process: the call is 2 us of a clip-sized launch's 15)
            if not torch.cuda.is_available():
                raise RuntimeError("pymotion_amd (torch path): no HIP device visible and there is no CPU fallback")
            TorchBackend._device_seen = True
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError(
                    "pymotion_amd kernels have no autograd; call under torch.no_grad() or detach() inputs"
                )
        devs = [t.device for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda]
        self.home = tensors[0].device if isinstance(tensors[0], torch.Tensor) else torch.device("cpu")
        cur = torch.cuda.current_device()
        self.dev = devs[0] if devs else torch.device("cuda", cur)
        # (a clip of real length is a few microseconds of kernel: the door's own cost counts -- bench.py's config1 line.  The device
        # guard is only entered when the tensors live on another device than the current one.)
        self._guard = None
        if self.dev.index is not None and self.dev.index != cur:
            self._guard = torch.cuda.device(self.dev)
            self._guard.__enter__()
        self._keep = []

    @staticmethod
    def wants_pipeline(F, bytes_per_frame):
        return False  # device tensors: nothing to overlap

    def stream(self):
        raw = TorchBackend._raw_stream
        if raw is None:  # torch's own accessor of the current stream's handle (what its inductor backend calls): no Stream object per launch
            raw = TorchBackend._raw_stream = getattr(self.torch._C, "_cuda_getCurrentRawStream", False)
        if raw and self.dev.index is not None:
            return C.c_void_p(raw(self.dev.index))
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def dev_in(self, x, shape=None, dtype=None):
        torch = self.torch
        dtype = dtype or torch.float32
        if (isinstance(x, torch.Tensor) and x.dtype is dtype and x.is_cuda and x.device == self.dev and x.is_contiguous()
                and (shape is None or tuple(x.shape) == tuple(shape))):
            return C.c_void_p(x.data_ptr())  # the common case: used where it lies (launched on torch's current stream: a later reuse of the block is ordered behind the kernel)
        t = x if isinstance(x, torch.Tensor) else torch.as_tensor(x)
        t = t.to(device=self.dev, dtype=dtype, non_blocking=True)
        if shape is not None and tuple(t.shape) != tuple(shape):
            t = t.expand(shape)
        t = t.contiguous()
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    def dev_out(self, shape):
        t = self.torch.empty(tuple(shape), dtype=self.torch.float32, device=self.dev)
        self._keep.append(t)
        return C.c_void_p(t.data_ptr()), t

    def result(self, handle, dtype):
        t = handle
        if t.dtype != dtype:
            t = t.to(dtype)
        if self.home.type != "cuda":
            t = t.to(self.home)
        return t

    def unroll_pair(self, nbytes):
        """(pair, key) for pm_unroll_onepass_f32 on the current stream of the tensors' device, or (None, None)"""
        if self.torch.cuda.is_current_stream_capturing():  # (a captured launch is replayed with the SAME blocks: the alternation is the caller's sequence of calls)
            return None, None
        key = ("torch", self.dev.index, self.stream().value or 0)

        def make():
            t = self.torch.zeros((2, _UNROLL_WS_BYTES), dtype=self.torch.uint8, device=self.dev)  # (zeroed on the current stream: in order with the launches)
            return _UnrollPair(t[0].data_ptr(), t[1].data_ptr(), t)

        return _unroll_pair(key, nbytes, make), key

    def scratch(self, nbytes):
        t = self.torch.empty(max(int(nbytes), 4), dtype=self.torch.uint8, device=self.dev)
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def moveaxis(x, src, dst):
        return x.movedim(src, dst)

    @staticmethod
    def radians(x):
        import torch

        return torch.deg2rad(x)

    def flags_alloc(self, n=3):
        t = self.torch.zeros(n, dtype=self.torch.int32, device=self.dev)
        self._keep.append(t)
        return C.c_void_p(t.data_ptr()), t

    def flags_read(self, handle):
        return [int(v) for v in handle.cpu().tolist()]  # the one host sync, where the reference's `if` syncs too

    def end(self):
        self._keep = []
        if self._guard is not None:
            self._guard.__exit__(None, None, None)
            self._guard = None

    # One-entry memos keyed on the tensor OBJECT (weak reference) + torch's in-place version counter.  An address is not
    # an identity: the caching allocator hands a freed block to the next tensor of the same size, and a tensor fresh
    # from `.cuda()` has version 0 -- keying on data_ptr made a new skeleton of the same J hit the previous one's entry.
    _ints_memo = {}

    @staticmethod
    def _memo_get(table, slot, x):
        hit = table.get(slot)
        if hit is not None and hit[0]() is x and hit[1] == x._version:
            return hit[2]
        return None

    @staticmethod
    def _memo_put(table, slot, x, value):
        import weakref

        table[slot] = (weakref.ref(x), x._version, value)

    def host_ints(self, x, slot="parents"):
        torch = self.torch
        if isinstance(x, torch.Tensor):
            if x.is_cuda:
                # J integers living on the device: the copy is a synchronisation, so remember it per tensor object and
                # version (a loop over clips with one `parents` tensor pays once; the reference re-reads every call,
                # skeleton_torch.py:56, which any NEW tensor still gets here)
                arr = self._memo_get(TorchBackend._ints_memo, slot, x)
                if arr is None:
                    arr = np.ascontiguousarray(x.detach().cpu().numpy(), dtype=np.int32)
                    self._memo_put(TorchBackend._ints_memo, slot, x, arr)
                return arr
            x = x.detach().numpy()
        return np.ascontiguousarray(np.asarray(x), dtype=np.int32)

    def interp_coefficients(self, sample_times, original_times):
        """ops/time_torch.py:49-54 with torch ops, wherever the two 1-D tensors live (no host sync)."""
        torch = self.torch
        st, ot = torch.as_tensor(sample_times), torch.as_tensor(original_times)
        ot = ot.to(st.device)
        idxs = torch.clamp(torch.searchsorted(ot, st) - 1, min=0, max=ot.shape[0] - 2)
        weights = (st - ot[idxs]) / (ot[idxs + 1] - ot[idxs])
        return idxs.to(torch.int32), weights

    @property
    def i32(self):
        return self.torch.int32


def numpy_backend():
    return NumpyBackend()


def torch_backend():
    return TorchBackend()
