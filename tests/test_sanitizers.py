"""CPU-only: the host-side code under AddressSanitizer + UndefinedBehaviorSanitizer ("-fsanitize=address,
undefined host build of the CPU restatement + ctypes layer").

  * oracle/_build/liboracle_asan.so (gcc): the whole oracle test-suite against the reference goldens, in a subprocess with
    gcc's libasan preloaded;
  * pymotion_amd/libpmhip_asan.so (hipcc, -Xarch_host -fsanitize=...): everything the library does on the host before a
    kernel launch -- argument validation, topology packing, the chain / program schedulers of to_root_dual_quat and
    from_root_positions -- driven through the C ABI on a box WITHOUT a GPU (the launch itself then fails with PM_EHIP,
    which is the expected end of every call here), in a subprocess with clang's ASan runtime preloaded.
A sanitizer report aborts the subprocess (halt_on_error) and fails the test."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ASAN_OPTS = "detect_leaks=0:halt_on_error=1:abort_on_error=1"
UBSAN_OPTS = "halt_on_error=1:print_stacktrace=1"


def _run(code_or_args, preload, extra_env):
    env = dict(os.environ)
    env.update({"LD_PRELOAD": preload, "ASAN_OPTIONS": ASAN_OPTS, "UBSAN_OPTIONS": UBSAN_OPTS, "PYTHONPATH": ROOT})
    env.update(extra_env)
    r = subprocess.run([sys.executable] + code_or_args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_oracle_suite_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    out = _run(["-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle.py", "tests/test_bvh_unroll.py", "tests/test_ik.py",
                "tests/test_time.py", "-m", "not gpu"], libasan, {"PM_ORACLE_ASAN": "1"})
    assert " passed" in out and "failed" not in out, out[-800:]


_HOST_DRIVER = r"""
import ctypes as C, numpy as np, sys
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
assert _lib.VARIANT_PATHS["asan"].endswith("libpmhip_asan.so")
h = _lib.lib()
buf = (C.c_float * 4096)()
p = C.cast(buf, C.c_void_p)
rng = np.random.default_rng(0)
calls = 0
def trees(J):
    yield np.maximum(np.arange(J) - 1, 0).astype(np.int32)          # chain
    yield np.zeros(J, np.int32)                                      # star
    yield ((np.arange(J) - 1) // 2).clip(0).astype(np.int32)         # heap
    c = np.maximum(np.arange(J) - 1, 0).astype(np.int32)             # a chain with two branch points (deep.hip's register slots)
    c[J // 2] = 0; c[3 * J // 4] = J // 4
    yield c
    n = np.maximum(np.arange(J) - 1, 0).astype(np.int32)             # branch points nested deeper than the slots go: the fallback
    for k in range(1, min(7, J // 8)):
        n[J - k] = k
    yield n
    for _ in range(3):
        yield syn.random_parents(J, rng)
    for w in (2, 3, 6):                                              # parents a few joints back: ik_order_plan's window, queue and register sets
        yield np.concatenate([[0], [rng.integers(max(0, j - w), j) for j in range(1, J)]]).astype(np.int32)
    if J == 52:
        yield syn.PARENTS_52                                         # SMPL-H's level-order table
for J in (1, 2, 3, 22, 27, 28, 52, 64, 65, 128, 250, 251, 254, 255, 512):
    for par in trees(J):
        pp = par.ctypes.data_as(C.c_void_p)
        for F in (1, 7, 4099):
            # no GPU here: every call runs its host side (validation, packing, scheduling, dispatch) and ends in PM_EHIP
            rcs = [h.pm_fk_f32(p, p, p, 0, pp, F, J, p, p, None),
                   h.pm_fk_f32(p, p, p, 1, pp, F, J, p, p, None),
                   h.pm_fk_from_ortho6d_f32(p, p, p, 0, pp, F, J, C.c_float(0.0), p, p, p, None),
                   h.pm_to_root_dq_f32(p, p, pp, p, F, J, p, None),
                   h.pm_from_root_dq_f32(p, pp, F, J, p, p, None),
                   h.pm_from_global_rotations_f32(p, pp, F, J, p, None),
                   h.pm_from_root_positions_f32(p, pp, p, F, J, p, None),
                   h.pm_mirror_rotations_f32(p, pp, None, 0, F, J, p, None)]
            calls += len(rcs)
            assert all(rc in (_lib.PM_EHIP, _lib.PM_EUNSUPPORTED) for rc in rcs), (J, F, rcs)
bad = np.array([0, 2, 1], np.int32)
assert h.pm_fk_f32(p, p, p, 0, bad.ctypes.data_as(C.c_void_p), 4, 3, p, p, None) == _lib.PM_ETOPOLOGY
assert h.pm_fk_f32(None, p, p, 0, bad.ctypes.data_as(C.c_void_p), 4, 3, p, p, None) == _lib.PM_EINVAL
assert h.pm_quat_mul_f32(p, p, 10, p, None) in (_lib.PM_EHIP,)
assert h.pm_quat_unroll_workspace_bytes(1000, 22) > 0
for T, S in ((1000, 22), (5, 64), (70000, 65), (3, 100000)):  # one-pass and three-pass dispatch, host side
    assert h.pm_quat_unroll_workspace_bytes(T, S) >= 64
    assert h.pm_quat_unroll_f32(p, T, S, p, p, None) == _lib.PM_EHIP
    assert h.pm_dq_unroll_f32(p, T, S, p, p, None) == _lib.PM_EHIP
    for B in (1, 3, 5000):  # batches of clips: tile size per clip, single-tile clips, wide clips one after the other
        assert h.pm_quat_unroll_batched_workspace_bytes(B, T, S) >= h.pm_quat_unroll_workspace_bytes(T, S)
        assert h.pm_quat_unroll_batched_f32(p, B, T, S, p, p, None) == _lib.PM_EHIP
        assert h.pm_dq_unroll_batched_f32(p, B, T, S, p, p, None) == _lib.PM_EHIP
n = C.c_int64(-1)
for kind, B, T, S in ((0, 1, 1000, 22), (1, 3, 70000, 5), (2, 1, 5000, 64), (0, 64, 30, 22)):  # the scans on a workspace pair: host side
    order = np.zeros((64, 3), np.uint8)
    assert h.pm_unroll_onepass_f32(kind, p, order.ctypes.data_as(C.c_void_p), B, T, S, p, p, C.byref(n), C.c_void_p(p.value + 4096), 100, None) == _lib.PM_EHIP
    assert n.value >= 0
assert h.pm_unroll_onepass_f32(0, p, None, 1, 10, 65, p, p, C.byref(n), C.c_void_p(p.value + 4096), 0, None) == _lib.PM_EUNSUPPORTED
jobs = (C.c_uint32 * (50 * 16))()
for J in (1, 2, 17, 52, 129, 512):  # the wide walk's step list (width 16; fk's dispatch above ran width 4 too)
    for par in trees(J):
        st = h.pm_fk_wide_plan_debug(par.ctypes.data_as(C.c_void_p), J, jobs)
        assert st >= 0 or st == _lib.PM_EUNSUPPORTED
        calls += 1
        for op in (0, 1):  # the step words of round 6's step-list kernels (to_root_dual_quat, mirror) at every width
            for fpw in (1, 2, 4, 8):
                words = (C.c_uint32 * (16 * 56))()
                st = h.pm_step_list_plan_debug(par.ctypes.data_as(C.c_void_p), J, op, fpw, words)
                assert st >= 0 or st == _lib.PM_EUNSUPPORTED
                calls += 1
print("host paths exercised:", calls)
"""


def test_library_host_side_under_asan_ubsan():
    from pymotion_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: the launches would go through (the sanitizer run is for the host side)")
    # always through make: a library left over from an older source tree must not be what gets tested
    subprocess.run(["make", "-C", os.path.join(ROOT, "pymotion_amd", "csrc"), "-j8", "asan"], check=True, stdout=subprocess.DEVNULL)
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    assert rts, "clang's ASan runtime not found"
    out = _run(["-c", _HOST_DRIVER], rts[0], {"PMHIP_VARIANT": "asan"})
    assert "host paths exercised" in out
