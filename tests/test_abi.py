"""CPU-only: the C-ABI library loads, exports every symbol include/pmhip.h declares, and its
host-side argument validation works without touching a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from pymotion_amd import _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pmhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree():
    names = declared_symbols()
    assert len(names) >= 40
    h = _lib.lib()
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, f"declared in pmhip.h but not exported: {missing}"
    # and the Python binding table covers the whole header
    unbound = set(names) - set(_lib.SIGNATURES) - {"pm_last_error_string", "pm_last_kernel_name"}
    assert not unbound, f"declared but not bound in _lib.SIGNATURES: {unbound}"
    stale = set(_lib.SIGNATURES) - set(names)
    assert not stale, f"bound but not declared: {stale}"


def test_production_library_never_reads_the_environment():
    """Tuning / ablation aids live in libpmhip_tuning.so (-DPM_TUNING) only: the shipped library must not even
    import getenv, so no PM_* variable can change a result (round-1 ADVICE)."""
    import subprocess

    def undefined(path):
        out = subprocess.run(["nm", "-D", "--undefined-only", path], check=True, capture_output=True, text=True).stdout
        return {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}

    assert "getenv" not in undefined(_lib.VARIANT_PATHS["prod"])
    if os.path.exists(_lib.VARIANT_PATHS["tuning"]):
        assert "getenv" in undefined(_lib.VARIANT_PATHS["tuning"])


def test_every_built_variant_exports_the_whole_header():
    names = declared_symbols()
    for name, path in _lib.VARIANT_PATHS.items():
        if name == "asan":  # only loadable with the ASan runtime preloaded: tests/test_sanitizers.py binds every symbol there
            continue
        if not os.path.exists(path):
            assert name != "prod"
            continue
        h = C.CDLL(path)
        assert not [n for n in names if not hasattr(h, n)], name


def test_version_and_error_string():
    h = _lib.lib()
    assert h.pm_version() == 1
    assert isinstance(h.pm_last_error_string(), bytes)


def _dummy():
    buf = (C.c_float * 64)()
    return C.cast(buf, C.c_void_p), buf


def test_topology_is_validated_on_the_host():
    h = _lib.lib()
    p, keep = _dummy()
    bad = np.array([0, 2, 1], dtype=np.int32)  # parents[1] = 2 >= 1
    rc = h.pm_fk_f32(p, p, p, 0, bad.ctypes.data_as(C.c_void_p), 4, 3, p, p, None)
    assert rc == _lib.PM_ETOPOLOGY
    assert b"topological" in h.pm_last_error_string()
    with pytest.raises(ValueError):
        _lib.check(rc)
    neg = np.array([0, -1, 1], dtype=np.int32)
    assert h.pm_to_root_dq_f32(p, p, neg.ctypes.data_as(C.c_void_p), p, 4, 3, p, None) == _lib.PM_ETOPOLOGY
    assert h.pm_from_root_dq_f32(p, bad.ctypes.data_as(C.c_void_p), 4, 3, p, p, None) == _lib.PM_ETOPOLOGY


def test_bad_arguments_are_rejected_without_a_launch():
    h = _lib.lib()
    p, keep = _dummy()
    ok = np.array([0, 0, 1], dtype=np.int32).ctypes.data_as(C.c_void_p)
    assert h.pm_fk_f32(None, p, p, 0, ok, 4, 3, p, p, None) == _lib.PM_EINVAL
    assert h.pm_fk_f32(p, p, p, 0, ok, -1, 3, p, p, None) == _lib.PM_EINVAL
    assert h.pm_fk_f32(p, p, p, 0, ok, 4, 0, p, p, None) == _lib.PM_EINVAL
    assert h.pm_fk_f32(p, p, p, 0, ok, 4, 100000, p, p, None) == _lib.PM_EINVAL
    assert h.pm_quat_mul_f32(p, None, 4, p, None) == _lib.PM_EINVAL
    assert h.pm_quat_to_matrix_f32(p, -5, p, None) == _lib.PM_EINVAL
    # empty problems are a no-op success, still without a device
    assert h.pm_fk_f32(p, p, p, 0, ok, 0, 3, p, p, None) == _lib.PM_OK
    assert h.pm_quat_mul_f32(p, p, 0, p, None) == _lib.PM_OK


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: on a box without a HIP device the public API must raise."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import pymotion_amd.ops.skeleton as sk
    import pymotion_amd.rotations.quat as quat

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        quat.mul(np.zeros((2, 4)), np.zeros((2, 4)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sk.fk(np.zeros((2, 3, 4)), np.zeros((2, 3)), np.zeros((3, 3)), np.array([0, 0, 1]))
    import torch

    import pymotion_amd.rotations.quat_torch as quat_t

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        quat_t.mul(torch.zeros(2, 4), torch.zeros(2, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pymotion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower() or f == "synthetic.py", f"{f} mentions the oracle"


def test_kernel_names_listed_for_the_oracle_tie_are_the_ones_the_sources_can_report():
    """tests/test_gpu_dispatch.py ties every kernel `pm_last_kernel_name()` can name to the oracle on the production library; its list must be
    the `set_kernel_name` sites of the sources"""
    import glob
    import re

    names = set()
    for f in glob.glob(os.path.join(ROOT, "pymotion_amd", "csrc", "*.hip")):
        src = open(f).read()
        for call in re.finditer(r"set_kernel_name\((.*?)\);", src, re.S):
            names.update(re.findall(r"pm::(\w+_kernel)", call.group(1)))
    import test_gpu_dispatch as d

    assert names == d.ALL_SKELETON_KERNELS, (sorted(names ^ d.ALL_SKELETON_KERNELS))
