import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """tests/golden/<name>.npz written by oracle/make_golden.py: keys '<case>|<kind>|<array>'."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.cases = {}
        for k in self.z.files:
            case, kind, arr = k.split("|")
            self.cases.setdefault(case, {}).setdefault(kind, {})[arr] = k

    def names(self, prefix=""):
        return sorted(c for c in self.cases if c.startswith(prefix))

    def get(self, case, kind):
        return {a: self.z[k] for a, k in self.cases[case].get(kind, {}).items()}


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope="session")
def g_elementwise():
    return golden("elementwise.npz")


@pytest.fixture(scope="session")
def g_trig():
    return golden("trig.npz")


@pytest.fixture(scope="session")
def g_skeleton():
    return golden("skeleton.npz")


def up64(d):
    return {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in d.items()}


def assert_close(a, b, atol, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all(), f"{what}: NaN pattern differs"
    if a.size:
        err = np.abs(np.where(nan_a, 0, a) - np.where(nan_b, 0, b)).max()
        assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"


@pytest.fixture
def lane_per_frame_at_test_sizes(monkeypatch):
    """The long-skeleton kernels (one lane per frame: deep.hip, mirror_deep_kernel, from_root_positions_order_kernel, and fk's streamed walk)
    only take calls with enough joint-frames to fill the chip (common.hpp: lane_per_frame_pays -- a clip of real length is faster on the tile
    kernels).  Parity tests that want THOSE kernels at sizes the oracle finishes in seconds run on the tuning build with the threshold at 0:
    same kernels, same dispatch otherwise -- except that the step-list kernels of round 6 (to_root_dq_wide_kernel, dqwide.hip; mirror_wide_kernel, mirror.hip),
    which the dispatch asks before them, stand aside (PM_DQ_WIDE=0, PM_MIRROR_WIDE=0): their own suites are tests/test_gpu_dqwide.py and
    tests/test_gpu_mirror_wide.py.  The production library's choice on either
    side of the threshold: tests/test_gpu_dispatch.py."""
    from pymotion_amd import _lib

    monkeypatch.setenv("PM_LPF_MIN_JOINT_FRAMES", "0")
    monkeypatch.setenv("PM_DQ_WIDE", "0")
    monkeypatch.setenv("PM_MIRROR_WIDE", "0")
    with _lib.variant("tuning"):
        yield


class ReferenceSuite:
    """tests/golden/reference_suite.npz (oracle/record_reference_suite.py): every call the reference's own test suite makes into the
    hot-path modules -- module, function, arguments and the result the reference returned -- as data."""

    def __init__(self):
        import json

        z = np.load(os.path.join(GOLDEN, "reference_suite.npz"))
        self.pools = {k[5:]: z[k] for k in z.files if k.startswith("pool_")}
        self.records = json.loads(str(z["manifest"]))

    def value(self, spec, as_tensor=True, device=None):
        """materialise a recorded value; arrays recorded from torch tensors come back as tensors (on `device`) unless as_tensor is False"""
        k = spec["k"]
        if k == "none":
            return None
        if k == "scalar":
            return spec["v"]
        if k in ("tuple", "list"):
            items = [self.value(s, as_tensor, device) for s in spec["items"]]
            return tuple(items) if k == "tuple" else items
        n = int(np.prod(spec["shape"], dtype=np.int64))
        a = self.pools[spec["dtype"]][spec["off"]:spec["off"] + n].reshape(spec["shape"]).copy()
        if spec["tensor"] and as_tensor:
            import torch

            t = torch.from_numpy(a)
            return t.to(device) if device is not None else t
        return a

    def by_function(self):
        out = {}
        for i, r in enumerate(self.records):
            out.setdefault(r["module"] + "." + r["function"], []).append(i)
        return out


def reference_suite():
    if "reference_suite" not in _cache:
        _cache["reference_suite"] = ReferenceSuite()
    return _cache["reference_suite"]
