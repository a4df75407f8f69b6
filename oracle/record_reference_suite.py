#!/usr/bin/env python3
"""Record every call the REFERENCE's own test suite makes into the hot-path modules (this container only).

    python oracle/record_reference_suite.py            # writes tests/golden/reference_suite.npz
    python oracle/record_reference_suite.py --check    # re-records and compares with the committed fixture (seeded: identical)

How: the public functions of the reference's hot-path modules (pymotion.rotations.{quat,dual_quat,ortho6d}{,_torch},
pymotion.ops.{skeleton,time}{,_torch}) are replaced by recording wrappers BEFORE the reference's test modules are imported; then
the reference's pytest (/root/reference/pymotion/rotations/tests, ops/tests/test_skeleton.py, ops/tests/test_time.py) runs in this
process.  Every call made from a test function's own frame (depth 0) is stored as DATA: module, function, positional arguments,
keyword arguments (copied before the call -- `mirror(mode="symmetry")` writes into its argument, ops/skeleton.py:325) and the
result.  Calls the reference makes internally (fk -> quat.normalize ...) are not stored: the fixture holds what the reference's
authors chose to call and assert on.  `np.random` / `torch` are seeded first, so the suite's unseeded `np.random.rand` inputs are
reproducible.  Nothing of the reference travels: no source text, no bytecode -- arrays and names only.

The test that needs `test.bvh` (ops/tests/test_skeleton.py:215, the file is git-ignored upstream) is deselected.

Fixture layout (np.savez_compressed, no pickle):
    manifest               one JSON string: [{"module", "function", "test", "args": [spec...], "kwargs": {name: spec}, "result": spec}, ...]
    pool_<dtype>           one flat array per dtype; every recorded array is a slice of its pool (thousands of tiny zip members cost more
                           than their payload)
spec = {"k": "array", "dtype", "shape", "off", "tensor": bool} | {"k": "scalar", "v": ...} | {"k": "none"} | {"k": "tuple" | "list", "items": [spec...]}
"""
import argparse
import json
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "reference_suite.npz")
MODULES = [
    "pymotion.rotations.quat", "pymotion.rotations.quat_torch", "pymotion.rotations.dual_quat", "pymotion.rotations.dual_quat_torch",
    "pymotion.rotations.ortho6d", "pymotion.rotations.ortho6d_torch", "pymotion.ops.skeleton", "pymotion.ops.skeleton_torch",
    "pymotion.ops.time", "pymotion.ops.time_torch",
]
TEST_PATHS = ["pymotion/rotations/tests", "pymotion/ops/tests/test_skeleton.py", "pymotion/ops/tests/test_time.py"]
DESELECT = ["pymotion/ops/tests/test_skeleton.py::TestSkeleton::test_from_positions"]  # needs test.bvh (not in the repository)


class Recorder:
    def __init__(self):
        self.records = []
        self.pools = {}     # dtype string -> list of flat arrays
        self.sizes = {}     # dtype string -> elements so far
        self.n_arrays = 0
        self.depth = 0
        self.current_test = ""

    def _spec(self, v, slot):
        import torch

        if v is None:
            return {"k": "none"}
        if isinstance(v, (torch.Tensor, np.ndarray)):
            tensor = isinstance(v, torch.Tensor)
            a = np.array(v.detach().cpu().numpy() if tensor else v, order="C")  # (a copy; np.ascontiguousarray would turn 0-d into 1-d)
            if a.dtype == object:
                a = a.astype(str)
            dt = a.dtype.str.lstrip("<|=")  # "f8", "f4", "i8", "b1", "U1" ...
            off = self.sizes.get(dt, 0)
            self.pools.setdefault(dt, []).append(a.reshape(-1).copy())
            self.sizes[dt] = off + a.size
            self.n_arrays += 1
            return {"k": "array", "dtype": dt, "shape": list(a.shape), "off": off, "tensor": tensor}
        if isinstance(v, (bool, np.bool_)):
            return {"k": "scalar", "v": bool(v)}
        if isinstance(v, (int, np.integer)):
            return {"k": "scalar", "v": int(v)}
        if isinstance(v, (float, np.floating)):
            return {"k": "scalar", "v": float(v)}
        if isinstance(v, str):
            return {"k": "scalar", "v": v}
        if isinstance(v, (tuple, list)):
            return {"k": "tuple" if isinstance(v, tuple) else "list", "items": [self._spec(x, f"{slot}.{i}") for i, x in enumerate(v)]}
        raise TypeError(f"cannot record a {type(v)} in {slot}")

    def wrap(self, modname, name, fn):
        import functools

        rec = self

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            if rec.depth > 0:
                return fn(*args, **kwargs)
            i = len(rec.records)
            entry = {"module": modname.replace("pymotion.", ""), "function": name, "test": rec.current_test,
                     "args": [rec._spec(a, f"r{i}|a{k}") for k, a in enumerate(args)],
                     "kwargs": {k: rec._spec(v, f"r{i}|k_{k}") for k, v in kwargs.items()}}
            rec.records.append(entry)
            rec.depth += 1
            try:
                out = fn(*args, **kwargs)
            except BaseException:
                rec.records.pop()  # a call that raises pins nothing (the suite has none; its arguments stay in the pools, unreferenced)
                raise
            finally:
                rec.depth -= 1
            entry["result"] = rec._spec(out, f"r{i}|out")
            return out

        return wrapper


def record():
    sys.path.insert(0, REF)
    import importlib

    import pytest
    import torch

    rec = Recorder()
    for modname in MODULES:
        mod = importlib.import_module(modname)
        for name, fn in list(vars(mod).items()):
            if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", None) == modname:
                setattr(mod, name, rec.wrap(modname, name, fn))

    class Plugin:
        def pytest_runtest_setup(self, item):
            rec.current_test = item.nodeid.split("pymotion/")[-1]
            seed = sum(map(ord, rec.current_test)) % (2 ** 31)  # one seed per test, independent of the order they run in
            np.random.seed(seed)
            torch.manual_seed(seed)

    cwd = os.getcwd()
    os.chdir(REF)
    try:
        args = ["-q", "-p", "no:cacheprovider", "--no-header"] + TEST_PATHS
        for d in DESELECT:
            args += ["--deselect", d]
        rc = pytest.main(args, plugins=[Plugin()])
    finally:
        os.chdir(cwd)
    if rc != 0:
        raise SystemExit(f"the reference's own suite did not pass here (rc {rc}); nothing written")
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="compare a fresh recording with the committed fixture instead of writing")
    a = ap.parse_args()
    sys.dont_write_bytecode = True  # never leave __pycache__ under /root/reference
    rec = record()
    manifest = json.dumps(rec.records)
    by_fn = {}
    for r in rec.records:
        by_fn[r["module"] + "." + r["function"]] = by_fn.get(r["module"] + "." + r["function"], 0) + 1
    pools = {"pool_" + dt: np.concatenate(parts) for dt, parts in rec.pools.items()}
    print(f"{len(rec.records)} calls recorded from {len({r['test'] for r in rec.records})} reference tests, {rec.n_arrays} arrays, "
          f"{len(by_fn)} distinct functions")
    for k in sorted(by_fn):
        print(f"  {k:55s} {by_fn[k]}")
    if a.check:
        z = np.load(OUT)
        assert str(z["manifest"]) == manifest, "manifest differs"
        for k, v in pools.items():
            assert z[k].dtype == v.dtype and z[k].shape == v.shape and np.array_equal(z[k], v, equal_nan=v.dtype.kind == "f"), k
        assert set(z.files) == set(pools) | {"manifest"}
        print("committed fixture == fresh recording (bit for bit)")
        return
    np.savez_compressed(OUT, manifest=np.asarray(manifest), **pools)
    print(f"{OUT}: {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
