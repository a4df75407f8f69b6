#!/usr/bin/env python3
"""The timing level of an fk launch belongs to the ALLOCATION (profiles/r06_levels.txt): the same kernel reads 150 us on one set of arrays and 166 us on
the next, visit after visit.  Does the ORDER in which the workgroups sweep the arrays decide it?  Tuning build: three sets of arrays per workload
(J = 52 at 2^18 frames, J = 22 at 2^20), each timed under
    default            every XCD one contiguous eighth of the tiles (xcd_tile)
    PM_FK_ABLATE=4     linear order (neighbouring tiles on different XCDs)
    PM_FK_XCHUNK=n     the XCDs' ranges cut into chunks of n tiles / tile groups that take turns (xcd_tile_chunked)
and the copy kernel of the shape on the same arrays.

    python tools/alloc_level_probe.py [chunks ...]"""
import ctypes as C
import os
import sys

os.environ.setdefault("PMHIP_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
SEC = float(os.environ.get("ALP_SECONDS", "0.35"))


def sustained(fn, seconds):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): fn()
    e1.record(); torch.cuda.synchronize()
    n = max(40, int(seconds * 1e3 / (e0.elapsed_time(e1) / 40)))
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def workload(J, F, par):
    par = np.ascontiguousarray(par, dtype=np.int32)
    src = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    big = torch.empty((F, J, 12), device="cuda")
    fk = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
    cp = lambda: _lib.call("pm_stream_ceiling_f32", P(src), P(big), F, 4 * J, 12 * J, None)  # noqa: E731
    return fk, cp, (src, root, off, pos, rm, big, par)


def main():
    chunks = [int(x) for x in sys.argv[1:]] or [1, 8, 64, 448, 2048]
    for J, F, par in ((52, 1 << 18, syn.PARENTS_52), (22, 1 << 20, syn.PARENTS_22)):
        sets = [workload(J, F, par) for _ in range(3)]
        modes = [("default", {}), ("linear", {"PM_FK_ABLATE": "4"})] + [(f"chunk {c}", {"PM_FK_XCHUNK": str(c)}) for c in chunks]
        if J == 52:
            modes += [("plain stores", {"PM_FK_ABLATE": "256"}), ("plain loads", {"PM_FK_ABLATE": "512"}), ("plain both", {"PM_FK_ABLATE": "768"})]
        print(f"## J = {J}, F = {F}: us a launch per allocation set (set 0 / 1 / 2), % of the HBM spec of the slowest", flush=True)
        ref = None
        for name, env in modes:
            for k in ("PM_FK_ABLATE", "PM_FK_XCHUNK"): os.environ.pop(k, None)
            os.environ.update(env)
            ts = [sustained(fk, SEC) for fk, _, _ in sets]
            fk0, _, keep = sets[0]
            fk0(); torch.cuda.synchronize()
            out = (keep[3].clone(), keep[4].clone())
            same = "" if ref is None else f"  bit-equal to default {bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))}"
            if ref is None: ref = out
            print(f"   {name:12s} " + "  ".join(f"{t:7.1f}" for t in ts) + f"   {F * (64 * J + 12) / max(ts) / 8e6 * 100:5.1f} %{same}", flush=True)
        for k in ("PM_FK_ABLATE", "PM_FK_XCHUNK"): os.environ.pop(k, None)
        print("   copy kernel  " + "  ".join(f"{sustained(cp, SEC):7.1f}" for _, cp, _ in sets), flush=True)
        print("   default again" + "  ".join(f"{sustained(fk, SEC):7.1f}" for fk, _, _ in sets), flush=True)
        del sets


if __name__ == "__main__":
    main()
