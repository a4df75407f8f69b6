#!/usr/bin/env python3
"""fk's sixteen-frame quad walk: which frames share a half-wave (q4_frame_map, fk.hip) -- same-process A/B on the tuning build.
PM_FK_FMAP=0 is the block split of rounds 4-5 (frames 0..7 | 8..15), unset = the host's pick.  Alternates the two, ROUNDS times each
per joint count, sustained launches; prints both times, the copy kernel of the shape and bit equality of the outputs.

    PMHIP_VARIANT=tuning python tools/fmap_ab.py [J ...]      (default 22; 2^20 frames at J = 22, 2^19 elsewhere)"""
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
ROUNDS = int(os.environ.get("FMAP_ROUNDS", "4"))


def sustained(fn, n):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    Js = [int(x) for x in sys.argv[1:]] or [22]
    for J in Js:
        F = 1 << 20 if J == 22 else 1 << 19
        par = np.ascontiguousarray(syn.PARENTS_22 if J == 22 else np.array([0] + [max(0, j - 1 - (j % 3 == 0) * 2) for j in range(1, J)]), dtype=np.int32)
        src = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        big = torch.empty((F, J, 12), device="cuda")
        fk = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
        cp = lambda: _lib.call("pm_stream_ceiling_f32", P(src), P(big), F, 4 * J, 12 * J, None)  # noqa: E731
        n = max(200, int(1.0e6 / (F * (64 * J + 12) / 5.5e6)))  # ~1 s of launches
        os.environ["PM_FK_FMAP"] = "0"
        fk(); torch.cuda.synchronize()
        kern = _lib.lib().pm_last_kernel_name().decode()
        ref = (pos.clone(), rm.clone())
        os.environ.pop("PM_FK_FMAP")
        fk(); torch.cuda.synchronize()
        same = bool(torch.equal(pos, ref[0]) and torch.equal(rm, ref[1]))
        ta, tb = [], []
        for _ in range(ROUNDS):
            os.environ["PM_FK_FMAP"] = "0"; ta.append(sustained(fk, n))
            os.environ.pop("PM_FK_FMAP"); tb.append(sustained(fk, n))
        tc = sustained(cp, n)
        b = F * (64 * J + 12)
        print(f"J={J:3d} F={F}: block split {min(ta):7.1f} us ({b / min(ta) / 8e6 * 100:5.1f} %)   host's split {min(tb):7.1f} us ({b / min(tb) / 8e6 * 100:5.1f} %)   "
              f"copy kernel {tc:7.1f} us   all rounds {['%.1f' % x for x in ta]} / {['%.1f' % x for x in tb]}   bit-equal {same}   {kern[:60]}", flush=True)
        del src, pos, rm, big


if __name__ == "__main__":
    main()
