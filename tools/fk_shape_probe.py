#!/usr/bin/env python3
"""fk at 2^20 x 22 under every tile shape the tuning build can force (frames per wave, tiles per workgroup), next to the copy
kernel of the same shapes: how much of the distance to the chip's mixed-stream rate is bytes in flight (waves per CU)?
    PMHIP_VARIANT=tuning python tools/fk_shape_probe.py
"""
import ctypes as C
import os
import sys

os.environ.setdefault("PMHIP_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from store_probe import sustained, p


def main():
    dev = torch.device("cuda:0")
    J = int(os.environ.get("PROBE_J", "22"))
    F = (1 << 20) if J <= 24 else (1 << 18)
    parents = syn.PARENTS_22 if J == 22 else (syn.PARENTS_52 if J == 52 else syn.random_parents(J, np.random.default_rng(J)))
    rot, root, off, parents = syn.fk_workload(F, parents=parents, seed=0)
    rot_d, root_d, off_d = (torch.from_numpy(x).to(dev) for x in (rot, root, off))
    pos = torch.empty((F, J, 3), device=dev)
    rm = torch.empty((F, J, 3, 3), device=dev)
    big = torch.empty(F * J * 12, device=dev)
    pp = parents.astype(np.int32).ctypes.data_as(C.c_void_p)
    nbytes = F * (64 * J + 12)

    def line(label, ms, nb=nbytes):
        print(f"{label:64s} {ms * 1e3:8.1f} us  {nb / ms / 1e6:7.1f} GB/s  {nb / ms / 1e6 / 80:5.1f} %   [{_lib.last_kernel_name()[:70]}]", flush=True)

    def with_env(env, fn):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            return fn()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    fk = lambda: _lib.call("pm_fk_f32", p(rot_d), p(root_d), p(off_d), 0, pp, F, J, p(pos), p(rm), None)  # noqa: E731
    ceil = lambda: _lib.call("pm_stream_ceiling_f32", p(rot_d), p(big), F, 4 * J, 12 * J, None)  # noqa: E731
    line("fk, production dispatch", sustained(fk))
    for fpw in (20, 16, 12, 8, 4):
        for nt in ((0,) if fpw != 4 else (0, 1, 2, 4)):
            env = {"PM_FK_FPW": fpw, "PM_FK_NT": nt}
            try:
                ms = with_env(env, lambda: sustained(fk))
                line(f"fk, {fpw} frames per wave" + (f", {nt} tiles per workgroup (pipelined)" if nt else ""), ms)
            except Exception as e:  # a shape the build does not carry
                print(f"fk, {fpw} frames per wave, nt {nt}: {e}")
    for fpw in (16, 20):
        ms = with_env({"PM_FK_FPW": fpw, "PM_FK_ABLATE": 4}, lambda: sustained(fk))
        line(f"fk, {fpw} frames per wave, LINEAR tile order (neighbouring tiles on different XCDs)", ms)
    for fpw in (16,):
        for nt in (2,):
            ms = with_env({"PM_FK_FPW": fpw, "PM_FK_PIPE3": nt}, lambda: sustained(fk))
            line(f"fk, {fpw} frames per wave, {nt} tiles per workgroup, next tile prefetched", ms)
    line("fk, production dispatch (again)", sustained(fk))
    for abl in (2,):
        for fpw in (16, 8):
            ms = with_env({"PM_FK_FPW": fpw, "PM_FK_ABLATE": abl}, lambda: sustained(fk))
            line(f"fk, {fpw} frames per wave, tree walk ablated", ms)
    for fpw in (32, 20, 16, 12, 8, 4, 2):
        ms = with_env({"PM_CEIL_FPW": fpw}, lambda: sustained(ceil))
        line(f"copy kernel, {fpw} frames per wave ({fpw * J * 48 / 1024:.1f} KB of LDS)", ms, F * 64 * J)


if __name__ == "__main__":
    main()
