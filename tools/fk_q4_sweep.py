#!/usr/bin/env python3
"""fk on mid-size skeletons (24 ... 96 joints): the production dispatch against the tile shapes the tuning build can force
(PM_FK_FPW = 16 / 12 / 8: a quad per frame, L shared through DPP; 4: twelve lanes per frame, pipelined tiles), quaternion and
ortho6d source (with / without the quaternion output), chain-like and SMPL-H skeletons.  2^18 frames (2^19 up to 40 joints).
    PMHIP_VARIANT=tuning python tools/fk_q4_sweep.py [J,J,...]"""
import ctypes as C
import os
import sys

os.environ.setdefault("PMHIP_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
from store_probe import sustained

P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


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


def main():
    Js = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "24,28,32,36,40,48,52,56,64,72".split(","))]
    shapes = [int(x) for x in os.environ.get("SWEEP_SHAPES", "0,16,12,8,4").split(",")]
    for J in Js:
        F = (1 << 19) if J <= 40 else (1 << 18)
        if J == 52:
            par = syn.PARENTS_52.astype(np.int32)
        else:
            par = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
            par[J // 2] = 0
            par[3 * J // 4] = J // 4
        rot = torch.randn((F, J, 4), device="cuda")
        o6 = torch.randn((F, J, 3, 2), device="cuda")
        root = torch.randn((F, 3), device="cuda")
        off = torch.randn((J, 3), device="cuda") * 0.15
        pos = torch.empty((F, J, 3), device="cuda")
        rm = torch.empty((F, J, 3, 3), device="cuda")
        qo = torch.empty((F, J, 4), device="cuda")
        pp_ = par.ctypes.data_as(C.c_void_p)
        ops = [
            ("fk", lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None), 64 * J + 12),
            ("o6d->fk", lambda: _lib.call("pm_fk_from_ortho6d_f32", P(o6), P(root), P(off), 0, pp_, F, J, C.c_float(0.0), P(pos), P(rm), None, None), 72 * J + 12),
            ("o6d->fk+q", lambda: _lib.call("pm_fk_from_ortho6d_f32", P(o6), P(root), P(off), 0, pp_, F, J, C.c_float(0.0), P(pos), P(rm), P(qo), None), 88 * J + 12),
        ]
        for name, fn, nb in ops:
            line = f"J={J:3d} {name:10s}"
            for fpw in shapes:
                env = {"PM_FK_FPW": fpw} if fpw else {}
                if fpw >= 100:  # 108 / 208: eight frames per wave, pipelined, 1 / 2 tiles per workgroup (PM_FK_PIPE3)
                    env = {"PM_FK_FPW": fpw % 100, "PM_FK_PIPE3": fpw // 100}
                try:
                    ms = with_env(env, lambda: sustained(fn, n=40, warm=60))
                    k = _lib.last_kernel_name().split("pm::")[-1].split("(")[0]
                    line += f" | {'prod' if not fpw else fpw:>4}: {ms * 1e3:7.1f} us {F * nb / ms / 1e6 / 80:5.1f}% {k[:34]:34s}"
                except Exception as e:
                    line += f" | {fpw:>4}: {str(e)[:40]}"
            print(line, flush=True)
        del rot, o6, pos, rm, qo


if __name__ == "__main__":
    main()
