#!/usr/bin/env python3
"""What each arithmetic level of fk costs and buys (tuning build of the library, PM_FK_PREC):
   0 fp32 as in round 1 | 1 residual-scaled phase A | 6 float64 phase A + fixed-point translation chain |
   17 = what production runs: per tile, 1 for human-scale metre data, 6 when bones / roots are big.
For every level: sustained time at 2^20 x 22 and 2^18 x 52, and the max error against the float64 C oracle on a
2^14-frame sample at metre scale (offsets 0.3, root 2) and centimetre scale (offsets 30, root 200), in absolute terms
and in ulps of the batch's largest |coordinate|."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from oracle import c_oracle as co
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn


def p(t):
    return C.c_void_p(t.data_ptr())


def sustained(fn, n=100, warm=150):
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))
    for _ in range(warm):
        fn()
    _lib.call("pm_event_record", ev[0], None)
    for _ in range(n):
        fn()
    _lib.call("pm_event_record", ev[1], None)
    ms = C.c_float()
    _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
    return ms.value / n


def main():
    dev = torch.device("cuda:0")
    levels = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,6,17".split(","))]
    with _lib.variant("tuning"):
        for J, parents, F in ((22, syn.PARENTS_22, 1 << 20), (52, syn.PARENTS_52, 1 << 18)):
            g = torch.Generator(device=dev)
            g.manual_seed(J)
            rot = torch.randn((F, J, 4), generator=g, device=dev)
            pos = torch.empty((F, J, 3), device=dev)
            rm = torch.empty((F, J, 3, 3), device=dev)
            pp = parents.ctypes.data_as(C.c_void_p)
            n = 1 << 14
            rot_h = rot[:n].cpu().numpy().astype(np.float64)
            for lvl in levels:
                os.environ["PM_FK_PREC"] = str(lvl)
                line = f"J={J:3d} PREC={lvl}:"
                for name, osc, rsc in (("m", 0.3 if J == 22 else 0.15, 2.0), ("cm", 30.0, 200.0)):
                    rng = np.random.default_rng(7)
                    off_np = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
                    off_np[0] = 0
                    root = (torch.rand((F, 3), generator=g, device=dev) * 2 - 1) * rsc
                    off = torch.from_numpy(off_np).to(dev)
                    fn = lambda: _lib.call("pm_fk_f32", p(rot), p(root), p(off), 0, pp, F, J, p(pos), p(rm), None)  # noqa: E731
                    fn()
                    torch.cuda.synchronize()
                    p_o, r_o = co.fk(rot_h, root[:n].cpu().numpy().astype(np.float64), off_np.astype(np.float64), parents)
                    ep = np.abs(pos[:n].cpu().numpy() - p_o).max()
                    er = np.abs(rm[:n].cpu().numpy() - r_o).max()
                    ulp = 2.0 ** (np.floor(np.log2(np.abs(p_o).max())) - 23)
                    ms = sustained(fn)
                    line += f" | {name}: {ms * 1e3:6.1f} us ({F * (64 * J + 12) / ms / 1e6 / 8000 * 100:4.1f}%) pos {ep:.2e} ({ep / ulp:4.2f} ulp) rot {er:.2e}"
                print(line, flush=True)


if __name__ == "__main__":
    main()
