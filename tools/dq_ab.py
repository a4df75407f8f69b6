#!/usr/bin/env python3
"""to_root_dual_quat, SAME process / box / buffers, two builds of the library (prod against the `ab` variant, tools/ab_file.sh):
metre-scale (fp32 step) and centimetre-scale (precise step) data, SMPL-H at 52 joints, random trees elsewhere.
    python tools/dq_ab.py 22,24,28,32,36,40,52 prod,ab"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 60
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
Js = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,22,24,28,32,36,40,52").split(",")]
variants = (sys.argv[2] if len(sys.argv) > 2 else "prod,ab").split(",")
for J in Js:
    par = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if par is None:
        par = syn.random_parents(J, np.random.default_rng(J))
    F = (1 << 20) if J <= 24 else (1 << 18)
    rot = torch.randn((F, J, 4), device="cuda"); rot /= rot.norm(dim=-1, keepdim=True)
    root = torch.randn((F, 3), device="cuda")
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(J), 0.3)).cuda()
    dq = torch.empty((F, J, 8), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    for scale, tag in ((1.0, "metre"), (100.0, "centimetre")):
        o, r = off * scale, root * scale
        row = []
        for rep in range(2):
            for v in variants:
                with _lib.variant(v):
                    ms, _ = pp.timeit(lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(r), pp_, P(o), F, J, P(dq), None))
                    name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
                row.append(f"{v}: {ms * 1e3:7.1f} us {F * (48 * J + 12) / ms / 1e6 / 80:5.1f}%")
        print(f"J={J:3d} {tag:10s} {name[:32]:32s} | " + " | ".join(row), flush=True)
    del rot, dq
