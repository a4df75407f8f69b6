#!/usr/bin/env python3
"""fused ortho6d -> fk (SMPL-H, 2^18 frames): bench.py's allocation sequence with dummy allocations of varying size between the arrays --
does the placement the allocator happens to give decide between the 0.65 and the 0.70 the same kernel reads?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F, J = 1 << 18, 52
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32); pp4 = par.ctypes.data_as(C.c_void_p)
MB = 1 << 20
# what bench.py holds when its secondary section starts: the headline arrays
hold = [torch.randn((1 << 20, 22, 4), device="cuda"), torch.empty((1 << 20, 22, 3), device="cuda"), torch.empty((1 << 20, 22, 3, 3), device="cuda")]
for pads in ((0, 0, 0), (2, 0, 0), (0, 2, 0), (0, 0, 2), (64, 0, 0), (0, 64, 0), (0, 0, 64), (256, 0, 0), (0, 256, 0), (0, 0, 256), (100, 200, 300), (0, 0, 0)):
    torch.cuda.empty_cache()
    d0 = torch.empty(pads[0] * MB, dtype=torch.uint8, device="cuda") if pads[0] else None
    x = torch.randn((F, J, 3, 2), device="cuda")
    root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(4), 0.15)).cuda()
    d1 = torch.empty(pads[1] * MB, dtype=torch.uint8, device="cuda") if pads[1] else None
    pos = torch.empty((F, J, 3), device="cuda")
    d2 = torch.empty(pads[2] * MB, dtype=torch.uint8, device="cuda") if pads[2] else None
    rm = torch.empty((F, J, 3, 3), device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x), P(root), P(off), 0, pp4, F, J, C.c_float(0.0), P(pos), P(rm), None, None))
    print(f"pads {pads}: x {x.data_ptr():#x} pos {pos.data_ptr():#x} rm {rm.data_ptr():#x}  (pos - x {(pos.data_ptr() - x.data_ptr()) / MB:8.1f} MB, rm - x {(rm.data_ptr() - x.data_ptr()) / MB:8.1f} MB): "
          f"{ms * 1e3:7.1f} us {F * (72 * J + 12) / ms / 1e6 / 80:5.1f}%", flush=True)
    del x, pos, rm, d0, d1, d2, root, off
