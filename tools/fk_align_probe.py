#!/usr/bin/env python3
"""fk (SMPL-H, 2^18 frames; the 22-joint body, 2^20) with its arrays carved out of ONE buffer at chosen byte offsets: does the kernel's time depend on
where the allocator put them?  (bench.py's arrays come from torch's caching allocator -- blocks split off bigger cached ones, 512-byte aligned --
and perf_probe's from fresh 2 MB-aligned segments; the two read the same kernel 8 % apart on one box.)"""
import ctypes as C, os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 30
big = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
base = (big.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
for J, F, par in ((52, 1 << 18, syn.PARENTS_52), (22, 1 << 20, syn.PARENTS_22)):
    par = np.ascontiguousarray(par, dtype=np.int32)
    nb = {"src": F * J * 16, "pos": F * J * 12, "rm": F * J * 36}
    def carve(off, nbytes):
        a = base + off
        return torch.frombuffer((C.c_char * nbytes).from_address(0), dtype=torch.uint8) if False else a
    root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    # slots 1 GB apart (2 MB aligned), plus a delta per array
    rows = []
    deltas = (0, 512, 4096, 65536, 1 << 20, (1 << 20) + 4096 + 512)
    src0 = torch.randn((F, J, 4), device="cuda")
    for d_src, d_pos, d_rm in [(0, 0, 0)] + [(a, b, c) for a in deltas[1:4] for b in (0,) for c in (0,)] + [(0, b, 0) for b in deltas[1:]] + [(0, 0, c) for c in deltas[1:]] + [(512, 4096, 65536), (65536, 512, 4096), (4096, 65536, 512)]:
        a_src, a_pos, a_rm = base + d_src, base + (1 << 30) + d_pos, base + (3 << 29) + d_rm
        assert a_rm + nb["rm"] <= big.data_ptr() + big.numel()
        _lib.call("pm_memcpy_d2d", C.c_void_p(a_src), C.c_void_p(src0.data_ptr()), nb["src"], None) if hasattr(_lib.lib(), "pm_memcpy_d2d") else None
        call = lambda: _lib.call("pm_fk_f32", C.c_void_p(a_src), C.c_void_p(root.data_ptr()), C.c_void_p(off.data_ptr()), 0, par.ctypes.data_as(C.c_void_p), F, J, C.c_void_p(a_pos), C.c_void_p(a_rm), None)  # noqa: E731
        ms, _ = pp.timeit(call)
        print(f"J={J} deltas src/pos/rm = {d_src:8d} / {d_pos:8d} / {d_rm:8d}: {ms * 1e3:7.1f} us  {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%  {_lib.last_kernel_name()[9:40]}", flush=True)
