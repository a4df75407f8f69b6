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
big = torch.empty(7 << 29, dtype=torch.uint8, device="cuda")
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
    deltas = (0, 512, 65536, 1 << 20)
    src0 = torch.randn((F, J, 4), device="cuda")
    MB = 1 << 20
    def up(x): return (x + 2 * MB - 1) // (2 * MB) * (2 * MB)
    layouts = [("src pos rm, 1 GB apart", 0, 1024 * MB, 1536 * MB), ("src pos rm, 400 / 800 MB", 0, 400 * MB, 800 * MB),
               ("src pos rm, back to back (2 MB aligned)", 0, up(nb["src"]), up(nb["src"]) + up(nb["pos"])),
               ("rm pos src, back to back", up(nb["rm"]) + up(nb["pos"]), up(nb["rm"]), 0),
               ("pos rm src, back to back", up(nb["pos"]) + up(nb["rm"]), 0, up(nb["pos"])),
               ("rm src pos, back to back", up(nb["rm"]), up(nb["rm"]) + up(nb["src"]), 0),
               ("rm pos src, 1 GB apart", 2048 * MB, 1024 * MB, 0), ("rm pos src, 600 / 1200 MB", 1200 * MB, 600 * MB, 0)]
    for name, o_src, o_pos, o_rm in layouts:
        a_src, a_pos, a_rm = base + o_src, base + o_pos, base + o_rm
        spans = sorted([(a_src, nb["src"]), (a_pos, nb["pos"]), (a_rm, nb["rm"])])
        assert all(spans[i][0] + spans[i][1] <= spans[i + 1][0] for i in range(2)) and spans[2][0] + spans[2][1] <= big.data_ptr() + big.numel(), name
        dist_pos, dist_rm = o_pos - o_src, o_rm - o_src
        call = lambda: _lib.call("pm_fk_f32", C.c_void_p(a_src), C.c_void_p(root.data_ptr()), C.c_void_p(off.data_ptr()), 0, par.ctypes.data_as(C.c_void_p), F, J, C.c_void_p(a_pos), C.c_void_p(a_rm), None)  # noqa: E731
        ms, _ = pp.timeit(call)
        print(f"J={J} {name:42s} (pos - src {dist_pos // MB:6d} MB, rm - src {dist_rm // MB:6d} MB): {ms * 1e3:7.1f} us  {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%", flush=True)
    # separate torch allocations (what perf_probe times)
    t_src = torch.randn((F, J, 4), device="cuda"); t_pos = torch.empty((F, J, 3), device="cuda"); t_rm = torch.empty((F, J, 3, 3), device="cuda")
    call = lambda: _lib.call("pm_fk_f32", C.c_void_p(t_src.data_ptr()), C.c_void_p(root.data_ptr()), C.c_void_p(off.data_ptr()), 0, par.ctypes.data_as(C.c_void_p), F, J, C.c_void_p(t_pos.data_ptr()), C.c_void_p(t_rm.data_ptr()), None)  # noqa: E731
    ms, _ = pp.timeit(call)
    print(f"J={J} separate torch allocations (pos - src {(t_pos.data_ptr() - t_src.data_ptr()) / MB:.1f} MB, rm - src {(t_rm.data_ptr() - t_src.data_ptr()) / MB:.1f} MB): {ms * 1e3:7.1f} us  {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%", flush=True)
    del t_src, t_pos, t_rm
