#!/usr/bin/env python3
"""Which allocations get the fast level of fk J = 52 (2^18 frames)?  In ONE process: sets of (src, pos, rotmats) made in different ways -- separate
torch allocations (twice), carved back to back out of one buffer, carved with gaps, separate allocations with padded sizes, pm_malloc (plain hipMalloc)
in two orders -- each timed twice in turn.  Prints the device pointers beside the times.
    python tools/alloc_kind_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

J, F = 52, 1 << 18
NS, NP, NR = F * J * 4, F * J * 3, F * J * 9


def timed(fn, n=400):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
    root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    seed = torch.randn(NS, device="cuda")
    seed_host = np.ascontiguousarray(seed.cpu().numpy())
    sets, keep = [], []

    def add(name, ps, pp, pr):
        torch.cuda.synchronize()
        sets.append((name, ps, pp, pr))

    def torch_set(name, pad=0):
        s = torch.empty(NS + pad, device="cuda"); p = torch.empty(NP + pad, device="cuda"); r = torch.empty(NR + pad, device="cuda")
        s[:NS].copy_(seed); keep.extend([s, p, r])
        add(name, s.data_ptr(), p.data_ptr(), r.data_ptr())

    def carve(name, gap_floats):
        al = lambda n: (n + (1 << 19) - 1) & ~((1 << 19) - 1)  # noqa: E731
        tot = al(NS) + al(NP) + al(NR) + 2 * gap_floats
        b = torch.empty(tot, device="cuda"); keep.append(b)
        b[:NS].copy_(seed)
        o1 = al(NS) + gap_floats; o2 = o1 + al(NP) + gap_floats
        add(name, b.data_ptr(), b.data_ptr() + 4 * o1, b.data_ptr() + 4 * o2)

    def raw_set(name, order):
        ptr = {}
        for k in order:
            d = C.c_void_p()
            _lib.call("pm_malloc", C.byref(d), 4 * {"s": NS, "p": NP, "r": NR}[k])
            ptr[k] = d.value
        _lib.call("pm_memcpy_h2d", C.c_void_p(ptr["s"]), C.c_void_p(seed_host.ctypes.data), 4 * NS, None)
        _lib.call("pm_stream_synchronize", None)
        torch.cuda.synchronize()
        add(name, ptr["s"], ptr["p"], ptr["r"])

    if os.environ.get("AKP_ALTERNATE") == "1":   # torch / hipMalloc / torch / hipMalloc ...: is it the allocator?
        for i in range(6):
            torch_set(f"torch, separate #{i + 1}")
            raw_set(f"hipMalloc #{i + 1}", "spr" if i % 2 == 0 else "rps")
    else:
        torch_set("torch, separate #1")
        torch_set("torch, separate #2")
        carve("carved, back to back", 0)
        carve("carved, 64 MB gaps", 16 << 20)
        carve("carved, 1 GB gaps", 256 << 20)
        torch_set("torch, sizes + 2 MB", 1 << 19)
        torch_set("torch, sizes + 96 MB", 24 << 20)
        raw_set("hipMalloc src, pos, rotmats", "spr")
        raw_set("hipMalloc rotmats, pos, src", "rps")
        torch_set("torch, separate #3")
        torch_set("torch, separate #4")
    print(f"fk J = {J}, F = {F}: us a launch, two visits; pointers src / pos / rotmats")
    res = [[] for _ in sets]
    for rep in range(2):
        for i, (name, ps, pp, pr) in enumerate(sets):
            fn = lambda: _lib.call("pm_fk_f32", C.c_void_p(ps), C.c_void_p(root.data_ptr()), C.c_void_p(off.data_ptr()), 0, par.ctypes.data_as(C.c_void_p), F, J,  # noqa: E731
                                   C.c_void_p(pp), C.c_void_p(pr), None)
            res[i].append(timed(fn))
    for (name, ps, pp, pr), ts in zip(sets, res):
        print(f"  {name:30s} " + "  ".join(f"{t:6.1f}" for t in ts) + f"   {ps:#x} {pp:#x} {pr:#x}")
    for pth in ("/sys/kernel/debug/dri", "/sys/kernel/debug/kfd"):
        print(pth, os.path.isdir(pth) and os.listdir(pth)[:12])


if __name__ == "__main__":
    main()
