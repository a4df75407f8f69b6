#!/usr/bin/env python3
"""Is the timing level of an allocation a property of WHERE it lies?  One big buffer, cut into windows: write-only (fill), read-only (a reduction that
reads every byte once) and copy bandwidth per window, several passes; then fk J = 52 (2^18 frames) on arrays carved out of the buffer at each window.
    python tools/bw_map_probe.py [GB] [window MB]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn


def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
    wmb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nwin = int(gb * 1024 // wmb)
    wf = wmb * (1 << 20) // 4
    buf = torch.empty(nwin * wf, dtype=torch.float32, device="cuda")
    buf.zero_()
    print(f"buffer {nwin * wmb / 1024:.1f} GB at {hex(buf.data_ptr())}, windows of {wmb} MB; GB/s per window: write (fill_), read (pm_stream_plain 1:0 is not available: torch.sum), copy to the next window")
    rows = []
    for w in range(nwin):
        x = buf[w * wf:(w + 1) * wf]
        y = buf[((w + 1) % nwin) * wf:((w + 1) % nwin + 1) * wf]
        tw = timed(lambda: x.fill_(1.0), 20)
        tr = timed(lambda: torch.sum(x), 20)
        tc = timed(lambda: y.copy_(x), 10)
        rows.append((wmb * 1.048576e-3 / tw, wmb * 1.048576e-3 / tr, 2 * wmb * 1.048576e-3 / tc))
    for w, (a, b, c) in enumerate(rows):
        print(f"  window {w:3d} (+{w * wmb / 1024:6.2f} GB): write {a:7.0f}  read {b:7.0f}  copy {c:7.0f}")
    a = np.array(rows)
    print("  min / median / max: write %.0f / %.0f / %.0f   read %.0f / %.0f / %.0f   copy %.0f / %.0f / %.0f" % (
        a[:, 0].min(), np.median(a[:, 0]), a[:, 0].max(), a[:, 1].min(), np.median(a[:, 1]), a[:, 1].max(), a[:, 2].min(), np.median(a[:, 2]), a[:, 2].max()))
    # fk J = 52 on arrays carved at window w: src | pos | rotmats back to back (2 MB aligned), 873 MB in all
    J, F = 52, 1 << 18
    par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
    need = F * J * (4 + 3 + 9) + (1 << 20)
    root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    al = lambda n: (n + (1 << 19) - 1) & ~((1 << 19) - 1)  # noqa: E731  (2 MB in floats)
    print(f"fk J = {J}, F = {F} with src | pos | rotmats carved at the window's start (us a launch), three passes:")
    res = {}
    for rep in range(3):
        for w in range(0, nwin - (need // wf + 1), max(1, (1024 // wmb))):
            base = w * wf
            src = buf[base:base + F * J * 4].view(F, J, 4)
            o1 = base + al(F * J * 4)
            pos = buf[o1:o1 + F * J * 3].view(F, J, 3)
            o2 = o1 + al(F * J * 3)
            rm = buf[o2:o2 + F * J * 9].view(F, J, 3, 3)
            if rep == 0: src.normal_()
            fn = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
            for _ in range(60): fn()
            res.setdefault(w, []).append(timed(fn, 300) * 1e6)
    for w, ts in res.items():
        print(f"  window {w:3d} (+{w * wmb / 1024:6.2f} GB): " + "  ".join(f"{t:6.1f}" for t in ts))


if __name__ == "__main__":
    main()
