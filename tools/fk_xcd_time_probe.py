#!/usr/bin/env python3
"""do the eight XCDs finish the J = 52 kernel at the same time?  The tuning build's fk_pipe_kernel records, per workgroup, the XCC it ran on and
its start / end on the 100 MHz counter (env PM_FK_TIMES_PTR = a device buffer).  Every XCD gets the same number of workgroups (xcd_tile: a
contiguous eighth of the tiles each), so the kernel ends with the slowest of them.  Prints per XCD: workgroups, the span from the kernel's first
start to that XCD's last end, the mean workgroup duration."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
F, J = 1 << 18, 52
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
off = torch.randn((J, 3), device="cuda") * 0.1
pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
fn = lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
ms, _ = pp.timeit(fn)
print(f"fk J=52 2^18 frames: {ms * 1e3:.1f} us  {_lib.last_kernel_name()[9:75]}")
nwg = 1 << 16
times = torch.zeros((nwg, 3), dtype=torch.int64, device="cuda")
os.environ["PM_FK_TIMES_PTR"] = str(times.data_ptr())
for rep in range(3):
    for _ in range(20): fn()
    times.zero_(); torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()
    t = times.cpu().numpy()
    t = t[t[:, 2] > 0]
    t0 = t[:, 1].min()
    print(f"rep {rep}: {len(t)} workgroups, kernel span {(t[:, 2].max() - t0) / 100:.1f} us")
    for x in range(8):
        m = t[t[:, 0] == x]
        if len(m):
            d = (m[:, 2] - m[:, 1]) / 100.0
            print(f"   XCD {x}: {len(m):6d} workgroups, last end at {(m[:, 2].max() - t0) / 100:7.1f} us, first start {(m[:, 1].min() - t0) / 100:5.1f} us, "
                  f"workgroup mean {d.mean():6.2f} us (p10 {np.percentile(d, 10):5.2f}, p90 {np.percentile(d, 90):5.2f})")
    # how busy is the chip over time: workgroups in flight at 10 us marks
    marks = np.arange(0, (t[:, 2].max() - t0) / 100, 10.0)
    infl = [int(((t[:, 1] - t0) / 100 <= m_).sum() - ((t[:, 2] - t0) / 100 <= m_).sum()) for m_ in marks]
    print("   workgroups in flight every 10 us:", infl)
os.environ.pop("PM_FK_TIMES_PTR")
