import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 200
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
par = syn.PARENTS_22; J = 22; pp_ = par.ctypes.data_as(C.c_void_p)
for lf in (12, 13, 14, 15, 16, 17, 18):
    F = 1 << lf
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.randn((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.15
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    line = f"fk 22 joints 2^{lf}:"
    for v in ("0", "20", "16", "12", "8", "4"):
        for k in list(os.environ):
            if k.startswith("PM_FK"): del os.environ[k]
        if v != "0": os.environ["PM_FK_FPW"] = v
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        line += f"  {('auto' if v == '0' else 'FPW' + v):>6s} {ms * 1e3:6.1f}"
    print(line + "  us", flush=True)
