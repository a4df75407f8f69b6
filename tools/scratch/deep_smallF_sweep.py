import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 100
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
cases = [("warm", chain_like(30), 14)] * 5
for name, par in (("SMPL-H", syn.PARENTS_52), ("chain-like 40", chain_like(40)), ("chain-like 72", chain_like(72)), ("chain-like 128", chain_like(128))):
    cases += [(name, par, lf) for lf in (10, 12, 14, 15, 16, 17, 18)]
for name, par, lf in cases:
    J = len(par); F = 1 << lf
    rot = torch.randn((F, J, 4), device="cuda"); rot /= rot.norm(dim=-1, keepdim=True)
    root = torch.randn((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.15
    dq = torch.empty((F, J, 8), device="cuda"); mi = torch.empty((F, J, 4), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    line = f"{name:16s} 2^{lf}"
    for var, fn in (("PM_DQ_DEEP", lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None)),
                    ("PM_MIRROR_DEEP", lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, None, 0, F, J, P(mi), None))):
        for v in ("0", "1"):
            for k in list(os.environ):
                if k.startswith("PM_DQ") or k.startswith("PM_MIRROR"): del os.environ[k]
            os.environ[var] = v
            ms, _ = pp.timeit(fn)
            line += f" | {ms * 1e3:6.1f} us {_lib.last_kernel_name().replace('void pm::', '').replace('pm::','')[:18]:18s}"
    if name != "warm": print(line, flush=True)
    del rot, dq, mi
