import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 200
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
for name, par in (("warm", syn.PARENTS_22), ("22-joint body", syn.PARENTS_22), ("SMPL-H (52)", syn.PARENTS_52), ("chain-like 72", chain_like(72)), ("chain-like 128", chain_like(128))):
    J = len(par); pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
    for lf in (10, 12, 14, 15):
        F = 1 << lf
        rot = torch.randn((F, J, 4), device="cuda"); rot /= rot.norm(dim=-1, keepdim=True)
        root = torch.randn((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.15
        dq = torch.empty((F, J, 8), device="cuda"); mi = torch.empty((F, J, 4), device="cuda")
        line = f"{name:14s} 2^{lf}"
        for var, fn in (("PM_DQ_CHAINS", lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None)),
                        ("PM_MIRROR_CHAINS", lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, None, 0, F, J, P(mi), None))):
            for v in ("-1", "0", "2", "4"):
                for k in list(os.environ):
                    if k.startswith("PM_DQ") or k.startswith("PM_MIRROR"): del os.environ[k]
                os.environ[var] = v
                try:
                    ms, _ = pp.timeit(fn)
                    line += f" {ms * 1e3:5.1f}"
                except Exception as e:
                    line += "   err"
            line += " |"
        if name != "warm": print(line + "   (to_root: default / 1 / 2 / 4 chains | mirror: same)", flush=True)
