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
for name, par in (("SMPL-H", syn.PARENTS_52), ("chain-like 32", chain_like(32)), ("chain-like 24", chain_like(24)), ("chain-like 64", chain_like(64)), ("chain-like 128", chain_like(128))):
    cases += [(name, par, lf) for lf in (10, 12, 13, 14, 15, 16, 17)]
for name, par, lf in cases:
    J = len(par); F = 1 << lf
    pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
    line = f"{name:16s} 2^{lf}"
    for env in ({"PM_IK_ORDER": "0"}, {"PM_IK_ORDER": "1"}):
        for k in list(os.environ):
            if k.startswith("PM_IK"): del os.environ[k]
        os.environ.update(env)
        ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), par.ctypes.data_as(C.c_void_p), P(off), F, J, P(out), None))
        line += f" | {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}% {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')[:26]:26s}"
    if name != "warm": print(line, flush=True)
    del pos, out
