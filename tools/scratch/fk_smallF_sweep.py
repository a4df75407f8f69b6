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
for name, par in (("chain-like 64", chain_like(64)), ("chain-like 128", chain_like(128)), ("chain-like 256", chain_like(256))):
    cases += [(name, par, lf) for lf in (10, 12, 14, 15, 16, 17)]
for name, par, lf in cases:
    J = len(par); F = 1 << lf
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.randn((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.15
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    line = f"{name:16s} 2^{lf}"
    for v in ("0", "1"):
        os.environ["PM_FK_STREAM"] = v
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        line += f" | {ms * 1e3:6.1f} us {_lib.last_kernel_name().replace('void pm::', '')[:22]:22s}"
    if name != "warm": print(line, flush=True)
    del rot, pos, rm
