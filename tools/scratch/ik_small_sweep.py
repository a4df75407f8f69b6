import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 60
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
cases = [("22 body", syn.PARENTS_22)] + [("chain-like %d" % J, chain_like(J)) for J in [int(x) for x in os.environ.get("JS", "8,12,16,18,20,21,22,23,24").split(",")]]
for name, par in cases:
    J = len(par); F = 1 << 20
    pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
    line = f"{name:16s}"
    for env in ({"PM_IK_ORDER": "0"}, {"PM_IK_ORDER": "1"}):
        for k in list(os.environ):
            if k.startswith("PM_IK"): del os.environ[k]
        os.environ.update(env)
        ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), par.ctypes.data_as(C.c_void_p), P(off), F, J, P(out), None))
        line += f" | {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}% {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')[:28]:28s}"
    print(line, flush=True)
    del pos, out
