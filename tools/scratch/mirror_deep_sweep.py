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
cases = [("warm", chain_like(30), 18)] * 5 + [("SMPL-H", syn.PARENTS_52, lf) for lf in (16, 17, 18, 19, 20)] + [("chain-like 52", chain_like(52), lf) for lf in (16, 17, 18, 19, 20)] + [("chain-like 40", chain_like(40), lf) for lf in (17, 18, 19, 20)] + [("chain-like 64", chain_like(64), lf) for lf in (17, 18, 19, 20)]
for name, par, lf in cases:
    J = len(par); F = 1 << lf
    rot = torch.randn((F, J, 4), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
    line = f"{name:16s} 2^{lf}"
    for env in ({"PM_MIRROR_DEEP": "0"}, {"PM_MIRROR_DEEP": "1"}):
        for k in list(os.environ):
            if k.startswith("PM_MIRROR"): del os.environ[k]
        os.environ.update(env)
        ms, _ = pp.timeit(lambda: _lib.call("pm_mirror_rotations_f32", P(rot), par.ctypes.data_as(C.c_void_p), None, 0, F, J, P(out), None))
        line += f" | {ms * 1e3:7.1f} us {F * 32 * J / ms / 1e6 / 80:5.1f}% {_lib.last_kernel_name().replace('void pm::', '')[:26]:26s}"
    print(line, flush=True)
    del rot, out
