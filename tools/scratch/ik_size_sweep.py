import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = int(os.environ.get("SUST", "60"))
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
par = syn.PARENTS_52
J = 52
for lf in (16, 17, 18, 19, 20, 21):
    F = 1 << lf
    pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
    pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
    ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(out), None))
    print(f"{os.environ.get('PMHIP_VARIANT','prod'):5s} 2^{lf} x {J:3d} {ms * 1e3:7.1f} us {F * 28 * J / ms / 1e6 / 80:5.1f}%", flush=True)
    del pos, out
