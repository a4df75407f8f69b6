import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 200
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for name, par in (("warm", syn.PARENTS_22), ("22-joint body", syn.PARENTS_22), ("SMPL-H (52)", syn.PARENTS_52)):
    J = len(par); pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
    for lf in (10, 12, 14, 15, 16, 17):
        F = 1 << lf
        pos = torch.randn((F, J, 3), device="cuda"); off = torch.randn((J, 3), device="cuda"); out = torch.empty((F, J, 4), device="cuda")
        line = f"{name:14s} 2^{lf}"
        for env in ({"PM_IK_ORDER": "0", "PM_IK_CHAINS": "1"}, {"PM_IK_ORDER": "0", "PM_IK_CHAINS": "2"}, {"PM_IK_ORDER": "0", "PM_IK_CHAINS": "4"}, {"PM_IK_ORDER": "0", "PM_IK_CHAINS": "4", "PM_IK_NT": "1"}):
            for k in list(os.environ):
                if k.startswith("PM_IK"): del os.environ[k]
            os.environ.update(env)
            ms, _ = pp.timeit(lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(out), None))
            line += f" | {ms * 1e3:6.1f} us {_lib.last_kernel_name().replace('void pm::from_root_positions_', '')[:24]:24s}"
        if name != "warm": print(line, flush=True)
