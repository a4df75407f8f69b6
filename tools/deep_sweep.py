import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = int(os.environ.get("SUSTAINED", "40"))
P = lambda t: C.c_void_p(t.data_ptr())
F = 1 << 19
for J in [int(x) for x in sys.argv[1].split(",")]:
    par = np.maximum(np.arange(J) - 1, 0).astype(np.int32); par[J // 2] = 0; par[3 * J // 4] = J // 4
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.randn((F, 3), device="cuda")
    off = torch.randn((J, 3), device="cuda") * 0.15; off[0] = 0
    dq = torch.empty((F, J, 8), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    ms, _ = pp.timeit(lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None))
    print(f"J={J:3d}: to_root {ms*1e3:7.1f} us {F*(48*J+12)/ms/1e6/80:5.1f}%  {_lib.last_kernel_name() if hasattr(_lib,'last_kernel_name') else ''}", flush=True)
    del rot, dq
