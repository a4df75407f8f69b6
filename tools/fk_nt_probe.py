import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())
F, J = 1 << 18, 52
par = np.ascontiguousarray(syn.PARENTS_52, dtype=np.int32)
rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
off = torch.randn((J, 3), device="cuda") * 0.1
pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
big = torch.empty(F * J * 12, device="cuda")
fn = lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)
for rep in range(2):
    ms, _ = pp.timeit(lambda: _lib.call("pm_stream_ceiling_f32", P(rot), P(big), F, 4 * J, 12 * J, None))
    print(f"copy ceiling {ms*1e3:7.1f} us")
    for nt in ("1", "2", "3", "4", "6", "8", "16"):
        os.environ["PM_FK_NT"] = nt
        ms, _ = pp.timeit(fn)
        print(f"PM_FK_NT={nt:2s}: {ms*1e3:7.1f} us  {F*(64*J+12)/ms/1e6/80:5.1f}%  {_lib.last_kernel_name()[9:70]}", flush=True)
    os.environ.pop("PM_FK_NT")
    for w4 in ("0", "1"):
        os.environ["PM_FK_W4"] = w4
        ms, _ = pp.timeit(fn)
        print(f"PM_FK_W4={w4}: {ms*1e3:7.1f} us  {F*(64*J+12)/ms/1e6/80:5.1f}%", flush=True)
    os.environ.pop("PM_FK_W4")
