import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymotion_amd import _lib, synthetic as syn
dev=torch.device("cuda:0"); F=1<<20; J=22
rot=torch.randn((F,J,4),device=dev); root=torch.rand((F,3),device=dev)*4-2
off=torch.from_numpy(syn.make_offsets(J,np.random.default_rng(0))).to(dev)
pos=torch.empty((F,J,3),device=dev); rm=torch.empty((F,J,3,3),device=dev)
pp=syn.PARENTS_22.ctypes.data_as(C.c_void_p); p=lambda t: C.c_void_p(t.data_ptr())
fn=lambda: _lib.call("pm_fk_f32",p(rot),p(root),p(off),0,pp,F,J,p(pos),p(rm),None)
ev=[C.c_void_p(),C.c_void_p()]
for e in ev: _lib.call("pm_event_create",C.byref(e))
def timed(n):
    _lib.call("pm_event_record",ev[0],None)
    for _ in range(n): fn()
    _lib.call("pm_event_record",ev[1],None)
    ms=C.c_float(); _lib.call("pm_event_elapsed_ms",ev[0],ev[1],C.byref(ms)); return ms.value/n*1e3
for _ in range(1500): fn()
torch.cuda.synchronize()
for idle_ms in (0, 0.2, 1, 3, 10, 30, 100, 0):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    if idle_ms: time.sleep(idle_ms/1e3)
    a=timed(20); b=timed(50); c=timed(200)
    print(f"idle {idle_ms:6.1f} ms -> first20 {a:.0f} next50 {b:.0f} next200 {c:.0f}", flush=True)
# many short segments with sync between (sync gap only)
for _ in range(300): fn()
torch.cuda.synchronize()
print("sync-gap segments:", " ".join("%.0f"%timed(20) for _ in range(8)))
