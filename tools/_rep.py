import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from pymotion_amd import _lib, synthetic as syn
dev=torch.device("cuda:0"); F=1<<20; J=22
def run(tag):
    rot=torch.randn((F,J,4),device=dev); root=torch.rand((F,3),device=dev)*4-2
    off=torch.from_numpy(syn.make_offsets(J,np.random.default_rng(0))).to(dev)
    pos=torch.empty((F,J,3),device=dev); rm=torch.empty((F,J,3,3),device=dev)
    pp=syn.PARENTS_22.ctypes.data_as(C.c_void_p); p=lambda t: C.c_void_p(t.data_ptr())
    fn=lambda: _lib.call("pm_fk_f32",p(rot),p(root),p(off),0,pp,F,J,p(pos),p(rm),None)
    ev=[C.c_void_p(),C.c_void_p()]
    for e in ev: _lib.call("pm_event_create",C.byref(e))
    for _ in range(3): fn()
    out=[]
    for rep in range(int(os.environ.get('REPS','6'))):
        _lib.call("pm_event_record",ev[0],None)
        for _ in range(50): fn()
        _lib.call("pm_event_record",ev[1],None)
        ms=C.c_float(); _lib.call("pm_event_elapsed_ms",ev[0],ev[1],C.byref(ms)); out.append(ms.value/50*1e3)
    print(tag, "ptrs %x %x %x"%(rot.data_ptr()>>21, pos.data_ptr()>>21, rm.data_ptr()>>21), " ".join("%.0f"%x for x in out), flush=True)
    return rot,pos,rm
keep=[]
for i in range(int(os.environ.get("ALLOCS","4"))):
    keep.append(run("alloc%d"%i))   # keep previous buffers alive -> new addresses each time
