import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymotion_amd import _lib, synthetic as syn
dev=torch.device("cuda:0"); F=1<<20; J=22
off=torch.from_numpy(syn.make_offsets(J,np.random.default_rng(0))).to(dev)
pp=syn.PARENTS_22.ctypes.data_as(C.c_void_p)
ev=[C.c_void_p(),C.c_void_p()]
for e in ev: _lib.call("pm_event_create",C.byref(e))
def timed(rot,root,pos,rm,n=40):
    p=lambda t: C.c_void_p(t.data_ptr())
    fn=lambda: _lib.call("pm_fk_f32",p(rot),p(root),p(off),0,pp,F,J,p(pos),p(rm),None)
    for _ in range(3): fn()
    _lib.call("pm_event_record",ev[0],None)
    for _ in range(n): fn()
    _lib.call("pm_event_record",ev[1],None)
    ms=C.c_float(); _lib.call("pm_event_elapsed_ms",ev[0],ev[1],C.byref(ms)); return ms.value/n*1e3
MB=1<<20
def carve(pool, offs_bytes):
    sizes=[F*J*16, F*12, F*J*12, F*J*36]; shapes=[(F,J,4),(F,3),(F,J,3),(F,J,3,3)]
    out=[]
    for o,sz,sh in zip(offs_bytes,sizes,shapes):
        out.append(pool[o:o+sz].view(torch.float32).view(sh))
    return out
pools=[]
for trial in range(3):
    pool=torch.empty(3*1024*MB, dtype=torch.uint8, device=dev); pools.append(pool)
    res=[]
    for delta in (0, 64*1024, 1*MB, 2*MB+4096, 37*MB):
        base=[0, 400*MB, 420*MB, 720*MB]
        offs=[base[0], base[1]+delta, base[2]+2*delta, base[3]+3*delta]
        rot,root,pos,rm=carve(pool, offs)
        rot.normal_(); root.uniform_(-2,2)
        res.append("%d:%.0f"%(delta//1024, timed(rot,root,pos,rm)))
    print("pool%d base %x"%(trial, pool.data_ptr()>>21), " ".join(res), flush=True)
# separate allocations for comparison
for trial in range(3):
    rot=torch.randn((F,J,4),device=dev); root=torch.rand((F,3),device=dev); pos=torch.empty((F,J,3),device=dev); rm=torch.empty((F,J,3,3),device=dev)
    pools.append((rot,root,pos,rm))
    print("separate%d"%trial, "%.0f"%timed(rot,root,pos,rm), flush=True)
