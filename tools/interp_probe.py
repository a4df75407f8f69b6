#!/usr/bin/env python3
"""interpolate_positions (2x up-sampling of a [T, J * 3] clip) over row lengths that are / are not multiples of four floats."""
import sys,os,ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED=100
p=lambda t: C.c_void_p(t.data_ptr())
dev='cuda'
for J in (22,24,21,52):
    Tn=1<<18; Sn=1<<19
    idx=(torch.arange(Sn,device=dev)//2).clamp(max=Tn-2).to(torch.int32); w=torch.rand(Sn,device=dev)
    src=torch.randn((Tn,J*3),device=dev); dst=torch.empty((Sn,J*3),device=dev)
    ms,_=pp.timeit(lambda: _lib.call("pm_interpolate_linear_f32",p(src),p(idx),p(w),1,Tn,Sn,J*3,p(dst),None))
    print(f"J={J} B={J*3}: {ms*1e3:.1f} us  {(Tn+Sn)*J*12/ms/1e6/80:.1f}%")
