import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pymotion_amd.ops.skeleton_torch as skt
import pymotion_amd.rotations.ortho6d_torch as o6t
from pymotion_amd import synthetic as syn
F=1<<18
g=torch.Generator(device="cuda"); g.manual_seed(4)
x=torch.randn((F,52,3,2),generator=g,device="cuda"); root=torch.rand((F,3),generator=g,device="cuda")*4-2
off=torch.from_numpy(syn.make_offsets(52,np.random.default_rng(4),0.15)).cuda(); par=torch.from_numpy(syn.PARENTS_52)
pos,rm,q=skt.fk_from_ortho6d(x,root,off,par,return_quat=True)
q2=o6t.to_quat(x); p2,r2=skt.fk(q2,root,off,par)
print("q",float((q-q2).abs().max()),"pos",float((pos-p2).abs().max()),"rm",float((rm-r2).abs().max()))
p3,r3=skt.fk_from_ortho6d(x,root,off,par)
print("noquat vs quat", float((pos-p3).abs().max()), float((rm-r3).abs().max()))
