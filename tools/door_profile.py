#!/usr/bin/env python3
"""Where the torch front door's microseconds go on a 1000-frame clip (config 1): cProfile over 20 000 calls of skeleton_torch.fk."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pymotion_amd.ops.skeleton_torch as skt
from pymotion_amd import synthetic as syn

rot, root, off, par = syn.fk_workload(1000, normalized=True)
tr, tg, to = (torch.from_numpy(a).cuda() for a in (rot, root, off))
tp = torch.from_numpy(par)
for _ in range(1000):
    skt.fk(tr, tg, to, tp)
torch.cuda.synchronize()
N = 20000
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    skt.fk(tr, tg, to, tp)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
