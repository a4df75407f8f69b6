#!/usr/bin/env python3
"""Where the NumPy front door's microseconds go on a 1000-frame clip (config 1): cProfile over 2000 calls of skeleton.fk
(host arrays in, float64 host arrays out), and the same call with float32 outputs."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pymotion_amd.ops.skeleton as sk
from pymotion_amd import config, synthetic as syn

rot, root, off, par = syn.fk_workload(1000, normalized=True)
rot64, root64, off64 = rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64)
for name, args in (("float32 inputs", (rot, root, off, par)), ("float64 inputs (what get_data returns)", (rot64, root64, off64, par))):
    for _ in range(200):
        sk.fk(*args)
    t0 = time.perf_counter()
    for _ in range(1000):
        sk.fk(*args)
    print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} us per call")
config.numpy_float64_outputs = False
t0 = time.perf_counter()
for _ in range(1000):
    sk.fk(rot, root, off, par)
print(f"float32 inputs, float32 outputs: {(time.perf_counter() - t0) * 1e3:.1f} us per call")
config.numpy_float64_outputs = True
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    sk.fk(rot64, root64, off64, par)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
