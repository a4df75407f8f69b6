#!/usr/bin/env python3
"""Clips of real length (2^10 ... 2^16 frames): every skeleton op of the path on the 22-joint BVH body and SMPL-H's 52 joints, microseconds per
launch (back-to-back launches between two HIP events) and the kernel the production dispatch picked.  At these sizes a launch is a handful of
waves per CU at most: time is one tile's latency, not bandwidth -- which is why the long-skeleton kernels (64 frames to a wave) stay out
(common.hpp: lane_per_frame_pays)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib, synthetic as syn
pp.SUSTAINED = 200
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
short = lambda: _lib.last_kernel_name().replace("void pm::", "").replace("pm::", "").split("(")[0][:34]  # noqa: E731
warm = torch.empty(1 << 24, device="cuda")
for _ in range(200): warm.add_(1.0)
torch.cuda.synchronize()
for name, par in (("22-joint body", syn.PARENTS_22), ("SMPL-H (52)", syn.PARENTS_52)):
    J = len(par); pp_ = np.asarray(par, np.int32).ctypes.data_as(C.c_void_p)
    for lf in (10, 12, 14, 16):
        F = 1 << lf
        rot = torch.randn((F, J, 4), device="cuda"); rot /= rot.norm(dim=-1, keepdim=True)
        root = torch.randn((F, 3), device="cuda"); off = torch.randn((J, 3), device="cuda") * 0.15; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda"); dq = torch.empty((F, J, 8), device="cuda")
        tr = torch.empty((F, J, 3), device="cuda"); q = torch.empty((F, J, 4), device="cuda")
        rows = []
        for label, fn in (("fk", lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None)),
                          ("to_root_dual_quat", lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(root), pp_, P(off), F, J, P(dq), None)),
                          ("from_root_dual_quat", lambda: _lib.call("pm_from_root_dq_f32", P(dq), pp_, F, J, P(tr), P(q), None)),
                          ("mirror (all)", lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, None, 0, F, J, P(q), None)),
                          ("from_root_positions", lambda: _lib.call("pm_from_root_positions_f32", P(pos), pp_, P(off), F, J, P(q), None))):
            ms, _ = pp.timeit(fn)
            rows.append(f"{label} {ms * 1e3:5.1f} us [{short()}]")
        print(f"{name:14s} 2^{lf}: " + " | ".join(rows), flush=True)
