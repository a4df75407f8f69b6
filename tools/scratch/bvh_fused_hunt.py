"""reproduce the slow state of pm_bvh_rotations_f32 at 2^12 / 2^14 frames seen inside tools/unroll_probe.py and find which buffer it follows"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
import pymotion_amd.rotations.quat_torch as quat_t
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = 22
# the front part of unroll_probe
for lg in (10, 12, 14, 16, 18, 20):
    T = 1 << lg
    q = torch.randn((T, S, 4), device="cuda"); out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
    pp.timeit(lambda: _lib.call("pm_quat_unroll_f32", P(q), T, S, P(out), P(ws), None))
for B, T in ((16384, 64), (4096, 256), (64, 16384)):
    q = torch.randn((B, T, S, 4), device="cuda"); out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_batched_workspace_bytes(B, T, S)) + 16, dtype=torch.uint8, device="cuda")
    pp.timeit(lambda: _lib.call("pm_quat_unroll_batched_f32", P(q), B, T, S, P(out), P(ws), None))
    pp.timeit(lambda: quat_t.unroll(q, 1))
order_h = np.tile(np.array([2, 0, 1], np.uint8), (S, 1))
order_d = torch.from_numpy(order_h).cuda()
def t_fused(deg, out, ws, T):
    ms, _ = pp.timeit(lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None))
    return ms * 1e3
for lg in (10, 12, 14, 16):
    T = 1 << lg
    deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
    rad = torch.deg2rad(deg)
    q1, q2, out = (torch.empty((T, S, 4), device="cuda") for _ in range(3))
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
    base = t_fused(deg, out, ws, T)
    line = f"T=2^{lg}: fused {base:7.1f} us  ptrs deg {deg.data_ptr():#x} out {out.data_ptr():#x} ws {ws.data_ptr():#x} ({ws.numel()} B)"
    ws2 = torch.empty(ws.numel() + 8192, dtype=torch.uint8, device="cuda")
    out2 = torch.empty((T, S, 4), device="cuda"); deg2 = deg.clone()
    line += f" | new ws {t_fused(deg, out, ws2, T):7.1f} | ws+4096 {t_fused(deg, out, ws2[4096:], T):7.1f} | new out {t_fused(deg, out2, ws, T):7.1f} | new deg {t_fused(deg2, out, ws, T):7.1f} | all new {t_fused(deg2, out2, ws2, T):7.1f} | again {t_fused(deg, out, ws, T):7.1f}"
    print(line, flush=True)
