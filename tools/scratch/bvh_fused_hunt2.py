"""in a process where the all-new buffer set is slow: which buffers, which phase"""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
import pymotion_amd.rotations.quat_torch as quat_t
pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = 22
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
def t_fused(deg, out, ws, T, env=None):
    for k in list(os.environ):
        if k.startswith("PM_UNROLL"): del os.environ[k]
    if env: os.environ.update(env)
    ms, _ = pp.timeit(lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None))
    return ms * 1e3
T = 1 << 10
deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
rad = torch.deg2rad(deg)
q1, q2, out = (torch.empty((T, S, 4), device="cuda") for _ in range(3))
ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
ws2 = torch.empty(ws.numel() + 8192, dtype=torch.uint8, device="cuda")
out2 = torch.empty((T, S, 4), device="cuda"); deg2 = deg.clone()
names = {"deg": deg, "deg2": deg2, "out": out, "out2": out2, "ws": ws, "ws2": ws2, "q1": q1, "q2": q2}
print({k: hex(v.data_ptr()) for k, v in names.items()}, {k: v.numel() * v.element_size() for k, v in names.items()})
for d in ("deg", "deg2"):
    for o in ("out", "out2", "q1", "q2"):
        for w in ("ws", "ws2"):
            print(f"{d:5s} {o:5s} {w:4s}: {t_fused(names[d], names[o], names[w], T):7.1f} us", flush=True)
print("phases on (deg2, out2, ws2):", {k: round(t_fused(deg2, out2, ws2, T, {"PM_UNROLL_STATIC": k}), 1) for k in ("0", "1", "4", "5", "8")})
qq = torch.randn((T, S, 4), device="cuda")
ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_f32", P(qq), T, S, P(out2), P(ws2), None)); print("plain unroll -> out2, ws2:", round(ms * 1e3, 1))
out2.zero_(); torch.cuda.synchronize()
print("after out2.zero_():", round(t_fused(deg2, out2, ws2, T), 1))
