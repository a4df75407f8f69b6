import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = int(os.environ.get("SUST", "40"))
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = 22
order_h = np.tile(np.array([2, 0, 1], np.uint8), (S, 1))
order_d = torch.from_numpy(order_h).cuda()
for T in [int(x) for x in os.environ.get("TS", "4096,8192,12288,16384,20000,24576,32768,65536").split(",")]:
    row = []
    for rep in range(4):
        deg = (torch.randn((T, S, 3), device="cuda").cumsum(0) * 5.0).contiguous()
        out = torch.empty((T, S, 4), device="cuda")
        ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
        if os.environ.get("GARBAGE"): ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device="cuda")) if os.environ["GARBAGE"] == "rand" else ws.fill_(255)
        fused = lambda: _lib.call("pm_bvh_rotations_f32", P(deg), order_h.ctypes.data_as(C.c_void_p), T, S, P(out), P(ws), None)  # noqa: E731
        ms, _ = pp.timeit(fused)
        name = _lib.last_kernel_name().replace("void pm::", "")[:44]
        # one isolated launch, timed with sync
        torch.cuda.synchronize(); import time; t0 = time.perf_counter(); fused(); torch.cuda.synchronize(); iso = (time.perf_counter() - t0) * 1e6
        # correctness vs three launches
        rad = torch.deg2rad(deg); q1 = torch.empty((T, S, 4), device="cuda"); q2 = torch.empty_like(q1); o3 = torch.empty_like(q1)
        _lib.call("pm_quat_from_euler_f32", P(rad), P(order_d), S, T * S, P(q1), None)
        _lib.call("pm_quat_unroll_f32", P(q1), T, S, P(q2), P(ws), None)
        _lib.call("pm_quat_normalize_f32", P(q2), T * S, C.c_float(1e-8), P(o3), None)
        fused(); torch.cuda.synchronize()
        err = float((out - o3).abs().max())
        row.append(f"{ms * 1e3:7.1f}us(iso {iso:6.0f}, err {err:.1e})")
        del deg, out, ws
    print(f"T={T:6d} {name}: " + " ".join(row), flush=True)
