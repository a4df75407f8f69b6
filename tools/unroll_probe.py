#!/usr/bin/env python3
"""quat.unroll over clip lengths (S = 22 series unless given): the one-pass look-back kernel against the three-pass scan
(PMHIP_VARIANT=tuning PM_UNROLL_ONEPASS=0) and its tile sizes (PM_UNROLL_R, PM_UNROLL_NT).  Tuning aid."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import tools.perf_probe as pp
from pymotion_amd import _lib

pp.SUSTAINED = 40
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
S = int(sys.argv[1]) if len(sys.argv) > 1 else 22
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PM_UNROLL"))
for lg in (10, 12, 14, 16, 18, 20):
    T = 1 << lg
    q = torch.randn((T, S, 4), device="cuda")
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_workspace_bytes(T, S)) + 16, dtype=torch.uint8, device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_f32", P(q), T, S, P(out), P(ws), None))
    print(f"[{tag}] T=2^{lg} S={S}: {ms * 1e3:8.1f} us  {T * S * 32 / ms / 1e6 / 80:5.1f}% of 8 TB/s on 32 B/quaternion", flush=True)
# batches of clips [B, T, S, 4] along T: one launch, nothing transposed (raw ABI), and the torch door end to end
import pymotion_amd.rotations.quat_torch as quat_t

for B, T in ((16384, 64), (4096, 256), (64, 16384)):
    q = torch.randn((B, T, S, 4), device="cuda")
    out = torch.empty_like(q)
    ws = torch.empty(int(_lib.lib().pm_quat_unroll_batched_workspace_bytes(B, T, S)) + 16, dtype=torch.uint8, device="cuda")
    ms, _ = pp.timeit(lambda: _lib.call("pm_quat_unroll_batched_f32", P(q), B, T, S, P(out), P(ws), None))
    ms2, _ = pp.timeit(lambda: quat_t.unroll(q, 1))
    print(f"[{tag}] batch B={B} T={T} S={S}: {ms * 1e3:8.1f} us  {B * T * S * 32 / ms / 1e6 / 80:5.1f}% of 8 TB/s on 32 B/quaternion;"
          f" quat_torch.unroll(q, 1) end to end {ms2 * 1e3:8.1f} us", flush=True)
