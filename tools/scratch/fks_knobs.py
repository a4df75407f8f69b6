import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
os.environ["PMHIP_VARIANT"] = "tuning"
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
os.environ["PM_FK_STREAM"] = "1"
KN = ("PM_FKS_FPW", "PM_FKS_CARRY", "PM_FKS_LDSX", "PM_FK_ABLATE", "PM_FKS_RS", "PM_FKS_PS", "PM_FKS_CHS")
for J in (96, 97, 100, 104, 112, 120, 127, 128, 129, 132, 144, 161, 192, 200, 250, 252, 256, 300, 384, 400, 511, 512):
    par = chain_like(J); F = (1 << 19) if J <= 128 else (1 << 18)
    rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
    off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
    pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
    pp_ = par.ctypes.data_as(C.c_void_p)
    cases = [{"PM_FKS_CHS": 32}, {"PM_FKS_CHS": 28}, {"PM_FKS_CHS": 24}] + ([{"PM_FKS_CHS": 32, "PM_FKS_FPW": 20 if J <= 384 else 16}, {"PM_FKS_CHS": 24, "PM_FKS_FPW": 20 if J <= 384 else 16}] if J >= 250 else [])
    row = []
    for env in cases:
        for k in KN: os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in env.items()})
        ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
        row.append(f"{','.join(f'{k[7:]}={v}' for k, v in env.items()) or 'default':18s} {ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}%")
    print(f"J={J:3d} " + " | ".join(row), flush=True)
