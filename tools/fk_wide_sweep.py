#!/usr/bin/env python3
"""fk on wide (random) trees: the wave-per-frame walk over a host-made step list (fk_wide_kernel, fkwide.hip) against what ran before
(PM_FK_WIDE=0 PM_FK_STREAM=0: the tile kernels; PM_FK_WIDE=0: with the streamed walk where it takes the tree), same box, same arrays, tuning
build; the last column is the production dispatch.  FKW_KINDS=bushy,humanoid,chain picks the trees.  Also says whether the
two results agree to the bit (same local rotations, same products in the same order on metre data)."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
kinds = (os.environ.get("FKW_KINDS") or "bushy").split(",")
def humanoid(J):
    """tests/test_gpu_deep.py: spine, head, legs, arms and three-joint fingers off the wrists for as many joints as are left"""
    p = [0]
    def chain(start, n):
        for i in range(n): p.append(start if i == 0 else len(p) - 1)
        return len(p) - 1
    se = chain(0, 6); chain(se, 3); chain(0, 5); chain(0, 5); lw = chain(se, 4); rw = chain(se, 4)
    side = 0
    while len(p) + 3 <= J:
        chain(lw if side == 0 else rw, 3); side ^= 1
    while len(p) < J: p.append(len(p) - 1)
    return np.asarray(p[:J], dtype=np.int32)
def chain_like(J):
    p = np.maximum(np.arange(J) - 1, 0).astype(np.int32); p[J // 2] = 0; p[3 * J // 4] = J // 4
    return p
def main():
    for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,96,128,129,130,160,192,200,256,300,384,400,511,512").split(",")]:
        for kind in kinds:
            par = chain_like(J) if kind == "chain" else humanoid(J) if kind == "humanoid" else syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
            depth = int(syn.depth_of(par).max())
            F = int(os.environ.get("FKW_F", (1 << 19) if J <= 128 else (1 << 18)))
            rot = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
            off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
            pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
            pp_ = par.ctypes.data_as(C.c_void_p)
            row, outs = [], []
            for env in ({"PM_FK_WIDE": "0", "PM_FK_STREAM": "0"}, {"PM_FK_WIDE": "0"}, {"PM_FK_WIDE": "1", "PM_FK_STREAM": "0"}, {}):
                for k in ("PM_FK_WIDE", "PM_FK_STREAM"): os.environ.pop(k, None)
                os.environ.update(env)
                ms, _ = pp.timeit(lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp_, F, J, P(pos), P(rm), None))
                name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
                row.append(f"{ms * 1e3:7.1f} us {F * (64 * J + 12) / ms / 1e6 / 80:5.1f}% {name[:30]:30s}")
                outs.append((pos.clone(), rm.clone()))
            same = bool(torch.equal(outs[0][0].view(torch.int32), outs[2][0].view(torch.int32)) and torch.equal(outs[0][1].view(torch.int32), outs[2][1].view(torch.int32)))
            print(f"J={J:3d} {kind:5s} depth {depth:3d}: " + " | ".join(row) + f" | bits equal: {same}", flush=True)
            del rot, pos, rm, outs


if __name__ == "__main__":
    main()
