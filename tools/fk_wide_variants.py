#!/usr/bin/env python3
"""fk's other sources on long wide trees -- per-frame offsets, the fused ortho6d source with / without its quaternions, both -- on the wide walk
(fk_wide_kernel, the production dispatch) against what ran before (PM_FK_WIDE=0: the four-frame tiles), same box, same arrays, tuning build.
FKW_FORCE=1: the second column forces the wide walk (joint counts where the dispatch does not pick it); FKW_KIND=humanoid: a body with hands.
Per cent of the 8 TB/s spec on each variant's own algorithmic bytes; whether the two results agree to the bit."""
import ctypes as C, os, sys
os.environ["PMHIP_VARIANT"] = "tuning"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = 30
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def main():
    for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "129,160,200,256,384,512").split(",")]:
        if os.environ.get("FKW_KIND") == "humanoid":
            from tools.fk_wide_sweep import humanoid
            par = humanoid(J)
        else:
            par = syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
        F = int(os.environ.get("FKW_F", 1 << 17))
        rot = torch.randn((F, J, 4), device="cuda"); x6 = torch.randn((F, J, 3, 2), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
        offs = (off[None] * torch.linspace(0.7, 1.3, F, device="cuda")[:, None, None]).contiguous()
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda"); qo = torch.empty((F, J, 4), device="cuda")
        pp_ = par.ctypes.data_as(C.c_void_p)
        variants = [
            ("quat pfo   ", 64 + 12, lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(offs), 1, pp_, F, J, P(pos), P(rm), None)),
            ("o6d        ", 72, lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(off), 0, pp_, F, J, C.c_float(0), P(pos), P(rm), None, None)),
            ("o6d +q     ", 88, lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(off), 0, pp_, F, J, C.c_float(0), P(pos), P(rm), P(qo), None)),
            ("o6d +q pfo ", 100, lambda: _lib.call("pm_fk_from_ortho6d_f32", P(x6), P(root), P(offs), 1, pp_, F, J, C.c_float(0), P(pos), P(rm), P(qo), None)),
        ]
        for label, bpj, fn in variants:
            row, outs = [], []
            for env in ({"PM_FK_WIDE": "0"}, {"PM_FK_WIDE": os.environ["FKW_FORCE"]} if os.environ.get("FKW_FORCE") else {}):
                os.environ.pop("PM_FK_WIDE", None)
                os.environ.update(env)
                qo.zero_()
                ms, _ = pp.timeit(fn)
                name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
                row.append(f"{ms * 1e3:7.1f} us {F * (bpj * J + 12) / ms / 1e6 / 80:5.1f}% {name[:44]:44s}")
                outs.append((pos.clone(), rm.clone(), qo.clone()))
            same = all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(*outs))
            print(f"J={J:3d} {label}: " + " | ".join(row) + f" | bits equal: {same}", flush=True)
        del rot, x6, pos, rm, qo, offs


if __name__ == "__main__":
    main()
