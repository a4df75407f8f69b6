#!/usr/bin/env python3
"""to_root_dual_quat: the step-list kernel of dqwide.hip (PM_DQ_WIDE = frames a wave, tuning build) against what the dispatch picked before it existed
(PM_DQ_WIDE=0: the tile / scheduled / lane-per-frame kernels), same box, same arrays; metre-scale and centimetre-scale bones (the precise step); the last
column is the dispatch's own pick through the raw ABI (no scale hint) and -- `hint` -- with the front doors' hint; says whether the results agree to the bit.
DQW_KINDS=bushy,humanoid,chain,body picks the trees (body: the 22-joint body / SMPL-H at J = 22 / 52), DQW_FPW the candidates."""
import ctypes as C, os, sys
os.environ.setdefault("PMHIP_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from tools.fk_wide_sweep import humanoid, chain_like
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = int(os.environ.get("DQW_SUSTAINED", 30))
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
kinds = (os.environ.get("DQW_KINDS") or "bushy").split(",")
fpws = [int(x) for x in (os.environ.get("DQW_FPW") or "1,2,4,8").split(",")]


def main():
    for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "22,52,64,96,128,200,250").split(",")]:
        for kind in kinds:
            if kind == "body":
                if J not in (22, 52): continue
                par = np.asarray(syn.PARENTS_22 if J == 22 else syn.PARENTS_52, dtype=np.int32)
            else:
                par = chain_like(J) if kind == "chain" else humanoid(J) if kind == "humanoid" else syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
            depth = int(syn.depth_of(par).max())
            F = int(os.environ.get("DQW_F", (1 << 20) if J <= 32 else (1 << 19) if J <= 128 else (1 << 18)))
            rot = torch.randn((F, J, 4), device="cuda"); rot /= rot.norm(dim=-1, keepdim=True)
            root = torch.rand((F, 3), device="cuda") * 4 - 2
            off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
            dq = torch.empty((F, J, 8), device="cuda")
            pp_ = par.ctypes.data_as(C.c_void_p)
            for scale, label in ((1.0, "m "), (100.0, "cm")):
                o, r = off * scale, root * scale
                row, outs = [], []
                hint = C.c_float(float(o.abs().max()))
                for fpw in [0] + fpws + [-1, -2]:
                    os.environ.pop("PM_DQ_WIDE", None)
                    if fpw >= 0: os.environ["PM_DQ_WIDE"] = str(fpw)
                    dq.fill_(float("nan"))
                    if fpw == -2: call = lambda: _lib.call("pm_to_root_dq_hint_f32", P(rot), P(r), pp_, P(o), F, J, P(dq), hint, None)  # noqa: E731
                    else: call = lambda: _lib.call("pm_to_root_dq_f32", P(rot), P(r), pp_, P(o), F, J, P(dq), None)  # noqa: E731
                    ms, _ = pp.timeit(call)
                    name = _lib.last_kernel_name().replace("void pm::", "").replace("to_root_dq_", "").split("(")[0]
                    if fpw > 0 and "wide" not in name:
                        row.append(f"fpw {fpw}: declined"); outs.append(None); continue
                    tag = {0: "before", -1: "pick", -2: "hint"}.get(fpw, "fpw " + str(fpw))
                    row.append(f"{tag}: {ms * 1e3:7.1f} us {F * (48 * J + 12) / ms / 1e6 / 80:5.1f}% {name[:28] if fpw <= 0 else ''}")
                    outs.append(dq.clone())
                same = [("-" if x is None else "=" if torch.equal(outs[0].view(torch.int32), x.view(torch.int32)) else f"{(outs[0] - x).abs().max().item():.1e}") for x in outs[1:]]
                print(f"J={J:3d} {kind:8s} depth {depth:3d} {label}: " + " | ".join(row) + " | vs before: " + " ".join(same), flush=True)
            del rot, dq


if __name__ == "__main__":
    main()
