#!/usr/bin/env python3
"""mirror (rotation part, mode 'all' and with a joint mapping): the step-list kernel of mirror.hip (PM_MIRROR_WIDE = frames a wave, tuning build) against what the
dispatch picked before it existed (PM_MIRROR_WIDE=0), same box, same arrays; the last column is the dispatch's own pick; says whether the results agree to the bit.
MW_KINDS=bushy,humanoid,chain,body picks the trees, MW_FPW the candidates, MW_MAP=1 mirrors with a random joint permutation (mode 'symmetry')."""
import ctypes as C, os, sys
os.environ.setdefault("PMHIP_VARIANT", "tuning")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.perf_probe as pp
from tools.fk_wide_sweep import humanoid, chain_like
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
pp.SUSTAINED = int(os.environ.get("MW_SUSTAINED", 30))
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
kinds = (os.environ.get("MW_KINDS") or "bushy").split(",")
fpws = [int(x) for x in (os.environ.get("MW_FPW") or "1,2,4,8").split(",")]


def main():
    for J in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "22,52,64,96,128,200,250").split(",")]:
        for kind in kinds:
            if kind == "body":
                if J not in (22, 52): continue
                par = np.asarray(syn.PARENTS_22 if J == 22 else syn.PARENTS_52, dtype=np.int32)
            else:
                par = chain_like(J) if kind == "chain" else humanoid(J) if kind == "humanoid" else syn.random_parents(J, np.random.default_rng(J)).astype(np.int32)
            depth = int(syn.depth_of(par).max())
            F = int(os.environ.get("MW_F", (1 << 20) if J <= 32 else (1 << 19) if J <= 128 else (1 << 18)))
            rot = torch.randn((F, J, 4), device="cuda")
            out = torch.empty((F, J, 4), device="cuda")
            pp_ = par.ctypes.data_as(C.c_void_p)
            mapping = None
            if os.environ.get("MW_MAP"):
                mapping = np.random.default_rng(J).permutation(J).astype(np.int32); mapping[0], mapping[list(mapping).index(0)] = 0, mapping[0]
            mp_ = mapping.ctypes.data_as(C.c_void_p) if mapping is not None else None
            row, outs = [], []
            for fpw in [0] + fpws + [-1]:
                os.environ.pop("PM_MIRROR_WIDE", None)
                if fpw >= 0: os.environ["PM_MIRROR_WIDE"] = str(fpw)
                out.fill_(float("nan"))
                ms, _ = pp.timeit(lambda: _lib.call("pm_mirror_rotations_f32", P(rot), pp_, mp_, 0, F, J, P(out), None))
                name = _lib.last_kernel_name().replace("void pm::", "").split("(")[0]
                if fpw > 0 and "wide" not in name:
                    row.append(f"fpw {fpw}: declined"); outs.append(None); continue
                tag = {0: "before", -1: "pick"}.get(fpw, "fpw " + str(fpw))
                row.append(f"{tag}: {ms * 1e3:7.1f} us {F * 32 * J / ms / 1e6 / 80:5.1f}% {name[:26] if fpw <= 0 else ''}")
                outs.append(out.clone())
            same = [("-" if x is None else "=" if torch.equal(outs[0].view(torch.int32), x.view(torch.int32)) else f"{(outs[0] - x).abs().max().item():.1e}") for x in outs[1:]]
            print(f"J={J:3d} {kind:8s} depth {depth:3d}: " + " | ".join(row) + " | vs before: " + " ".join(same), flush=True)
            del rot, out


if __name__ == "__main__":
    main()
