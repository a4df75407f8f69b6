#!/usr/bin/env python3
"""from_root_positions on a deep any-order table: follow the worst records of tools/ik_win_diag.py down their root paths.
Per joint of the path: children, the angle of the alignment (rest direction against the measured one, in the parent's frame, from
the float64 oracle's own rotations), the roll angles of the further children, the kernel's error, the reference's own movement under
one-ulp input perturbations (48 draws) and the kernel's (12 draws).  A conditioning problem shows err ~ mov ~ sens growing together;
a defect shows a jump of err at one joint with nothing special about its angles.
Usage: python tools/ik_path_diag.py [kind] [F] [frames...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle import c_oracle as co  # noqa: E402
import pymotion_amd.ops.skeleton as sk  # noqa: E402
from pymotion_amd import _lib, synthetic as syn  # noqa: E402
from ik_win_diag import parents_of, err_of  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "win2_511"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    par = parents_of(kind)
    J = len(par)
    rot, root, off, par = syn.fk_workload(F, parents=par, seed=J + F, normalized=True, offset_scale=0.1)
    pos, _ = co.fk(rot.astype(np.float64), np.zeros((F, 3)), off.astype(np.float64), par)
    pos = pos.astype(np.float32)
    got = sk.from_root_positions(pos, par, off)
    print(_lib.last_kernel_name())
    ref = co.from_root_positions(pos.astype(np.float64), par, off.astype(np.float64))
    err = err_of(got, ref)
    frames = [int(x) for x in sys.argv[3:]] or list(np.argsort(-err.max(1))[:3])
    kids = [[] for _ in range(J)]
    for j in range(1, J):
        kids[par[j]].append(j)
    off64 = off.astype(np.float64)
    for f in frames:
        p1 = pos[f:f + 1]
        r1 = ref[f:f + 1]
        g1 = sk.from_root_positions(p1, par, off)
        print(f"== frame {f}: single-frame call equals the batch call bit for bit: {np.array_equal(g1[0], got[f])}")
        sens = np.zeros(J)
        mov = np.zeros(J)
        for k in range(48):
            up = np.random.default_rng(k + 1).random(p1.shape) < 0.5
            p2 = np.nextafter(p1, np.where(up, np.inf, -np.inf).astype(np.float32))
            sens = np.maximum(sens, err_of(co.from_root_positions(p2.astype(np.float64), par, off64), r1)[0])
            if k < 12:
                mov = np.maximum(mov, err_of(sk.from_root_positions(p2, par, off), g1)[0])
        # world rotations of the reference's answer (float64)
        _, G = co.fk(r1, np.zeros((1, 3)), off64, par)
        G = G[0]
        P = p1[0].astype(np.float64)
        j = int(np.argmax(err[f]))
        path = []
        while True:
            path.append(j)
            if j == 0:
                break
            j = par[j]
        path = path[::-1]
        print(f"   worst joint {path[-1]} err {err[f, path[-1]]:.3e}; path of {len(path)} joints")
        print("   joint par nkids  |u|      align_angle(deg)  roll_angles(deg)            err        sens48     mov12")
        for j in path:
            c = kids[j]
            line = f"   {j:4d} {par[j]:4d} {len(c):3d}"
            if c:
                Gp = G[par[j]] if j else np.eye(3)
                u = off64[c[0]]
                d = P[c[0]] - P[j]
                v = Gp.T @ d
                ang = np.degrees(np.arctan2(np.linalg.norm(np.cross(u, v)), u @ v))
                line += f"   {np.linalg.norm(u):.4f}   {ang:9.4f}      "
                rolls = []
                for gc in c[1:]:
                    ug = off64[gc]
                    vg = G[j].T @ (P[gc] - P[j])
                    rolls.append(np.degrees(np.arctan2(np.linalg.norm(np.cross(ug, vg)), ug @ vg)))
                line += " ".join(f"{a:8.3f}" for a in rolls).ljust(28)
            else:
                line += " " * 62
            line += f" {err[f, j]:.3e}  {sens[j]:.3e}  {mov[j]:.3e}"
            print(line)


if __name__ == "__main__":
    main()
