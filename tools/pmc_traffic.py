#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) over `tools/perf_probe.py --only fk,ceiling,dq,o6d`
into the per-launch HBM traffic summary that bench.py reads from profiles/.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [round] > profiles/rNN_fk_hbm_traffic.json

Corrections (same guide): FETCH_SIZE is in KiB and under-reports by 1/2 on gfx950 for these streams (x2,
checked here on pm::ceiling_kernel whose bytes are known exactly); WRITE_SIZE is in KiB and exact."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

F22, F52 = 1 << 20, 1 << 18  # the probe's frame counts for J = 22 / 52

# kernel-name substring, grid size (workgroups x 64 lanes) -> (label, algorithmic bytes per launch)
def _grid(frames, fpw, nt=1):
    tiles = -(-frames // fpw)
    groups = -(-tiles // nt)
    return -(-groups // 8) * 8 * 64


KNOWN = [
    ("fk_kernel<16", _grid(F22, 16), "fk_J22", F22 * (64 * 22 + 12)),
    ("ceiling_kernel", None, "ceiling", None),
    ("to_root_dq_kernel<16", _grid(F22, 16), "to_root_dq_J22", F22 * (48 * 22 + 12)),
    ("to_root_dq_sched_kernel<2", _grid(F22, 8, 2), "to_root_dq_J22", F22 * (48 * 22 + 12)),
    ("to_root_dq_wide_kernel<8, 3", _grid(F22, 8), "to_root_dq_J22", F22 * (48 * 22 + 12)),  # round 6: the step-list kernel (dqwide.hip), one tile a workgroup
    ("gather_parent_kernel<0", _grid(F22, 8), "from_root_dq_J22", F22 * 60 * 22),
    ("fk_pipe_kernel<4, 4, true, 0, false, false, false", _grid(F52, 4, 2), "fk_J52", F52 * (64 * 52 + 12)),
    ("fk_pipe_kernel<4, 4, true, 1, false, false, false", _grid(F52, 4, 2), "fk_from_ortho6d_J52", F52 * (72 * 52 + 12)),
    ("to_root_dq_kernel<8", _grid(F52, 8), "to_root_dq_J52", F52 * (48 * 52 + 12)),
    ("to_root_dq_sched_kernel<2", _grid(F52, 8, 2), "to_root_dq_J52", F52 * (48 * 52 + 12)),
    ("to_root_dq_wide_kernel<4, 4", _grid(F52, 4), "to_root_dq_J52", F52 * (48 * 52 + 12)),
    ("gather_parent_kernel<0", _grid(F52, 4), "from_root_dq_J52", F52 * 60 * 52),
]


def read_pass(d, counter):
    rows = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                rows[(r["Kernel_Name"], int(r["Grid_Size"]), int(float(r.get("LDS_Block_Size") or 0)))].append(float(r["Counter_Value"]))  # (LDS size: the copy kernel runs two frame sizes on one grid)
    return {k: (statistics.median(v), len(v), v) for k, v in rows.items()}


def main():
    fetch = read_pass(sys.argv[1], "FETCH_SIZE")
    write = read_pass(sys.argv[2], "WRITE_SIZE")
    rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    out = {"round": rnd,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over "
                  "tools/perf_probe.py --only fk,ceiling,dq,o6d --sustained 20 (median over dispatches); "
                  "bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024",
           "kernels": {}}
    for (name, grid, lds), (fkb, nf, fall) in sorted(fetch.items()):
        wkb, nw, wall = write.get((name, grid, lds), (None, 0, []))
        if wkb is None:
            continue
        if "ceiling_kernel" in name:
            # the known-byte copy kernel (the calibration of the x2) runs both frame sizes, on the same grid since its tile follows
            # fk's: split the dispatches by size instead of taking one median over both
            for frames, joints in ((F22, 22), (F52, 52)):
                fs = [v for v in fall if abs(v * 2048 / (frames * 16 * joints) - 1.0) < 0.1]
                ws = [v for v in wall if abs(v * 1024 / (frames * 48 * joints) - 1.0) < 0.1]
                if fs and ws:
                    rd, wr, algo = statistics.median(fs) * 2048, statistics.median(ws) * 1024, frames * 64 * joints
                    out["kernels"][f"ceiling_J{joints}"] = {"kernel": name, "grid": grid, "dispatches": [len(fs), len(ws)],
                                                           "FETCH_SIZE_KB": statistics.median(fs), "WRITE_SIZE_KB": statistics.median(ws),
                                                           "read_bytes_corrected": rd, "write_bytes": wr, "total": rd + wr, "algorithmic": algo,
                                                           "traffic_over_algorithmic": (rd + wr) / algo}
            continue
        label, algo = None, None
        for sub, g, lab, ab in KNOWN:
            if sub.replace(" ", "") in name.replace(" ", "") and (g is None or g == grid):
                label, algo = lab, ab
                break
        if label is None:
            continue
        rd, wr = fkb * 1024 * 2, wkb * 1024
        if label == "ceiling":  # known-byte copy kernel (the calibration of the x2): 16 J read + 48 J written per frame
            for frames, joints in ((F22, 22), (F52, 52)):
                if abs((rd + wr) / (frames * 64 * joints) - 1.0) < 0.1:
                    label, algo = f"ceiling_J{joints}", frames * 64 * joints
        e = {"kernel": name, "grid": grid, "dispatches": [nf, nw], "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
             "read_bytes_corrected": rd, "write_bytes": wr, "total": rd + wr}
        if algo:
            e["algorithmic"] = algo
            e["traffic_over_algorithmic"] = (rd + wr) / algo
        out["kernels"][label] = e
    fk = out["kernels"].get("fk_J22")
    if fk:  # the fields bench.py reads
        out["kernel"] = fk["kernel"]
        out["workload"] = {"frames": F22, "joints": 22}
        out["corrected_bytes_per_launch"] = {"read": fk["read_bytes_corrected"], "write": fk["write_bytes"], "total": fk["total"]}
        out["algorithmic_bytes_per_launch"] = fk["algorithmic"]
        out["traffic_over_algorithmic"] = fk["traffic_over_algorithmic"]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
