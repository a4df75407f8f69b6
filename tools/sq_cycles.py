#!/usr/bin/env python3
"""Summarise one rocprofv3 PMC pass of SQ wave-cycle counters (collected on its own, no trace domains) into
profiles/rNN_sq_wave_cycles.json: per kernel (name + grid) the medians over dispatches and their fractions of
SQ_WAVE_CYCLES -- where the waves of each kernel spend their life (issuing VALU / LDS, parked on s_waitcnt, ...).

    python tools/sq_cycles.py <dir of the pass> [round] > profiles/rNN_sq_wave_cycles.json"""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

COUNTERS = "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"


def main():
    d = sys.argv[1]
    rnd = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if not r["Kernel_Name"].startswith("void pm::"):
                    continue
                rows[f'{r["Kernel_Name"]} grid={r["Grid_Size"]}'][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"round": rnd,
           "how": f"rocprofv3 --pmc {COUNTERS} -- python tools/perf_probe.py --only fk,dq,mirror,o6d,ik,unroll --sustained 10 "
                  "(own pass, no trace domains); medians over dispatches; fractions are of SQ_WAVE_CYCLES (summed over waves; "
                  "WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)",
           "kernels": {}}
    for k, cs in sorted(rows.items()):
        raw = {c: statistics.median(v) for c, v in cs.items()}
        wc = raw.get("SQ_WAVE_CYCLES") or 0.0
        if not wc:
            continue
        frac = {c.lower().replace("sq_", ""): round(raw[c] / wc, 4) for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                                             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if c in raw}
        out["kernels"][k] = {"dispatches": len(cs.get("SQ_WAVE_CYCLES", [])), "raw": raw, "frac_of_wave_cycles": frac}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
