#!/bin/bash
# SQ counters of the kernels one perf_probe selection launches (own rocprofv3 pass, no trace domains):  tools/pmc_sq_quick.sh ew euler
#   PMC_CMD="python tools/deep_sweep.py 64,65" tools/pmc_sq_quick.sh - deep     (any other command instead of perf_probe)
R=$(pwd); only=$1; match=${2:-pm::}
export PYTHONPATH=$R; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES -d /tmp/pq -o q --output-format csv -- ${PMC_CMD:-python $R/tools/perf_probe.py --only $only --sustained 5} > /tmp/pq.log 2>&1
python - "$match" <<'PY'
import csv,glob,collections,statistics,sys
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        if sys.argv[1] in r["Kernel_Name"]: rows[r["Kernel_Name"][:70]+" grid="+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(rows.items()):
    m={c: statistics.median(v) for c,v in cs.items()}
    wc=m.get("SQ_WAVE_CYCLES",1)
    print(k); print("   insts_valu %.3g  per wave %.0f  valu_active/wave_cycles %.2f  wait_any %.2f  wait_inst %.2f  waves %d" % (m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU"]/max(m["SQ_WAVES"],1), m["SQ_ACTIVE_INST_VALU"]/wc, m["SQ_WAIT_ANY"]/wc, m["SQ_WAIT_INST_ANY"]/wc, m["SQ_WAVES"]))
PY
