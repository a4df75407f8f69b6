#!/bin/bash
# same-box A/B of perf_probe rows: libpmhip_ab.so (tools/ab_build.sh <ref> <file.hip>) against the production library
#   tools/ab_probe.sh dq [rounds]
only=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  echo "== ab";   PMHIP_VARIANT=ab python tools/perf_probe.py --only "$only" --sustained 100 2>&1 | grep -v amdgpu.ids
  echo "== prod"; python tools/perf_probe.py --only "$only" --sustained 100 2>&1 | grep -v amdgpu.ids
done
