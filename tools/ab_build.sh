#!/bin/bash
# A/B library for same-box comparisons: pymotion_amd/libpmhip_ab.so = the current production objects, with the listed
# translation units taken from another git ref.  Run the probes with PMHIP_VARIANT=ab (tools only; never the product).
#   tools/ab_build.sh HEAD~1 dq.hip [fk.hip ...]        (AB_FLAGS="" builds them without -fno-slp-vectorize)
set -e
ref=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/pm_ab.XXXXXX)
mkdir -p "$tmp/pymotion_amd/csrc" "$tmp/include"
git -C "$root" archive "$ref" pymotion_amd/csrc include | tar -x -C "$tmp"; [ -f "$tmp/pymotion_amd/csrc/deep.hip" ] || cp "$root/pymotion_amd/csrc/deep.hip" "$tmp/pymotion_amd/csrc/"; [ -f "$tmp/pymotion_amd/csrc/fkwide.hip" ] || cp "$root/pymotion_amd/csrc/fkwide.hip" "$tmp/pymotion_amd/csrc/"
objs=""
for f in fk fkwide dq deep mirror elementwise unroll ik interp probe host; do
  if [[ " $* " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC ${AB_FLAGS--fno-slp-vectorize} --offload-arch=gfx950 -c "$tmp/pymotion_amd/csrc/$f.hip" -o "$tmp/$f.o"
    objs="$objs $tmp/$f.o"
  else
    objs="$objs $root/pymotion_amd/csrc/build/prod/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/pymotion_amd/libpmhip_ab.so" $objs
rm -rf "$tmp"
echo "built pymotion_amd/libpmhip_ab.so with $* from $ref"
