#!/bin/bash
# A/B library from a FILE: pymotion_amd/libpmhip_<name>.so = the current production objects with one translation unit compiled from the given source
# (a variant kept outside the tree).  Run the probes with PMHIP_VARIANT=<name> (ab / ab2; tools only, never the product).
#   tools/ab_file.sh ab2 ik /tmp/ik_variant.hip
set -e
name=$1; unit=$2; src=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cp "$src" "$root/pymotion_amd/csrc/_ab_${name}_${unit}.hip"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -c "$root/pymotion_amd/csrc/_ab_${name}_${unit}.hip" -o "/tmp/_ab_${name}_${unit}.o"
rm -f "$root/pymotion_amd/csrc/_ab_${name}_${unit}.hip"
objs=""
for f in fk dq deep mirror elementwise unroll ik interp probe host; do
  if [ "$f" == "$unit" ]; then objs="$objs /tmp/_ab_${name}_${unit}.o"; else objs="$objs $root/pymotion_amd/csrc/build/prod/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/pymotion_amd/libpmhip_$name.so" $objs
echo "built pymotion_amd/libpmhip_$name.so with $unit from $src"
