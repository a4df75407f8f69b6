#!/bin/bash
# Does a LOWER shader clock make the two fk kernels FASTER (power shifting between the XCDs and the fabric / HBM)?  One process per setting.
#   bash tools/sclk_cap_probe.sh [MHz ...]      (needs root on the box; resets the device afterwards)
R=$(cd "$(dirname "$0")/.." && pwd)
export LP_SECONDS=${LP_SECONDS:-1.5}
echo "## default state"; python $R/tools/levels_probe.py 2 | grep -v "^RAW"
for mhz in ${@:-2100 1900 1700}; do
  echo "## rocm-smi --setperfdeterminism $mhz"; /opt/rocm/bin/rocm-smi --setperfdeterminism $mhz 2>&1 | grep -v "^=\|^$" | head -4
  python $R/tools/levels_probe.py 1 | grep "^process"
done
/opt/rocm/bin/rocm-smi --resetperfdeterminism 2>&1 | grep -v "^=\|^$" | head -3
echo "## after the reset"; python $R/tools/levels_probe.py 1 | grep "^process"
