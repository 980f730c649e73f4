#!/bin/bash
# Tuning helper (GPU box): rebuild the library with extra -D knobs and run bench.py on each variant.
# Usage: scripts/variant_bench.sh "<bench args>" "<flags of variant 1>" "<flags of variant 2>" ...
set -u
ARGS=$1; shift
for V in "$@"; do
  DPL_HIPCC_EXTRA="$V" python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  for r in 1 2; do
    python3 bench.py --cpu-seconds 0 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', '| img/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))"
  done
done
python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1
