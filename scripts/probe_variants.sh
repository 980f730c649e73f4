#!/bin/bash
# GPU box: k_octav_probe alone (single stream, every pair on its own sample) for a list of -D variants
for V in "$@"; do
  DPL_HIPCC_EXTRA="$V" python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  echo "variant [$V]"; DPL_SINGLE=1 DPL_OCTAV_PREDICT=probe scripts/prof_run.sh pv resnet50 12 4 | grep "probe\|single"
done
python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1
