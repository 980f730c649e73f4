#!/bin/bash
# GPU box: sampling-rate variants of k_octav_probe on one box (every pair on its own sample, +-10 % jitter)
for V in "$@"; do
  DPL_HIPCC_EXTRA="$V" python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  echo "variant [$V]"
  for r in 1 2; do DPL_BENCH_JITTER=0.1 timeout 200 python3 scripts/mse_run.py resnet50 128 | cut -c1-130; done
done
python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1
