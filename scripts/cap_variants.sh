#!/bin/bash
# GPU box: slice-cap variants of the one-read OCTAV form, cold mse runs on one box
for V in "$@"; do
  DPL_HIPCC_EXTRA="$V" python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  echo "variant [$V]"
  for r in 1 2; do timeout 200 python3 scripts/mse_run.py resnet50 128 | cut -c1-100; done
  DPL_BENCH_JITTER=0.1 timeout 200 python3 scripts/mse_run.py resnet50 128 | cut -c1-100
  timeout 200 python3 scripts/mse_run.py vit 32 9 | cut -c1-100
done
python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1
