#!/bin/bash
# GPU box: what the exact-tail OCTAV batch time is made of — timing-only builds of the library (results invalid) against the
# committed one, each run through scripts/mse_run.py (cold run of 64 batches of the ResNet-50 set, pipeline and single stream).
# Usage: scripts/tail_ablate.sh [workload] "<flags 1>" "<flags 2>" ...      (flags: -DDPL_TAIL_...; "" = the committed build)
set -u
W=${1:-resnet50}; shift
for V in "$@"; do
  so=/tmp/libdpl_variant.so
  bash scripts/build_variant.sh $so $V > /tmp/build.log 2>&1 || { echo "build failed: $V"; tail -5 /tmp/build.log; continue; }
  a=$(DPL_LIB=$so python3 scripts/mse_run.py $W 64 17 2>&1 | tail -1)
  b=$(DPL_LIB=$so DPL_SINGLE=1 python3 scripts/mse_run.py $W 64 17 2>&1 | tail -1)
  echo "[$V] $a"
  echo "[$V] $b"
done
