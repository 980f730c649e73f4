#!/bin/bash
# GPU box: same-box A/B of library variants over the mse workloads (boxes of the pool differ by a few per cent: only runs on ONE
# box compare).  Usage: scripts/ab_run.sh "<flags A>" "<flags B>" ...   (flags as for scripts/build_variant.sh; "" = committed)
VARIANTS=("$@")
for rep in 1 2; do
for V in "${VARIANTS[@]}"; do
  bash scripts/build_variant.sh /tmp/libv.so $V > /tmp/build.log 2>&1 || { echo "build failed: $V"; tail -3 /tmp/build.log; continue; }
  r=""
  for cfg in "resnet50 0 64" "resnet50 0.1 64" "resnet50_real 0.3 64" "vit 0 32"; do
    set -- $cfg
    ms=$(DPL_LIB=/tmp/libv.so DPL_BENCH_JITTER=$2 python3 scripts/mse_run.py $1 $3 17 2>&1 | tail -1 | sed 's/.*: \([0-9.]*\) ms\/batch.*/\1/')
    r="$r $1@$2=$ms"
  done
  echo "[$V]$r"
done
done
