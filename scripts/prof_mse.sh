#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of ONE cold mse sweep (no hist timing to speak of).
# Usage: scripts/prof_mse.sh <tag> [extra bench args]; environment (DPL_*) passes through.
set -u
TAG=${1:-mse}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 1 --mse-jitter "" "$@" > $OUT/bench.json 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/stats
head -24 $OUT/kernel_stats.md
