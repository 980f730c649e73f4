#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of ONE cold `-A mse` run (64 batches of the ResNet-50 set through
# ops.OctavPipeline, scripts/mse_run.py: nothing else in the process — no hist sweeps, no e2e children, no other workloads).
# Usage: scripts/prof_mse.sh <tag> [resnet50|resnet50_real|vit]; environment (DPL_*) passes through.
set -u
TAG=${1:-mse}; W=${2:-resnet50}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 scripts/mse_run.py $W 64 17 > $OUT/run.log 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/stats
tail -1 $OUT/run.log; head -14 $OUT/kernel_stats.md
