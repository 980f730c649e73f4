#!/bin/bash
# GPU box: kernel-trace stats of scripts/mse_run.py.  Usage: scripts/prof_run.sh <tag> <mse_run args...>
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 scripts/mse_run.py "$@" > $OUT/run.txt 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md > /dev/null
rm -rf $OUT/stats
cat $OUT/run.txt; grep -h "octav" $OUT/kernel_stats.md | cut -c1-60,92-200
