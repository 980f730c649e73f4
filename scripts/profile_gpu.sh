#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats, then PMC passes, each in its own rocprofv3 run.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --cpu-seconds 0 "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --cpu-seconds 0 --steps 6 --warmup 2 "$@" > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
  python3 scripts/summarize_prof.py pmc $OUT/pmc_$C $C $OUT/pmc_$C.json
done
# keep only the small summaries (raw traces stay on the box)
find $OUT -name "*.csv" | head -20; cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
