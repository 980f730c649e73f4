#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench line, then PMC passes (HBM traffic), each in its
# own rocprofv3 run (never --pmc together with a trace domain).  Usage: scripts/profile_gpu.sh <tag>
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --cpu-seconds 0 --e2e-images 0 --vit-images 0 --real-images 0 --mse-jitter "" > $OUT/bench_stats.json 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --cpu-seconds 0 --e2e-images 0 --vit-images 0 --real-images 0 --mse-jitter "" --fq-reps 0 --steps 2 --warmup 1 --mse-steps 1 > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
  python3 scripts/summarize_prof.py pmc $OUT/pmc_$C $C $OUT/pmc_$C.json > /dev/null
done
python3 scripts/summarize_prof.py traffic $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json $OUT/traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -14 $OUT/kernel_stats.md; cat $OUT/traffic.json; tail -c 1500 $OUT/bench_stats.json
