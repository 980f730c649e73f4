#!/bin/bash
# Run on the GPU box (via gpurun): every number of the bench line from a committed duration.
#   (i)   the default bench line: kernel-trace stats, then PMC passes (HBM traffic: FETCH_SIZE, WRITE_SIZE), each in its own
#         rocprofv3 run (never --pmc together with a trace domain)
#   (ii)  the +-10 % contrast-jitter mse sweep   (scripts/mse_run.py resnet50, DPL_BENCH_JITTER=0.1)
#   (iii) the ViT-B/16 mse sweep                 (scripts/mse_run.py vit)
#   (iv)  the images-alike mse sweep with the streaming kernels NOT overlapped (DPL_OCTAV_LANES=1): the kernel's own duration
# each of (ii), (iii): kernel stats + the two PMC passes -> traffic_<name>.json.  Usage: scripts/profile_gpu.sh <tag>
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
# counters for this library's kernels only (collecting them for every torch kernel made a torch.rand launch of the jittered set crash
# inside the profiler)
KERNELS="k_octav|k_abs_hist|k_minmax|k_hist|k_fake_quant"
# A warm-up process first: the MIOpen user find-db of a fresh box is empty, and the library's Find benchmark over every convolution
# configuration of the fake-quantised forward (8 launches of its naive kernel each, profiles/r06/conv_repro.md) would sit in the trace
python3 scripts/fq_forward_run.py > $OUT/warm.log 2>&1
python3 bench.py --cpu-seconds 0 --e2e-images 0 --vit-images 0 --real-images 0 --big-images 0 --mse-jitter "" --steps 1 --warmup 0 --mse-steps 0 --fq-reps 1 >> $OUT/warm.log 2>&1
python3 scripts/mse_run.py vit 2 2 >> $OUT/warm.log 2>&1
BENCH="python3 bench.py --cpu-seconds 0 --e2e-images 0 --vit-images 0 --real-images 0 --big-images 0 --mse-jitter"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH "" > $OUT/bench_stats.json 2> $OUT/stats.err
python3 scripts/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.md > /dev/null
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/pmc_$C -o bench -- $BENCH "" --fq-reps 0 --steps 2 --warmup 1 --mse-steps 1 > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
  python3 scripts/summarize_prof.py pmc $OUT/pmc_$C $C $OUT/pmc_$C.json > /dev/null
done
python3 scripts/summarize_prof.py traffic $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json $OUT/traffic.json
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
sweep() {   # name, workload, batches, env...
  N=$1; W=$2; NB=$3; shift 3
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_$N -o run -- python3 scripts/mse_run.py $W $NB 17 > $OUT/run_$N.log 2> $OUT/s_$N.err
  python3 scripts/summarize_prof.py stats $OUT/s_$N $OUT/kernel_stats_$N.md > /dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    env "$@" rocprofv3 --pmc $C --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/p_${N}_$C -o run -- python3 scripts/mse_run.py $W $NB 17 > /dev/null 2> $OUT/p_${N}_$C.err
    python3 scripts/summarize_prof.py pmc $OUT/p_${N}_$C $C $OUT/pmc_${N}_$C.json > /dev/null
  done
  python3 scripts/summarize_prof.py traffic $OUT/pmc_${N}_FETCH_SIZE.json $OUT/pmc_${N}_WRITE_SIZE.json $OUT/traffic_$N.json
  rm -rf $OUT/s_$N $OUT/p_${N}_FETCH_SIZE $OUT/p_${N}_WRITE_SIZE $OUT/*.err
}
sweep jitter0.1 resnet50 64 DPL_BENCH_JITTER=0.1
# the streaming kernel ALONE on the chip (DPL_OCTAV_LANES=1: every batch on the caller's stream, as in round 3): its own duration —
# by default the kernels of consecutive batches overlap (two lane streams), their elapsed times then add up to more than the sweep
sweep alike_lanes1 resnet50 64 DPL_OCTAV_LANES=1
sweep vit vit 32 X=1
# (v) the fake-quantised ResNet-50 forward, every tensor exposed against ReLU / Add + ReLU fused into the Q/DQ kernel
bash scripts/fq_forward_prof.sh > $OUT/run_fq_forward.log 2>&1
cp gpurun_out/prof_fqfwd/kernel_stats_fq_forward.md gpurun_out/prof_fqfwd/fq_forward.json $OUT/
# (vi) same-box A/B of the OCTAV side stream's priority: a two-lane sweep, then a one-stream sweep, per setting, twice
for p in -1 0 -1 0; do echo "DPL_OCTAV_SIDE_PRIO=$p"; DPL_OCTAV_SIDE_PRIO=$p python3 scripts/lanes1_after_lanes2.py l2_l1; done > $OUT/ab_side_prio.txt 2>&1
head -12 $OUT/kernel_stats.md; cat $OUT/traffic.json | head -40; tail -c 1200 $OUT/bench_stats.json; tail -1 $OUT/run_jitter0.1.log $OUT/run_vit.log
