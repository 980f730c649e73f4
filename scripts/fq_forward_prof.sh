#!/bin/bash
# GPU: KERNEL durations (rocprofv3 --kernel-trace) of the fake-quantised ResNet-50 forward at batch 64 in its two forms — every tensor
# exposed (A: ReLU, Add, Q/DQ separate launches) and only the output asked for (B: ReLU / Add + ReLU inside k_fake_quant<PRE>) —
# scripts/fq_forward_run.py; scripts/fq_forward_summary.py cuts the trace at the delimiters.  A warm-up process first: the box's
# MIOpen user find-db is empty, and the library's Find benchmark (its naive kernels among them) would sit in the trace.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_fqfwd
R=$PWD
rm -rf $OUT; mkdir -p $OUT
python3 scripts/fq_forward_run.py > $OUT/warm.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/s -o run -- python3 $R/scripts/fq_forward_run.py > $OUT/run.log 2> $OUT/run.err
cd $R
python3 scripts/fq_forward_summary.py $OUT/s $OUT/run.log $OUT
rm -rf $OUT/s
