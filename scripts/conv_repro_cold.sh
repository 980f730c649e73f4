#!/bin/bash
# Does MIOpen's naive convolution ever run in a user's process?  A kernel trace of scripts/conv_repro_probe.py with an EMPTY user
# find-db (MIOPEN_USER_DB_PATH -> a fresh directory; what every new account / box starts with), then the same command again with
# the db the first run wrote.  scripts/conv_repro_kernels.py prints the kernels per (configuration, call).
# (the library's own find mode: this package defaults to FAST, which never benchmarks and never launches the naive kernel)
export DPL_MIOPEN_FIND_MODE=library
out=$PWD/gpurun_out/conv_repro; mkdir -p $out
R=$PWD
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=/tmp/miopen_cold_db_$$
rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
cd /tmp
for pass in cold warm; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$pass -o t -- python3 $R/scripts/conv_repro_probe.py --net resnet50 --batch 16 --calls 3 --forwards 2 --out $out/$pass.json > $out/$pass.log 2>&1
  f=$(find $out/trace_$pass -name '*kernel_trace.csv' | head -1)
  python3 $R/scripts/conv_repro_kernels.py $f $out/$pass.json > $out/kernels_$pass.txt 2>&1
  rm -rf $out/trace_$pass
  { echo "== user find-db after the $pass pass"; ls -la $MIOPEN_USER_DB_PATH; } >> $out/cold_db.txt
done
unset MIOPEN_USER_DB_PATH
cd $R
# ViT-B/16: one convolution (the patch embedding), everything else hipBLASLt GEMMs — do two forwards agree to the last bit?
python3 scripts/conv_repro_probe.py --net vit_b16 --batch 4 --calls 3 --forwards 3 --quant --out $out/vit_default_b4.json > $out/vit_default_b4.log 2>&1
python3 scripts/conv_repro_probe.py --net vit_b16 --batch 4 --calls 3 --forwards 3 --quant --det --out $out/vit_det_b4.json > $out/vit_det_b4.log 2>&1
grep -c naive $out/kernels_cold.txt $out/kernels_warm.txt; tail -2 $out/vit_default_b4.log $out/vit_det_b4.log
