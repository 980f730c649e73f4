#!/bin/bash
# VERDICT r05 item 2 on one fresh box: what differs between two forwards, and does the naive convolution run?
# (the library's own find mode: this package defaults to FAST, which never benchmarks and never launches the naive kernel)
export DPL_MIOPEN_FIND_MODE=library
out=gpurun_out/conv_repro; mkdir -p $out
{ echo "== MIOpen user db / cache before"; ls -la ~/.cache/miopen ~/.config/miopen 2>&1 | head; } > $out/cache.txt
python3 scripts/conv_repro_probe.py --net resnet50 --batch 32 --quant --out $out/default_b32.json > $out/default_b32.log 2>&1
{ echo "== after the first process"; find ~/.cache/miopen ~/.config/miopen -type f 2>/dev/null | head -20; du -sh ~/.cache/miopen ~/.config/miopen 2>&1; } >> $out/cache.txt
python3 scripts/conv_repro_probe.py --net resnet50 --batch 32 --quant --det --out $out/det_b32.json > $out/det_b32.log 2>&1
python3 scripts/conv_repro_probe.py --net resnet18 --image 64 --batch 4 --quant --out $out/r18_default_b4.json > $out/r18_default_b4.log 2>&1
python3 scripts/conv_repro_probe.py --net resnet18 --image 64 --batch 4 --quant --det --out $out/r18_det_b4.json > $out/r18_det_b4.log 2>&1
python3 scripts/conv_repro_probe.py --net resnet50 --batch 64 --calls 3 --forwards 3 --out $out/default_b64.json > $out/default_b64.log 2>&1
MIOPEN_LOG_LEVEL=5 MIOPEN_ENABLE_LOGGING=1 python3 scripts/conv_repro_probe.py --net resnet18 --image 64 --batch 4 --calls 2 --forwards 2 > $out/r18_miopen_log.txt 2>&1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace_default -o t -- python3 $R/scripts/conv_repro_probe.py --net resnet50 --batch 32 --calls 3 --forwards 3 --out $R/$out/trace_default.json > $R/$out/trace_default.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$out/trace_det -o t -- python3 $R/scripts/conv_repro_probe.py --net resnet50 --batch 32 --calls 3 --forwards 3 --det --out $R/$out/trace_det.json > $R/$out/trace_det.log 2>&1
cd $R
for t in default det; do
  f=$(find $out/trace_$t -name '*kernel_trace.csv' | head -1)
  python3 scripts/conv_repro_kernels.py $f $out/trace_$t.json > $out/kernels_$t.txt 2>&1
  rm -rf $out/trace_$t
done
tail -3 $out/*.log
