#!/bin/bash
# GPU: KERNEL durations (rocprofv3 --kernel-trace --stats) of k_fake_quant inside the fake-quantised ResNet-50 forward at batch 64 —
# what bench.py's fake_quant.product_forward measures with HIP events around every node (whose own packets are in that figure).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_fqfwd
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o run -- python3 scripts/fq_forward_run.py > $OUT/run.log 2> $OUT/run.err
python3 scripts/summarize_prof.py stats $OUT/s $OUT/kernel_stats_fq_forward.md > /dev/null
tail -1 $OUT/run.log
python3 - <<PY
import re
rows=[l for l in open("$OUT/kernel_stats_fq_forward.md") if "k_fake_quant(" in l]
log=open("$OUT/run.log").read()
m=re.search(r"forwards (\d+), Q/DQ nodes per forward (\d+), bytes per forward (\d+)",log)
nf,nodes,nbytes=map(int,m.groups())
c=[x.strip() for x in rows[0].split("|")]
calls,total=int(c[2]),int(c[3])
# 13 forwards ran the kernel (3 warm-up + 10 counted) + the clip pass has none: calls = 13 x nodes
per_fwd_ns=total/(calls/nodes)
print("k_fake_quant: %d calls, %.1f us per forward over %d nodes = %.2f us per node; %.3f GB per forward -> %.0f GB/s = %.3f of 8 TB/s"%(calls,per_fwd_ns/1e3,nodes,per_fwd_ns/1e3/nodes,nbytes/1e9,nbytes/per_fwd_ns,nbytes/per_fwd_ns/8000))
import json, sys
sys.path.insert(0, "scripts")
from summarize_prof import source_sha
json.dump({"source_sha": source_sha(), "kernel": "k_fake_quant", "calls": calls, "nodes_per_forward": nodes, "bytes_per_forward": nbytes,
           "us_per_forward": per_fwd_ns / 1e3, "frac_of_8TBps": nbytes / per_fwd_ns / 8000,
           "how": "rocprofv3 --kernel-trace --stats over scripts/fq_forward_run.py: total kernel duration / forwards"}, open("$OUT/fq_forward.json", "w"), indent=1)
PY
rm -rf $OUT/s
