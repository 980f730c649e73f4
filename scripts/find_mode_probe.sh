#!/bin/bash
# GPU box: what MIOpen's find mode costs and buys on an EMPTY user find-db (a fresh account / container): the library's default
# (DYNAMIC_HYBRID: a miss benchmarks every applicable solver once, its naive kernel among them; DPL_MIOPEN_FIND_MODE=library) against
# FAST (this package's default, executor.py: a miss takes the library's heuristic choice, nothing is benchmarked).  Per mode: a cold
# process, then a second one on the db the first wrote; then other networks / batch sizes, cold.   bash scripts/find_mode_probe.sh
out=$PWD/gpurun_out/find_mode; mkdir -p $out
unset MIOPEN_FIND_MODE
for mode in library FAST; do
  export MIOPEN_USER_DB_PATH=/tmp/miopen_fm_${mode}_$$; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
  export DPL_MIOPEN_FIND_MODE=$mode
  for pass in cold warm; do
    python3 scripts/conv_repro_probe.py --net resnet50 --batch 64 --calls 4 --forwards 6 --out $out/${mode}_$pass.json > $out/${mode}_$pass.log 2>&1
  done
done
python3 - <<PY
import json
for mode in ("library", "FAST"):
    for p in ("cold", "warm"):
        r = json.load(open("$out/%s_%s.json" % (mode, p)))
        first = sum(a["gpu_ms_host_ms"][0][1] for a in r["part_a"])
        steady = sum(min(m[0] for m in a["gpu_ms_host_ms"][1:]) for a in r["part_a"])
        fw = r["part_b_fp"]["forward_ms"]
        print(f"resnet50_64 {mode:8s} {p}: first calls of the 23 configurations {first:7.0f} ms host; their steady GPU time {steady:.3f} ms; forwards {fw}")
PY
for net in "resnet18 --image 224 --batch 64" "vit_b16 --batch 16" "resnet50 --batch 16"; do
for mode in library FAST; do
  export MIOPEN_USER_DB_PATH=/tmp/miopen_fm2_${mode}_$$; rm -rf $MIOPEN_USER_DB_PATH; mkdir -p $MIOPEN_USER_DB_PATH
  export DPL_MIOPEN_FIND_MODE=$mode
  tag=$(echo $net | cut -d" " -f1)_$(echo $net | awk '{print $NF}')
  python3 scripts/conv_repro_probe.py --net $net --calls 4 --forwards 6 --out $out/${tag}_${mode}.json > $out/${tag}_${mode}.log 2>&1
  python3 - <<PY
import json
r = json.load(open("$out/${tag}_${mode}.json"))
first = sum(a["gpu_ms_host_ms"][0][1] for a in r["part_a"])
steady = sum(min(m[0] for m in a["gpu_ms_host_ms"][1:]) for a in r["part_a"])
print("%-12s %-8s cold: first calls of the %d configurations %7.0f ms host; their steady GPU time %.3f ms; forwards %s" % ("$tag", "$mode", len(r["part_a"]), first, steady, r["part_b_fp"]["forward_ms"]))
PY
done; done
unset DPL_MIOPEN_FIND_MODE MIOPEN_USER_DB_PATH
