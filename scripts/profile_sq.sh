#!/bin/bash
# GPU box: shader-core counters of the calibration kernels (one rocprofv3 --pmc run per counter group, no tracing
# domains).  Usage: scripts/profile_sq.sh <tag> [bench args...]   ->  gpurun_out/sq_<tag>.json
set -u
TAG=${1:-hist}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -f /tmp/sq_rows.jsonl
IFS=';' read -ra GROUPS_ <<< "${SQ_GROUPS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS;SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY}"
for GROUP in "${GROUPS_[@]}"; do
  rm -rf /tmp/sq_run
  rocprofv3 --pmc $GROUP --kernel-include-regex "k_octav|k_abs_hist|k_minmax|k_fake_quant" --output-format csv -d /tmp/sq_run -o b -- python3 bench.py --cpu-seconds 0 --steps 6 --warmup 2 --e2e-images 0 --vit-images 0 --real-images 0 --big-images 0 --mse-jitter "" --mse-steps 1 "$@" > /dev/null 2> /tmp/sq_err.txt || tail -3 /tmp/sq_err.txt
  python3 - <<'PY'
import csv, glob, json, re
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob('/tmp/sq_run/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'.*::(k_\w+).*', r'\1', r['Kernel_Name'])
        if k.startswith('k_'):
            acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
with open('/tmp/sq_rows.jsonl', 'a') as o:
    for (k, c), v in acc.items():
        o.write(json.dumps({"kernel": k, "counter": c, "launches": len(v), "mean": sum(v) / len(v)}) + "\n")
PY
done
python3 - "$OUT/sq_$TAG.json" <<'PY'
import json, sys
from collections import defaultdict
t = defaultdict(dict)
for line in open('/tmp/sq_rows.jsonl'):
    r = json.loads(line)
    t[r["kernel"]][r["counter"]] = round(r["mean"], 1)
    t[r["kernel"]]["launches"] = r["launches"]
for k, v in t.items():
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_bank_conflict_rate"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 4)
    if v.get("SQ_WAVE_CYCLES"):
        v["valu_insts_per_wave_cycle"] = round(v.get("SQ_INSTS_VALU", 0) / v["SQ_WAVE_CYCLES"], 4)
json.dump(t, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in t.items() if k in ("k_abs_hist", "k_minmax", "k_octav_tail", "k_fake_quant_items")}, indent=1))
PY
