#!/bin/bash
# Tuning helper (GPU box): rebuild with extra -D knobs, run bench.py under rocprofv3 --kernel-trace --stats and
# print the per-kernel averages of the kernels matching a pattern.
# Usage: scripts/variant_prof.sh "<bench args>" "<kernel regex>" "<flags of variant 1>" ...
set -u
ARGS=$1; PAT=$2; shift 2
export TMPDIR=/tmp
for V in "$@"; do
  DPL_HIPCC_EXTRA="$V" python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  rm -rf /tmp/vp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o b -- python3 bench.py --cpu-seconds 0 $ARGS > /tmp/vp.json 2>/dev/null
  echo "== [$V] $(python3 -c "import json;d=json.load(open('/tmp/vp.json'));print('img/s %.0f' % d['value'])")"
  python3 - "$PAT" <<'PY'
import csv, glob, re, sys
for f in glob.glob('/tmp/vp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[1], r['Name']):
            print('   %-28s avg %8.1f us  calls %s' % (re.sub(r'.*::(k_\w+).*', r'\1', r['Name'])[:28], float(r['AverageNs'])/1e3, r['Calls']))
PY
done
python3 -m dipoorlet_amd.csrc.build --force > /dev/null 2>&1
