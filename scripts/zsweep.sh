#!/bin/bash
# GPU box: probe-mode mse sweep at several bracket widths (one cold N=4096 sweep each)
for z in "$@"; do
  DPL_PROBE_Z=$z DPL_OCTAV_PREDICT=probe python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 1 --mse-jitter "" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mse']; p=m['prediction']
print('z=$z frac %.3f batch_ms %.3f miss/batch %.1f share %.4f ok %s' % (m['roofline']['frac'], m['roofline']['avg_batch_ms'], p['pairs_missed']/p['batches'], p['listed_share_of_elements'], m['sample_ok']))"
done
