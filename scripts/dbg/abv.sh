# A/B/C... on one box under the product schedule (no profiler): variants = hipcc flag sets; prints mse img/s, roofline fraction
# and the number of pairs that needed the compaction route in the last batches.   usage: scripts/dbg/abv.sh <repeats> default "<flags>" ...
export TMPDIR=/tmp
R=$1; shift
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  [ "$v" = default ] || bash scripts/build_variant.sh $PWD/gpurun_out/abv_$name.so $v > /dev/null 2>&1 || echo "build failed: $v"
done
for i in $(seq 1 $R); do
  for v in "$@"; do
    name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
    if [ "$v" != default ]; then export DPL_LIB=$PWD/gpurun_out/abv_$name.so; else unset DPL_LIB; fi
    env $ENVS python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 2 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('%-40s' % '$v', 'mse img/s %.0f  frac %.4f  checksum %.6f  missed %s/%s batches, %s pairs' % (d['mse']['value'], d['mse']['roofline']['frac'], d['mse']['clip_checksum'], d['mse']['prediction']['batches_with_a_miss'], d['mse']['prediction']['batches'], d['mse']['prediction']['pairs_missed']))"
  done
done
rm -f gpurun_out/abv_*.so
