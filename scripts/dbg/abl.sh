# tuning / ablation timing of the one-read OCTAV kernels (results of DPL_ABL_* builds are NOT valid; only durations are read)
# usage: [PIPE=1] [ENVS="A=1 B=2"] scripts/dbg/abl.sh default "<hipcc flags of variant 1>" ...
# PIPE=0 (default): kernels run back to back (clean per-kernel durations); PIPE=1: the product's two-stream schedule
export TMPDIR=/tmp
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  if [ "$v" != default ]; then bash scripts/build_variant.sh $PWD/gpurun_out/abl_$name.so $v > /dev/null 2>&1 || { echo "build failed: $v"; continue; }; export DPL_LIB=$PWD/gpurun_out/abl_$name.so; else unset DPL_LIB; fi
  rm -rf /tmp/abl; mkdir -p /tmp/abl
  env DPL_OCTAV_PIPELINE=${PIPE:-0} $ENVS rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl/stats -o bench -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 1 > /tmp/abl/bench.json 2> /tmp/abl/err.txt
  echo "== $v  [PIPE=${PIPE:-0} $ENVS]  $(python3 -c "import json;d=json.loads(open('/tmp/abl/bench.json').read().strip().splitlines()[-1]);print('mse img/s %.0f  frac %.4f' % (d['mse']['value'], d['mse']['roofline']['frac']))")"
  python3 scripts/summarize_prof.py stats /tmp/abl/stats /tmp/abl/ks.md | grep "octav_walk\|octav_oneread(\|k_minmax(\|octav_sort" | cut -c1-40,95-150
  rm -f gpurun_out/abl_$name.so
done
