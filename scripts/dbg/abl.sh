# ablation timing of k_octav_oneread (results of the ablated builds are NOT valid; only kernel durations are read)
export TMPDIR=/tmp
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  if [ "$v" != default ]; then bash scripts/build_variant.sh $PWD/gpurun_out/abl_$name.so $v > /dev/null 2>&1 || { echo "build failed: $v"; continue; }; export DPL_LIB=$PWD/gpurun_out/abl_$name.so; else unset DPL_LIB; fi
  rm -rf /tmp/abl; mkdir -p /tmp/abl
  DPL_OCTAV_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl/stats -o bench -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 1 > /tmp/abl/bench.json 2> /tmp/abl/err.txt
  echo "== $v"; python3 scripts/summarize_prof.py stats /tmp/abl/stats /tmp/abl/ks.md | grep "octav_walk\|octav_oneread(\|k_abs_hist\|k_minmax(" | cut -c1-40,95-150
  rm -f gpurun_out/abl_$name.so
done
