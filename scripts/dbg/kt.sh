export TMPDIR=/tmp
for v in ${KT_VARIANTS:-default}; do
  if [ $v != default ]; then export DPL_LIB=$PWD/scripts/dbg/$v.so; else unset DPL_LIB; fi
  rm -rf gpurun_out/prof_tmp; mkdir -p gpurun_out/prof_tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp/stats -o bench -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 2 > gpurun_out/prof_tmp/bench.json 2> gpurun_out/prof_tmp/err.txt
  echo "== $v"; python3 scripts/summarize_prof.py stats gpurun_out/prof_tmp/stats gpurun_out/prof_tmp/ks.md | grep "octav_walk\|octav_oneread(\|k_abs_hist" | cut -c1-150
  python3 -c "import json;d=json.loads(open('gpurun_out/prof_tmp/bench.json').read().strip().splitlines()[-1]);print(d['mse']['value'], d['mse']['roofline']['frac'])"
done
rm -rf gpurun_out/prof_tmp
