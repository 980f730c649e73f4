export TMPDIR=/tmp
for w in resnet50 resnet50_real vit; do
  for j in 0 0.3; do
    [ $w = vit ] && [ $j = 0.3 ] && continue
    DPL_BENCH_JITTER=$j python scripts/mse_run.py $w 64 17 2>&1 | tail -1
  done
done
DPL_SINGLE=1 python scripts/mse_run.py resnet50 64 17 2>&1 | tail -1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o t -- python3 $GRAFT_REPO_ROOT/scripts/mse_run.py resnet50 64 17 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/summarize_prof.py stats /tmp/prof1 gpurun_out/tail_stats.md | head -12
