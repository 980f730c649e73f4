# A/B on one box: the library in the tree against scripts/dbg/base.so (a build of another revision), product schedule
# usage: scripts/dbg/ab.sh [repeats]
export TMPDIR=/tmp
for i in $(seq 1 ${1:-2}); do
  for v in tree base; do
    if [ $v = base ]; then export DPL_LIB=$PWD/scripts/dbg/base.so; else unset DPL_LIB; fi
    python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 2 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'mse img/s %.0f  frac %.4f' % (d['mse']['value'], d['mse']['roofline']['frac']))"
  done
done
