# A/B on one box: this tree against a copy of another revision under scripts/dbg/base_tree (own package, own library)
# usage: [ENVS="..."] scripts/dbg/ab_tree.sh [repeats]
for i in $(seq 1 ${1:-2}); do
  for v in tree base; do
    d=$PWD; [ $v = base ] && d=$PWD/scripts/dbg/base_tree
    (cd $d && env $ENVS python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 2 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.read())['mse'];print('$v', 'mse img/s %.0f  frac %.4f  checksum %.6f' % (d['value'], d['roofline']['frac'], d['clip_checksum']), d.get('prediction'))")
  done
done
