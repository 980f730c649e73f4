# GPU box: the end-to-end CLI (`-A mse` N = 4096 and `-A hist` N = 1024, ResNet-50) under MIOpen's default find mode on a WARM user find-db
# (DPL_MIOPEN_FIND_MODE=library; an untimed run first fills the db) against FAST (this package's default), alternating, fresh processes.
python scripts/e2e_setup.py /tmp/e2e 4096 2>&1 | tail -1
DPL_MIOPEN_FIND_MODE=library python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 128 -A minmax -D trt --skip_profiling -O /tmp/e2e/out_w > /dev/null 2>&1
for rep in 1 2 3; do for M in library FAST; do for A in "mse 4096" "hist 1024"; do
set -- $A
DPL_MIOPEN_FIND_MODE=$M python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N $2 -A $1 -D trt --skip_profiling -O /tmp/e2e/out_$M --timing_json /tmp/t_$M.json > /tmp/cli_$M.log 2>&1 || tail -5 /tmp/cli_$M.log
python - <<PY
import json
t=json.load(open('/tmp/t_$M.json'))
print('%-8s -A $1 N=$2' % '$M', 'first batch', round(t.get('forward_first_batch_gpu_s',0),3), 'steady img/s', round(t.get('forward_steady_images_per_s',0)), 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round($2/t['tensor_calibration_wall_s']))
PY
done; done; done
