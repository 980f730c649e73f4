# GPU box: the end-to-end CLI (ResNet-50, N = 1024 .bin files), round 5: host shape rules vs the batch-1 device forward
# (DPL_INFER_DEVICE=1), by --calib_batch, -A hist and -A mse, alternating.  Prints the host-wall split of every run.
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
for rep in 1 2; do for A in hist mse; do for V in "0 32" "1 32" "0 64" "0 128"; do
set -- $V
DPL_INFER_DEVICE=$1 python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --calib_batch $2 --skip_profiling -O /tmp/e2e/out_$A$2 --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
print('$A infer_device=$1 batch $2', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(1024/t['tensor_calibration_wall_s']), 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t.get('forward_first_batch_gpu_s',0),4), 'stat', round(t['statistics_gpu_s'],4), 'steady', round(t.get('forward_steady_images_per_s',0)), 'load', round(t['load_model_wall_s'],3), t['host_wall'], {k: round(v,3) for k,v in t['startup'].items() if isinstance(v,float)})
PY
done; done; done
