python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
for rep in 1 2 3; do for V in "0 hist" "1 hist" "0 minmax"; do
set -- $V
DPL_INFER_DEVICE=$1 python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $2 -D trt --calib_batch 32 --skip_profiling -O /tmp/e2e/out_$2 --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
print('$2 infer_device=$1', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'fwd_gpu', round(t['forward_gpu_s'],4), t['forward_batches_ms'])
PY
done; done
python scripts/first_call_probe.py
