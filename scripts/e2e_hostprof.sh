# GPU: where the HOST spends a fresh `-A mse` (or $1) calibration run of ResNet-50, N = 1024: cProfile of tensor_calibration (DPL_PROFILE_HOST)
# next to the run's own timeline.  bash scripts/e2e_hostprof.sh [mse|hist]
A=${1:-mse}
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1
for rep in 1 2; do
sleep 2
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --skip_profiling -O /tmp/e2e/out_p --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
print(json.dumps({k: t[k] for k in ('host_wall','timeline_s','tensor_calibration_wall_s','forward_gpu_s','statistics_gpu_s','forward_first_batch_gpu_s')}, indent=0))
print([round(x,2) for x in t.get('forward_batches_ms', [])])
PY
done
sleep 2
DPL_PROFILE_HOST=/tmp/prof.txt python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --skip_profiling -O /tmp/e2e/out_p --timing_json /tmp/t.json > /tmp/cli.log 2>&1
head -75 /tmp/prof.txt | cut -c1-200
