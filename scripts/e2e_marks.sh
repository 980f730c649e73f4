# GPU: timeline marks of fresh `-A mse` / `-A hist` runs (ResNet-50, N = 1024), by the interpreter's thread switch interval
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1
for rep in 1 2 3; do for SW in ${DPL_SW_LIST:-"" 0.0005 0.05}; do for A in mse hist; do
sleep 2
DPL_SWITCH_INTERVAL=$SW python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --skip_profiling -O /tmp/e2e/out_p --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
tl=t['timeline_s']; s0=tl['main:calibration_starts']
print('$A', 'switch=$SW', 'cal_wall', round(t['tensor_calibration_wall_s'],3), round(1024/t['tensor_calibration_wall_s']), {k: round(v-s0,3) for k,v in tl.items() if v>=s0}, 'kernels_start', round(tl['warm:kernels:start']-s0,3))
PY
done; done; done
