python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1   # (page cache, VRAM)
run() {
sleep 2
env $1 python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $3 -D trt --calib_batch $2 --skip_profiling -O /tmp/e2e/out_h --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
h=t['host_wall']
print('$3 $2 $1', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(1024/t['tensor_calibration_wall_s']), 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t['forward_first_batch_gpu_s'],3), 'steady', round(t.get('forward_steady_images_per_s',0)), 'consts', h['session_consts_s'], 'infer', h['session_infer_host_s'], 'ranges', h['weight_ranges_s'], 'loop', round(h['pass1_loop_s'],3), 'results', h['results_to_host_s'], [round(x,1) for x in t['forward_batches_ms'][1:] if x > 11])
PY
}
for rep in 1 2 3; do
for A in hist mse; do
run "A=1" 32 $A
run "DPL_SWITCH_INTERVAL=0.0002" 32 $A
run "A=1" 64 $A
run "DPL_SWITCH_INTERVAL=0.0002" 64 $A
done; done
