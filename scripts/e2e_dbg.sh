python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1   # (page cache, VRAM)
run() {
sleep 2
env $1 python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $2 -D trt --skip_profiling -O /tmp/e2e/out_h --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
h=t['host_wall']
tl=t['timeline_s']
print('$2 $1', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(1024/t['tensor_calibration_wall_s']), 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t['forward_first_batch_gpu_s'],3), 'steady', round(t.get('forward_steady_images_per_s',0)), 'wait_convs', h.get('warm_wait_convs_s'), 'wait_blas', h.get('warm_wait_blas_s'), 'loop', round(h['pass1_loop_s'],3), 'ff', tl.get('first_forward:start'), tl.get('first_forward:issued'), 'done', tl.get('main:calibration_done'))
PY
}
for rep in 1 2 3; do
for A in hist mse; do
run "DPL_PREWARM_WAIT=1" $A
run "DPL_PREWARM_WAIT=0" $A
done; done
