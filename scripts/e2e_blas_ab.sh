# GPU: does the first GEMM's library load (hipBLASLt: 0.2 s on a helper thread) slow the rest of the start-up down?  A/B: ResNet-50's
# one Gemm on ops.gemm_small and no BLAS warm-up thread (the default) / through the library (DPL_GEMM_SMALL=0).
# (Also tried here: hipInit on a helper thread beside `import torch` — `import torch` got slower by what hipInit took: no gain, removed.)
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1   # (page cache, VRAM)
run() {
sleep 2
T0=$EPOCHREALTIME
env $1 python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $2 -D trt --skip_profiling -O /tmp/e2e/out_h --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
T1=$EPOCHREALTIME
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
h=t['host_wall']
tl=t['timeline_s']
print('$2 $1', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(1024/t['tensor_calibration_wall_s']), 'first', round(t['forward_first_batch_gpu_s'],3), 'wait_convs', h.get('warm_wait_convs_s'), 'wait_blas', h.get('warm_wait_blas_s'), 'loop', round(h['pass1_loop_s'],3), 'kernels_end', tl.get('warm:kernels:end'), 'blas_end', tl.get('warm:blas:end'), 'convs_end', tl.get('warm:convs:end'), 'ff', tl.get('first_forward:start'), tl.get('first_forward:issued'), 'starts', tl.get('main:calibration_starts'), 'done', tl.get('main:calibration_done'), 'process', round($T1 - $T0, 3), 'imports', round(t['startup']['interpreter_and_imports_s'], 3), 'dev', tl.get('main:group_and_device'))
PY
}
for rep in 1 2 3 4; do
for A in hist mse; do
run "X=1" $A
run "DPL_GEMM_SMALL=0" $A
done; done
