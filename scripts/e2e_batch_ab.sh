# GPU box: the end-to-end CLI run (ResNet-50, N = 1024 .bin files) by --calib_batch, -A hist and -A mse, alternating.
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -3; ls /tmp/e2e
for rep in 1 2 3; do for A in hist mse; do for CB in 32 64 128; do
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --calib_batch $CB --skip_profiling -O /tmp/e2e/out_$A$CB --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
print('$A batch $CB', 'loop', round(t['host_wall'].get('pass1_loop_s',0)+t['host_wall'].get('pass2_loop_s',0),4), 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t['forward_first_batch_gpu_s'],4), 'stat', round(t['statistics_gpu_s'],4), 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'steady', round(t.get('forward_steady_images_per_s',0)))
PY
done; done; done
