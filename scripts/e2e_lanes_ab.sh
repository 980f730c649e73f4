# GPU box: the end-to-end `-A mse` CLI run with the streaming kernels on the caller's stream (DPL_OCTAV_LANES=1) against two lane streams (2), alternating.
python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -2; ls /tmp/e2e | head
for rep in 1 2 3 4; do for L in 1 2; do
DPL_OCTAV_LANES=$L python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A mse -D trt --calib_batch 32 --skip_profiling -O /tmp/e2e/out_$L --timing_json /tmp/t_$L.json > /tmp/cli_$L.log 2>&1 || tail -5 /tmp/cli_$L.log
python - <<PY
import json
t=json.load(open('/tmp/t_$L.json'))
print('lanes=$L', 'pass1_loop', t['host_wall']['pass1_loop_s'], 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t['forward_first_batch_gpu_s'],4), 'stat', round(t['statistics_gpu_s'],4), 'cal_wall', round(t['tensor_calibration_wall_s'],3))
PY
done; done
