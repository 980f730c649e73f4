# GPU box: the end-to-end `-A mse` CLI run (N = 4096, default batch) with the OCTAV pipeline's side stream at high priority (-1: rounds 4 - 5)
# against normal priority (0: round 6), alternating, fresh processes.
python scripts/e2e_setup.py /tmp/e2e 4096 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 64 -A minmax -D trt --skip_profiling -O /tmp/e2e/out_w > /dev/null 2>&1
for rep in 1 2 3; do for P in -1 0; do
DPL_OCTAV_SIDE_PRIO=$P python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 4096 -A mse -D trt --skip_profiling -O /tmp/e2e/out_$P --timing_json /tmp/t_$P.json > /tmp/cli_$P.log 2>&1 || tail -5 /tmp/cli_$P.log
python - <<PY
import json
t=json.load(open('/tmp/t_$P.json'))
print('side prio $P', 'pass1_loop', t['host_wall'].get('pass1_loop_s'), 'fwd_gpu', round(t['forward_gpu_s'],4), 'steady img/s', round(t.get('forward_steady_images_per_s',0)), 'stat', round(t['statistics_gpu_s'],4), 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(4096/t['tensor_calibration_wall_s']))
PY
done; done
