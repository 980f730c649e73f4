python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 64 -A minmax -D trt --skip_profiling -O /tmp/e2e/o0 > /dev/null 2>&1   # (page cache)
for rep in 1 2; do for A in hist mse; do for CB in 32 64; do
sleep 2
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A $A -D trt --calib_batch $CB --skip_profiling -O /tmp/e2e/out_$A$CB --timing_json /tmp/t.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/t.json'))
print('$A batch $CB', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(1024/t['tensor_calibration_wall_s']), 'fwd_gpu', round(t['forward_gpu_s'],4), 'first', round(t.get('forward_first_batch_gpu_s',0),4), 'stat', round(t['statistics_gpu_s'],4), 'steady', round(t.get('forward_steady_images_per_s',0)), 'load', round(t['load_model_wall_s'],3), t['host_wall'], {k: round(v,3) for k,v in t['startup'].items() if isinstance(v,float)})
print('   timeline', t['timeline_s'])
PY
done; done; done
