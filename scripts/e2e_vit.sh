# GPU box: the CLI on ViT-B/16 (557 exposed tensors), -A mse, N = 256, by --calib_batch
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from dipoorlet_amd import models
d = "/tmp/e2e_vit"
os.makedirs(d + "/calib/input", exist_ok=True)
g = models.vit_b16(seed=5, attn_gain=10.0); g.output_dir = d; g.save_onnx_model("vit")
rs = np.random.default_rng(0)
base = rs.standard_normal((64, 3 * 224 * 224)).astype(np.float32)
for i in range(256):
    (base[i % 64] * np.float32(1.0 + 0.01 * (i // 64))).tofile(f"{d}/calib/input/{i}.bin")
PY
python -m dipoorlet_amd -M /tmp/e2e_vit/vit.onnx -I /tmp/e2e_vit/calib -N 32 -A minmax -D trt --skip_profiling -O /tmp/e2e_vit/o0 > /dev/null 2>&1
for rep in 1 2 3; do for CB in ${DPL_VIT_BATCHES:-16 32 64}; do
sleep 2
python -m dipoorlet_amd -M /tmp/e2e_vit/vit.onnx -I /tmp/e2e_vit/calib -N 256 -A mse -D trt --calib_batch $CB --skip_profiling -O /tmp/e2e_vit/out --timing_json /tmp/tv.json > /tmp/cli.log 2>&1 || tail -5 /tmp/cli.log
python - <<PY
import json
t=json.load(open('/tmp/tv.json'))
print('vit mse batch $CB', 'cal_wall', round(t['tensor_calibration_wall_s'],3), 'img/s', round(256/t['tensor_calibration_wall_s']), 'fwd_gpu', round(t['forward_gpu_s'],3), 'first', round(t.get('forward_first_batch_gpu_s',0),3), 'steady', round(t.get('forward_steady_images_per_s',0)), 'stat', round(t['statistics_gpu_s'],3), t['host_wall'], t['timeline_s'])
PY
done; done
