"""GPU: the CLI end to end (-A mse) on ResNet-50 at 448 x 448 input — tensors above one OCTAV slice — with the exact-tail form
(merge kernel) and with DPL_OCTAV_FORM=bracket (the two-read form): the clip ranges agree.  python scripts/e2e_big_check.py"""
import os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dipoorlet_amd import models
d = "/tmp/e2e_big"
os.makedirs(d + "/calib/input", exist_ok=True)
g = models.resnet50(image=448)
g.output_dir = d
g.save_onnx_model("r50_448")
rs = np.random.default_rng(0)
for i in range(16):
    rs.standard_normal(3 * 448 * 448).astype(np.float32).tofile(f"{d}/calib/input/{i}.bin")
res = {}
for multi, form in (("1", "tail"), ("0", "bracket")):
    env = dict(os.environ, DPL_OCTAV_FORM=form)
    r = subprocess.run([sys.executable, "-m", "dipoorlet_amd", "-M", f"{d}/r50_448.onnx", "-I", f"{d}/calib", "-N", "16", "-A", "mse", "-D", "trt",
                        "--calib_batch", "8", "--skip_profiling", "-O", f"{d}/out{multi}"], env=env, capture_output=True, text=True)
    print("form", form, "rc", r.returncode, [l for l in r.stderr.splitlines() if "OCTAV" in l][:2])
    res[multi] = json.load(open(f"{d}/out{multi}/act_clip_val.json"))
a, b = res["1"], res["0"]
worst = max(abs(a[k][j] - b[k][j]) / max(1.0, abs(b[k][j])) for k in a for j in (0, 1))
print("tensors", len(a), "worst relative difference", worst)
