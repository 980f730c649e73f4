"""Write ResNet-50 as an .onnx plus N synthetic .bin images under a directory (for end-to-end CLI runs / traces).
python scripts/e2e_setup.py <dir> <N>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dipoorlet_amd import models

d, n = sys.argv[1], int(sys.argv[2])
os.makedirs(d, exist_ok=True)
g = models.resnet50()
g.output_dir = d
g.save_onnx_model("r50")
os.makedirs(os.path.join(d, "calib", "input"), exist_ok=True)
rs = np.random.default_rng(0)
base = rs.standard_normal((64, 3 * 224 * 224)).astype(np.float32)
for i in range(n):
    (base[i % 64] * np.float32(1.0 + 0.01 * (i // 64))).tofile(os.path.join(d, "calib", "input", f"{i}.bin"))
print("wrote", d)
