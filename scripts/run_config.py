#!/usr/bin/env python3
"""End-to-end run of a BASELINE config on one GPU: build the network as a real .onnx, write N synthetic
.bin calibration images, run the CLI pipeline in-process and report wall time per phase.

  python scripts/run_config.py --model resnet50 --algo hist -N 256 --batch 16
  python scripts/run_config.py --model vit_b16 --algo mse -N 32 --batch 8 --bc
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="resnet50", choices=["resnet18", "resnet50", "vit_b16"])
    p.add_argument("--algo", default="hist")
    p.add_argument("--deploy", default="trt")
    p.add_argument("-N", type=int, default=256)
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--bc", action="store_true")
    p.add_argument("--profiling", action="store_true")
    a = p.parse_args()
    from dipoorlet_amd import models
    from dipoorlet_amd.__main__ import main as cli
    d = tempfile.mkdtemp(prefix="dpl_cfg_")
    t0 = time.time()
    g = getattr(models, a.model)()
    g.output_dir = d
    g.save_onnx_model("model")
    os.makedirs(os.path.join(d, "calib", "input"))
    rng = np.random.default_rng(0)
    for i in range(a.N):
        rng.standard_normal(3 * 224 * 224).astype(np.float32).tofile(os.path.join(d, "calib", "input", f"{i}.bin"))
    t1 = time.time()
    argv = ["-M", os.path.join(d, "model.onnx"), "-I", os.path.join(d, "calib"), "-N", str(a.N), "-A", a.algo, "-D",
            a.deploy, "-O", os.path.join(d, "out"), "--calib_batch", str(a.batch)]
    if a.bc:
        argv.append("--bc")
    if not a.profiling:
        argv.append("--skip_profiling")
    torch.cuda.synchronize()
    t2 = time.time()
    cli(argv)
    torch.cuda.synchronize()
    t3 = time.time()
    act = json.load(open(os.path.join(d, "out", "act_clip_val.json")))
    print(json.dumps({"model": a.model, "algo": a.algo, "N": a.N, "batch": a.batch, "bc": a.bc,
                      "tensors": len(act), "setup_s": round(t1 - t0, 2), "pipeline_s": round(t3 - t2, 2),
                      "images_per_s_end_to_end": round(a.N / (t3 - t2), 1),
                      "peak_hbm_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
