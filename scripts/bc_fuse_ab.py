"""GPU box: `--bc` over ResNet-50 (N = 256, -D trt) with the ReLU / Add + ReLU chains of the fake-quantised walk fused into the Q/DQ
kernel (default) against DPL_FUSE_RELU=0, alternating, fresh CLI processes under the library's deterministic algorithms; the biases
of the two runs are compared at the end.   python3 scripts/bc_fuse_ab.py [N]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = "/tmp/e2e_bc"
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "e2e_setup.py"), d, str(n)], check=True, capture_output=True)


def cli(fuse, nimg, tag):
    env = dict(os.environ, DPL_FUSE_RELU=str(fuse), DPL_DETERMINISTIC="1")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "dipoorlet_amd", "-M", f"{d}/r50.onnx", "-I", f"{d}/calib", "-N", str(nimg), "-A", "minmax", "-D", "trt",
                        "--skip_profiling", "--bc", "-O", f"{d}/out_{tag}"], cwd=ROOT, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print((r.stderr or r.stdout)[-1500:])
        sys.exit(1)
    return time.perf_counter() - t0


cli(1, 64, "warm")
for rep in range(3):
    for fuse in (0, 1):
        print(f"DPL_FUSE_RELU={fuse}: process wall {cli(fuse, n, 'f%d' % fuse):.2f} s", flush=True)
import numpy as np
from dipoorlet_amd.graph import ONNXGraph
a, b = ONNXGraph.load(f"{d}/out_f0/update_bias_model.onnx"), ONNXGraph.load(f"{d}/out_f1/update_bias_model.onnx")
nodes = [x for x in a.graph.node if x.op_type in ("Conv", "Gemm")]
print("biases bit-equal with and without the fusion:", all(np.array_equal(a.get_initializer(x.input[2]), b.get_initializer(x.input[2])) for x in nodes), len(nodes), "nodes")
