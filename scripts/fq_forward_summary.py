"""Cuts the kernel trace of scripts/fq_forward_run.py at its k_channel_diff_sum delimiters and writes, for the two forms of the
fake-quantised forward (A every tensor exposed, B ReLU / Add + ReLU fused into the Q/DQ kernel): launches and GPU time per forward
per kernel family, the Q/DQ kernels' own durations against the bytes they move -> kernel_stats_fq_forward.md + fq_forward.json.
python3 scripts/fq_forward_summary.py <trace dir> <run.log> <out dir>"""
import csv, glob, json, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import source_sha

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
log = next(l for l in open(sys.argv[2]) if l.startswith("FQFWD "))
meta = json.loads(log[6:])
cuts = [i for i, r in enumerate(rows) if "k_channel_diff_sum" in r[2]]
assert len(cuts) == 3, cuts
N = meta["A"]["forwards"]
# block A = (cut0, cut1) holds its 10 counted forwards THEN B's 3 warm-up forwards: take the first N forwards' worth by the
# classifier head's launch (k_gemm_small: once per forward)


def first_forwards(lo, hi, n):
    heads = [i for i in range(lo, hi) if "k_gemm_small" in rows[i][2] and "sum" not in rows[i][2]]
    end = heads[n - 1]
    while end + 1 < hi and "k_gemm_small_sum" in rows[end + 1][2]:
        end += 1
    return rows[lo + 1:end + 1]


def family(name):
    n = name
    m = re.search(r"k_fake_quant<(\d)>", n)
    if m:
        return "k_fake_quant<%s>" % {"0": "none", "1": "relu", "2": "add_relu"}[m.group(1)]
    if "k_fake_quant" in n:
        return "k_fake_quant"
    if "launch_clamp" in n or "relu" in n.lower():
        return "torch ReLU (clamp_min)"
    if "CUDAFunctor_add" in n or "CUDAFunctorOnSelf_add" in n:
        return "torch Add"
    if n.startswith(("miopen", "igemm", "Cijk", "_ZN2ck", "batched_transpose", "transpose_", "SubTensorOp", "Im2d2Col", "naive_conv")):
        return "convolutions (MIOpen / Tensile)"
    if "max_pool" in n:
        return "max pool"
    if "k_gemm_small" in n:
        return "k_gemm_small"
    return "other torch kernels"


out = {"source_sha": source_sha(), "batch": meta["batch"], "fused_pairs": meta["fused_pairs"],
       "how": "rocprofv3 --kernel-trace over scripts/fq_forward_run.py: kernel durations summed per forward (10 forwards per form)"}
table = {}
for tag, (lo, hi) in (("A", (cuts[0], cuts[1])), ("B", (cuts[1], cuts[2]))):
    ks = first_forwards(lo, hi, N)
    fam = defaultdict(lambda: [0, 0])
    for s, e, n in ks:
        f = family(n)
        fam[f][0] += 1
        fam[f][1] += e - s
    table[tag] = fam
    fq_ns = sum(v[1] for k, v in fam.items() if k.startswith("k_fake_quant")) / N
    nbytes = meta[tag]["qdq_bytes_per_forward"]
    out[tag] = {"gpu_us_per_forward": sum(v[1] for v in fam.values()) / N / 1e3, "launches_per_forward": sum(v[0] for v in fam.values()) / N,
                "relu_launches_per_forward": fam["torch ReLU (clamp_min)"][0] / N, "add_launches_per_forward": fam["torch Add"][0] / N,
                "qdq_nodes_per_forward": meta[tag]["qdq_nodes_per_forward"], "qdq_bytes_per_forward": nbytes, "qdq_us_per_forward": fq_ns / 1e3,
                "qdq_frac_of_8TBps": nbytes / fq_ns / 8000, "forward_ms_by_events": meta[tag]["forward_ms_by_events"],
                "naive_conv_launches": sum(1 for _, _, n in ks if "naive_conv" in n)}
out["gpu_time_saved"] = 1.0 - out["B"]["gpu_us_per_forward"] / out["A"]["gpu_us_per_forward"]
# (bench.py quotes these two as roofline.fake_quant.product_forward_kernel_trace: the fused form's, on the fused byte count)
out["kernel"], out["frac_of_8TBps"], out["us_per_forward"] = "k_fake_quant<PRE>", out["B"]["qdq_frac_of_8TBps"], out["B"]["qdq_us_per_forward"]
out["nodes_per_forward"], out["bytes_per_forward"] = out["B"]["qdq_nodes_per_forward"], out["B"]["qdq_bytes_per_forward"]
os.makedirs(sys.argv[3], exist_ok=True)
json.dump(out, open(os.path.join(sys.argv[3], "fq_forward.json"), "w"), indent=1)
fams = sorted(set(table["A"]) | set(table["B"]), key=lambda f: -(table["A"].get(f, [0, 0])[1] + table["B"].get(f, [0, 0])[1]))
with open(os.path.join(sys.argv[3], "kernel_stats_fq_forward.md"), "w") as f:
    f.write(f"fake-quantised ResNet-50 forward, batch {meta['batch']}, -D trt, per forward (mean of {N}); A: every tensor exposed, B: only the output asked for\n\n")
    f.write("| kernel family | A launches | A us | B launches | B us |\n|---|---|---|---|---|\n")
    for fam in fams:
        a, b = table["A"].get(fam, [0, 0]), table["B"].get(fam, [0, 0])
        f.write(f"| {fam} | {a[0] / N:.1f} | {a[1] / N / 1e3:.1f} | {b[0] / N:.1f} | {b[1] / N / 1e3:.1f} |\n")
    f.write(f"| **all** | {out['A']['launches_per_forward']:.1f} | {out['A']['gpu_us_per_forward']:.1f} | {out['B']['launches_per_forward']:.1f} | {out['B']['gpu_us_per_forward']:.1f} |\n\n")
    for tag in ("A", "B"):
        o = out[tag]
        f.write(f"{tag}: Q/DQ kernels {o['qdq_us_per_forward']:.1f} us per forward over {o['qdq_nodes_per_forward']} nodes for {o['qdq_bytes_per_forward'] / 1e9:.3f} GB = "
                f"{o['qdq_bytes_per_forward'] / o['qdq_us_per_forward'] / 1e3:.0f} GB/s = {o['qdq_frac_of_8TBps']:.3f} of 8 TB/s; forward {o['forward_ms_by_events']:.2f} ms by HIP events\n")
    f.write(f"GPU time of the forward: - {100 * out['gpu_time_saved']:.1f} % (B against A); fused pairs {meta['fused_pairs']}\n")
print(open(os.path.join(sys.argv[3], "kernel_stats_fq_forward.md")).read())
