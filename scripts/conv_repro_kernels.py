"""Cuts a rocprofv3 kernel trace of scripts/conv_repro_probe.py at the k_gemm_small delimiters and prints, per convolution
configuration and call, the kernels that ran (part A), then the kernel-name sequence of each forward of part B and whether the
sequences of two forwards are the same.   python3 scripts/conv_repro_kernels.py <kernel_trace.csv> [probe.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
probe = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
# split at delimiters
groups, cur, run = [], [], 0
for n, d in names:
    if "k_gemm_small" in n and "sum" not in n:
        run += 1
        continue
    if "k_gemm_small_sum" in n:
        continue
    if run:
        groups.append((run, []))
        run = 0
    if groups:
        groups[-1][1].append((n, d))
cfg, call = -1, 0
short = lambda n: n.split("(")[0][:70]
tail = None
for run, ks in groups:
    if run >= 3:
        tail = ks
        break
    if run == 2:
        cfg, call = cfg + 1, 0
    call += 1
    conv = [(short(n), d) for n, d in ks if not n.startswith("void at::native") or "conv" in n.lower()]
    label = ""
    if probe and cfg < len(probe["part_a"]):
        r = probe["part_a"][cfg]
        label = f'{r["node"]} x{r["x"]} w{r["w"]} s{r["stride"]}'
    print(f"cfg {cfg:2d} call {call} {label}: " + " + ".join(f"{n} [{d / 1e3:.0f}us]" for n, d in conv))
if tail:
    print("part B:", len(tail), "kernels;", sum(1 for n, _ in tail if "naive" in n), "naive_conv launches")
print("naive_conv launches in the whole trace:", sum(1 for n, _ in names if "naive" in n))
