"""profiles/r06/conv_repro.md from the files scripts/conv_repro.sh and scripts/conv_repro_cold.sh leave in gpurun_out/conv_repro/.
python3 scripts/conv_repro_report.py [dir] [out.md]"""
import collections
import json
import os
import re
import sys

d = (sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_repro").rstrip("/") + "/"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06/conv_repro.md"


def load(f):
    return json.load(open(d + f + ".json"))


def kern(f):
    res = collections.OrderedDict()
    for line in open(d + f):
        m = re.match(r"cfg\s+(\d+) call (\d) (\S+) (.*?): (.*)", line)
        if m:
            ks = [k.split(" [")[0] for k in m.group(5).split(" + ")]
            res[(int(m.group(1)), int(m.group(2)))] = (m.group(3), [k for k in ks if not k.startswith("__amd_rocclr")])
    return res


AUX = ("batched_transpose", "transpose_", "SubTensorOp")


def main_kernel(ks):
    ks = [x for x in ks if not x.startswith(AUX)]
    return ks[0][:48] if ks else "-"


out = ["# profiles/r06/conv_repro.md — what makes two forwards differ, and when MIOpen's naive convolution runs\n",
       "Produced on MI355X boxes of the pool by `scripts/conv_repro.sh` and `scripts/conv_repro_cold.sh` (`scripts/conv_repro_probe.py`: one\n"
       "process, no helper threads; kernel names from `rocprofv3 --kernel-trace`, cut per call by `scripts/conv_repro_kernels.py`; this file by\n"
       "`scripts/conv_repro_report.py`).  torch 2.10.0+rocm7.0, MIOpen 3.5.0.\n",
       "## 1. Which convolutions do not repeat bit for bit (ResNet-50, batch 32, 5 calls on the same input)\n"]
r, rd, k, kd = load("default_b32"), load("det_b32"), kern("kernels_default.txt"), kern("kernels_det.txt")
out += ["| configuration | default: kernel (call 2) | calls 2..5 == call 1 | max diff | `cudnn.deterministic`: kernel | == call 1 |", "|---|---|---|---|---|---|"]
for i, a in enumerate(r["part_a"]):
    zeroed = any(x.startswith("SubTensorOp") for x in k[(i, 2)][1])
    out.append(f"| {a['node']} x{a['x']} w{a['w']} s{a['stride'][0]} | `{main_kernel(k[(i, 2)][1])}`"
               f"{' (output zeroed first: `SubTensorOpWithScalar1d`)' if zeroed else ''} | {'yes' if all(a['bit_equal_to_call1']) else '**no**'} | "
               f"{a['maxdiff']:.3g} | `{main_kernel(kd[(i, 2)][1])}` | {'yes' if all(rd['part_a'][i]['bit_equal_to_call1']) else 'no'} |")
out += ["", "Forwards of ONE session over the same input (tensors that differ from forward 1 / max difference / root node):\n",
        "| run | fp forward 2 | fake-quantised forward 2 | two sessions of one quantised graph bit-equal |", "|---|---|---|---|"]
for f, name in (("default_b32", "ResNet-50 b32 default"), ("det_b32", "ResNet-50 b32 deterministic"), ("default_b64", "ResNet-50 b64 default"),
                ("r18_default_b4", "ResNet-18 64x64 b4 default (the tests' network)"), ("r18_det_b4", "ResNet-18 64x64 b4 deterministic"),
                ("vit_default_b4", "ViT-B/16 b4 default"), ("vit_det_b4", "ViT-B/16 b4 deterministic")):
    x = load(f)

    def fmt(p):
        if p not in x:
            return "-"
        y = x[p]["forwards"][0]
        return f"{y['differ']}/{y['tensors']} differ, max {y['maxdiff']:.3g}" + (f", root {y['root_nodes'][0][1]}" if y["root_nodes"] else "")
    out.append(f"| {name} | {fmt('part_b_fp')} | {fmt('part_b_quant')} | {x.get('two_sessions_bit_equal', '-')} |")
out += ["", "## 2. When the naive convolution runs (ResNet-50, batch 16, empty `MIOPEN_USER_DB_PATH`, then the same command again)\n"]
kc, kw, c, w = kern("kernels_cold.txt"), kern("kernels_warm.txt"), load("cold"), load("warm")
out += ["| configuration | cold db, call 1: launches (naive / other benchmarked kernels) | cold db, call 2 | warm db, call 1 | first call, host ms cold / warm |",
        "|---|---|---|---|---|"]
for i, a in enumerate(c["part_a"]):
    ks = kc[(i, 1)][1]
    others = sorted({x[:34] for x in ks if "naive" not in x and not x.startswith(AUX)})
    out.append(f"| {a['node']} | {sum('naive' in x for x in ks)} naive of {len(ks)}; {', '.join('`' + o + '`' for o in others[:4])} | "
               f"`{main_kernel(kc[(i, 2)][1])}` | `{main_kernel(kw[(i, 1)][1])}` ({sum('naive' in x for x in kw[(i, 1)][1])} naive) | "
               f"{a['gpu_ms_host_ms'][0][1]:.0f} / {w['part_a'][i]['gpu_ms_host_ms'][0][1]:.1f} |")
n_cold = sum(sum("naive" in x for x in v[1]) for v in kc.values())
first = sum(sum("naive" in x for x in v[1]) for (ci, call), v in kc.items() if call == 1)
out.append(f"\nnaive launches: cold pass {n_cold}, {first} of them in a configuration's FIRST call; warm pass "
           f"{sum(sum('naive' in x for x in v[1]) for v in kw.values())}.  Sum of the first calls' host times: "
           f"{sum(a['gpu_ms_host_ms'][0][1] for a in c['part_a']):.0f} ms cold, {sum(a['gpu_ms_host_ms'][0][1] for a in w['part_a']):.0f} ms warm.\n")
# section 3: the find mode on an empty user find-db (scripts/find_mode_probe.sh -> gpurun_out/find_mode/)
fm = os.path.join(os.path.dirname(d.rstrip("/")), "find_mode")
if os.path.isdir(fm):
    out += ["## 3. MIOpen's find mode on an EMPTY user find-db (`scripts/find_mode_probe.sh`; `library` = DYNAMIC_HYBRID, the library's default; FAST = this package's)\n",
            "| network, batch | find mode | find-db | first calls of all configurations, host ms | their steady GPU time, ms | forwards 1 … 6, ms |", "|---|---|---|---|---|---|"]
    rows = [("resnet50_64", m, p_, f"{m}_{p_}") for m in ("library", "FAST") for p_ in ("cold", "warm")]
    rows += [(t, m, "cold", f"{t}_{m}") for t in ("resnet18_64", "vit_b16_16", "resnet50_16") for m in ("library", "FAST")]
    for tag, mode, db, f in rows:
        path = os.path.join(fm, f + ".json")
        if not os.path.exists(path):
            continue
        r = json.load(open(path))
        first = sum(a["gpu_ms_host_ms"][0][1] for a in r["part_a"])
        steady = sum(min(m_[0] for m_ in a["gpu_ms_host_ms"][1:]) for a in r["part_a"])
        out.append(f"| {', batch '.join(tag.rsplit('_', 1))} | {mode} | {db} | {first:.0f} | {steady:.3f} | {', '.join('%.1f' % x for x in r['part_b_fp']['forward_ms'])} |")
    out.append("\nOn a miss the library's default benchmarks every applicable solver (section 2) and writes the winner to the user find-db; FAST takes the "
               "heuristic choice and benchmarks nothing: the cold first calls cost 0.02 – 0.08 s instead of 0.24 – 4.9 s, the steady forward is the same to 0 – 3 % "
               "(the convolutions' own time + 0 – 6 %), a process on a warm find-db is unchanged.  `executor.py` sets `MIOPEN_FIND_MODE=FAST` unless the caller has "
               "set it (`DPL_MIOPEN_FIND_MODE=library`: the library's default).\n")
open(dst, "w").write("\n".join(out))
print(dst)
