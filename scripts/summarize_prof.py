#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text/JSON files fit for profiles/.

  summarize_prof.py stats  <dir> <out.md>          kernel-trace --stats -> per-kernel table
  summarize_prof.py pmc    <dir> <counter> <out.json> [kernel-substring]   per-kernel mean of a PMC counter
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def stats(d, out):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    with open(out, "w") as o:
        o.write("| kernel | calls | total ns | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0)):
            o.write("| {} | {} | {} | {:.0f} | {} | {} | {} |\n".format(
                r.get("Name", "?")[:90], r.get("Calls"), r.get("TotalDurationNs"),
                float(r.get("AverageNs", 0) or 0), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
    print(open(out).read())


def pmc(d, counter, out, sub=""):
    acc = defaultdict(list)
    for f in find(d, "counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter and sub in r.get("Kernel_Name", ""):
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    res = {k[:120]: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in acc.items()}
    with open(out, "w") as o:
        json.dump({"counter": counter, "kernels": res}, o, indent=1)
    print(json.dumps(res, indent=1))


def source_sha():
    """sha256 over the kernel sources: a traffic figure is only quoted for the code it was measured on."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dipoorlet_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def traffic(fetch_json, write_json, out):
    """HBM bytes per launch = 2 x FETCH_SIZE (gfx950: wide coalesced reads are tallied at half, MI355X_MICROARCH.md) + WRITE_SIZE,
    both in KiB per the counter definition, per kernel."""
    f = json.load(open(fetch_json))["kernels"]
    w = json.load(open(write_json))["kernels"]
    res = {"source_sha": source_sha(), "batch": 32, "kernels": {}}
    for k, v in f.items():
        wr = w.get(k, {"mean": 0.0})["mean"]
        m = re.search(r"\bk_[a-z0-9_]+", k)
        if not m:
            continue
        res["kernels"][m.group(0)] = {"fetch_size_mean": v["mean"], "write_size_mean": wr, "launches": v["launches"],
                                                           "hbm_bytes_per_launch": (2.0 * v["mean"] + wr) * 1024.0}
    with open(out, "w") as o:
        json.dump(res, o, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
