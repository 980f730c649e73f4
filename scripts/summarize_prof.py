#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text/JSON files fit for profiles/.

  summarize_prof.py stats  <dir> <out.md>          kernel-trace --stats -> per-kernel table
  summarize_prof.py pmc    <dir> <counter> <out.json> [kernel-substring]   per-kernel mean of a PMC counter
"""
import csv
import glob
import json
import os
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
