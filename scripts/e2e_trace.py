"""Kernel trace of an end-to-end CLI run (rocprofv3 --kernel-trace --output-format csv -d <dir>): per queue the busy time and
the idle gaps, the kernels that take the time, and — for the main queue — how long the network forward of a batch takes from
its first to its last kernel.  python scripts/e2e_trace.py <dir> [<dir2> ...]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*", "", n)[:48]


for d in sys.argv[1:]:
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    if not rows:
        print(d, "no kernels")
        continue
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"== {d}: {len(rows)} kernels over {(t1 - t0) / 1e6:.1f} ms")
    byq = defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in rs)
        gaps = [max(0, b[0] - a[1]) for a, b in zip(rs[:-1], rs[1:])]
        print(f"  queue {q}: {len(rs)} kernels, busy {busy / 1e6:.1f} ms, span {(rs[-1][1] - rs[0][0]) / 1e6:.1f} ms, idle between kernels {sum(gaps) / 1e6:.1f} ms "
              f"(gaps > 20 us: {sum(1 for g in gaps if g > 20000)}, > 200 us: {sum(1 for g in gaps if g > 200000)})")
    agg = defaultdict(lambda: [0, 0])
    for s, e, k, q in rows:
        agg[k][0] += 1
        agg[k][1] += e - s
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {k:48s} x{n:6d}  total {t / 1e6:8.2f} ms  avg {t / n / 1e3:8.1f} us")
