"""GPU box, under rocprofv3: where the time between two streaming kernels of the `-A mse` pipeline goes.
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python3 scripts/mse_run.py resnet50 64 17 ; python3 scripts/trace_gaps.py <dir>"""
import csv, glob, re, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r".*::(k_\w+).*", r"\1", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
rows.sort()
main = "k_octav_tail"
ones = [i for i, r in enumerate(rows) if r[2] == main]
# the second (warm) run — or, with a count as second argument, the last <count> launches (bench.py: the last sweep of `mse_lanes1`)
ones = ones[-int(sys.argv[2]):] if len(sys.argv) > 2 else ones[len(ones) // 2:]
gaps, between = [], defaultdict(list)
for a, b in zip(ones[:-1], ones[1:]):
    e_prev, s_next = rows[a][1], rows[b][0]
    gaps.append((s_next - e_prev) / 1e3)
    q = rows[b][3]
    for r in rows[a + 1:b]:
        if r[3] == q:                 # same queue as the streaming kernel: the caller's stream
            between[r[2]].append((r[1] - r[0]) / 1e3)
dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in ones]
big = sorted(((g, k) for k, g in enumerate(gaps)), reverse=True)[:5]
print("largest gaps (us, after launch #):", [(round(g, 1), k) for g, k in big])
print(f"{main}: {len(ones)} launches, mean {sum(dur) / len(dur):.1f} us; gap to the next one: mean {sum(gaps) / len(gaps):.1f} us "
      f"(min {min(gaps):.1f}, max {max(gaps):.1f})")
for k, v in sorted(between.items()):
    print(f"  on the same queue in the gap: {k:28s} {len(v) / len(gaps):.2f} per batch, mean {sum(v) / len(v):.1f} us")
# the other queues (the pipeline's side stream): what runs there per batch, and how much of it overlaps a streaming kernel
side = defaultdict(list)
lo, hi = rows[ones[0]][0], rows[ones[-1]][1]
tails = [(rows[i][0], rows[i][1]) for i in ones]
for r in rows:
    if r[2] != main and lo <= r[0] <= hi and r[2].startswith("k_"):
        ov = sum(max(0, min(r[1], e) - max(r[0], s)) for s, e in tails)
        side[r[2]].append(((r[1] - r[0]) / 1e3, ov / 1e3))
for k, v in sorted(side.items()):
    print(f"  {k:28s} {len(v) / len(ones):.2f} per batch, mean {sum(d for d, _ in v) / len(v):.1f} us, of which beside a streaming kernel {sum(o for _, o in v) / len(v):.1f} us")
span = (hi - lo) / 1e3 / len(ones)
print(f"per batch, first start to last end: {span:.1f} us")
