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
ones = ones[len(ones) // 2:]          # the second (warm) run
gaps, between = [], defaultdict(list)
for a, b in zip(ones[:-1], ones[1:]):
    e_prev, s_next = rows[a][1], rows[b][0]
    gaps.append((s_next - e_prev) / 1e3)
    q = rows[b][3]
    for r in rows[a + 1:b]:
        if r[3] == q:                 # same queue as the streaming kernel: the caller's stream
            between[r[2]].append((r[1] - r[0]) / 1e3)
dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in ones]
print(f"{main}: {len(ones)} launches, mean {sum(dur) / len(dur):.1f} us; gap to the next one: mean {sum(gaps) / len(gaps):.1f} us "
      f"(min {min(gaps):.1f}, max {max(gaps):.1f})")
for k, v in sorted(between.items()):
    print(f"  on the same queue in the gap: {k:28s} {len(v) / len(gaps):.2f} per batch, mean {sum(v) / len(v):.1f} us")
