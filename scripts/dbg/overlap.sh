# kernel trace of the mse sweep: do k_octav_walk (side stream) and k_octav_oneread (main stream) overlap?
export TMPDIR=/tmp
rm -rf gpurun_out/prof_tmp; mkdir -p gpurun_out/prof_tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tmp/tr -o bench -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 0 --mse-steps 1 > gpurun_out/prof_tmp/bench.json 2> gpurun_out/prof_tmp/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_tmp/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_octav' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) * 3 // 4:]
t0 = int(rows[0]['Start_Timestamp'])
import re
for r in rows[:40]:
    n = re.search(r'k_octav_[a-z_]+', r['Kernel_Name']).group(0)
    print(f"{n:24s} q{r.get('Queue_Id','?'):>3} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
rm -rf gpurun_out/prof_tmp
