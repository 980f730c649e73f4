rm -rf /tmp/e2e; python scripts/e2e_setup.py /tmp/e2e 1024 2>&1 | tail -1
python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --calib_batch 32 --skip_profiling -O /tmp/e2e/o1 > /dev/null 2>&1
/usr/bin/time -v python -m dipoorlet_amd -M /tmp/e2e/r50.onnx -I /tmp/e2e/calib -N 1024 -A hist -D trt --calib_batch 32 --skip_profiling -O /tmp/e2e/o2 2>&1 | grep -E "Elapsed|Maximum resident"
python - <<'PY' 2>&1 | grep -v "^$" | tail -70
import cProfile, pstats, sys, runpy, time
sys.argv=["dipoorlet_amd","-M","/tmp/e2e/r50.onnx","-I","/tmp/e2e/calib","-N","1024","-A","hist","-D","trt","--calib_batch","32","--skip_profiling","-O","/tmp/e2e/o3"]
pr=cProfile.Profile(); t0=time.time(); pr.enable()
try:
    runpy.run_module("dipoorlet_amd", run_name="__main__")
except SystemExit: pass
pr.disable(); print("main wall", time.time()-t0)
st=pstats.Stats(pr); st.sort_stats("cumulative").print_stats(50)
PY
