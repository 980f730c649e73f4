bash scripts/build_variant.sh /tmp/libprof.so -DDPL_RES_PROF > /tmp/b.log 2>&1 || tail -20 /tmp/b.log
DPL_LIB=/tmp/libprof.so python3 scripts/tail_prof.py 0
DPL_LIB=/tmp/libprof.so python3 scripts/tail_prof.py 0.1
