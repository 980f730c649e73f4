#!/bin/bash
# The full GPU suite K times in a row on this box, as the driver runs it (pytest -m gpu -x -q); appends one line per run to
# gpurun_out/gputest_repeats_<tag>.txt (profiles/r06/gputest_repeats.txt = the leases' files one after the other).   scripts/gputest_repeats.sh <K> <lease tag>
K=${1:-4}; TAG=${2:-lease}
out=gpurun_out/gputest_repeats_$TAG.txt   # (one file per lease: a lease starts from an empty gpurun_out/, and what it writes replaces the file of the same name here)
echo "== $TAG: $(hostname) $(date -u +%FT%TZ) sources $(cat dipoorlet_amd/csrc/*.hip dipoorlet_amd/csrc/*.hpp tests/*.py dipoorlet_amd/*.py | sha256sum | cut -c1-12)" >> $out
for i in $(seq 1 $K); do
  python -m pytest tests -m gpu -x -q > gpurun_out/gputest_${TAG}_$i.log 2>&1
  echo "$TAG run $i: rc=$? $(tail -1 gpurun_out/gputest_${TAG}_$i.log)" >> $out
done
tail -$((K+1)) $out
