#!/bin/bash
# GPU box: what state is this box in?  k_abs_hist takes 518 - 530 us on a rested box and 543 - 570 us for a few minutes after
# profiling passes / long mse sweeps: clocks, power cap and partition modes beside the kernels' times.
rocm-smi --showclocks --showmaxpower --showmemorypartition --showcomputepartition 2>&1 | grep -i "fclk\|mclk\|Max Graphics\|Partition" | sed 's/^GPU\[0\]\s*: //'
python scripts/kbench.py --kernel hist --rounds 2 --batch 32 2>&1 | tail -1
python scripts/kbench.py --kernel minmax --rounds 2 --batch 32 2>&1 | tail -1
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | sed 's/^GPU\[0\]\s*: //'
cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -2
