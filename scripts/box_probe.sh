#!/bin/bash
# GPU box: what kind of box is this?  The boxes of the pool come in two kinds for k_abs_hist (518 - 530 us / 555 - 570 us) and for the
# sparse read + write pattern: clocks, power cap and partition modes beside the kernel's time.
rocm-smi --showclocks --showmaxpower --showmemorypartition --showcomputepartition 2>&1 | grep -i "fclk\|mclk\|Max Graphics\|Partition" | sed 's/^GPU\[0\]\s*: //'
python scripts/kbench.py --kernel hist --rounds 2 --batch 32 2>&1 | tail -1
python scripts/kbench.py --kernel minmax --rounds 2 --batch 32 2>&1 | tail -1
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" | sed 's/^GPU\[0\]\s*: //'
cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -2
