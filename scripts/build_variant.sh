#!/bin/bash
# usage: scripts/build_variant.sh <out.so> <extra hipcc flags...>   — a tuning / debugging build of the library
out=$1; shift
cd "$(dirname "$0")/../dipoorlet_amd/csrc" && hipcc --offload-arch=gfx950 ${DPL_OPT:--O3} -std=c++17 -shared -fPIC -fno-fast-math -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function "$@" -o "$out" calib_kernels.hip octav_kernels.hip octav_tail_host.hip round_kernels.hip gemm_small.hip
