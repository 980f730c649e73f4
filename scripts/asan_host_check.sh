#!/bin/bash
# CPU box: the HOST side of the C ABI (dpl_build_work_items, dpl_build_balanced_items, dpl_build_octav_slices, dpl_octav_plan_*,
# dpl_octav_fallback_layout) under AddressSanitizer + UndefinedBehaviorSanitizer: the library is built with the sanitizers on
# the host code only (-Xarch_host; the device code objects are the shipping ones) and tests/test_capi_load.py runs against it.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/dpl_asan
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -shared -fPIC -fno-fast-math -ffp-contract=off -munsafe-fp-atomics \
  -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -o $OUT/libdipoorlet_hip.so \
  $ROOT/dipoorlet_amd/csrc/calib_kernels.hip $ROOT/dipoorlet_amd/csrc/octav_kernels.hip $ROOT/dipoorlet_amd/csrc/octav_tail_host.hip \
  $ROOT/dipoorlet_amd/csrc/round_kernels.hip $ROOT/dipoorlet_amd/csrc/gemm_small.hip
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 DPL_LIB=$OUT/libdipoorlet_hip.so \
  python -m pytest tests/test_capi_load.py -q -m "not gpu" -p no:cacheprovider
