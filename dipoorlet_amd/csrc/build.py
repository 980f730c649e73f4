"""Builds libdipoorlet_hip.so (gfx950 code object + C ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored
but travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, f) for f in ("calib_kernels.hip", "octav_kernels.hip", "octav_tail_host.hip", "round_kernels.hip", "gemm_small.hip")]
HDR = [os.path.join(HERE, "..", "..", "include", "dipoorlet_hip.h"), os.path.join(HERE, "common.hpp"),
       os.path.join(HERE, "octav_common.hpp"), os.path.join(HERE, "octav_tail.hpp")]
OUT = os.path.join(HERE, "libdipoorlet_hip.so")


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-fno-fast-math", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-o", OUT] + SRC
    cmd += os.environ.get("DPL_HIPCC_EXTRA", "").split()  # tuning knobs (-DDPL_...=N), see scripts/variant_bench.sh
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose and r.stderr.strip():
        print(r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
