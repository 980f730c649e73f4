"""ctypes wrapper of oracle/c_oracle.c (CPU ORACLE — test infrastructure, not product code)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def _cpu_tag():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def lib():
    global _lib
    if _lib is None:
        # built with -march=native: rebuild when the library came from another machine (the build container's
        # .so travels to the GPU box with the snapshot) or is older than the source
        tag_file = os.path.join(_HERE, "_build", "cpu.txt")
        tag = _cpu_tag()
        stale = (not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "c_oracle.c"))
                 or not os.path.exists(tag_file) or open(tag_file).read() != tag)
        if stale:
            subprocess.run(["make", "-s", "-C", _HERE, "clean"], check=True)
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
            with open(tag_file, "w") as f:
                f.write(tag)
        l = C.CDLL(_SO)
        l.dplo_minmax.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.dplo_abs_hist.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]
        l.dplo_abs_hist.restype = C.c_int
        l.dplo_octav_scale.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        l.dplo_octav_scale.restype = C.c_float
        l.dplo_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
        l.dplo_batch.restype = C.c_int
        _lib = l
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32).ravel()


def minmax(x):
    x = _f32(x)
    lo, hi = C.c_float(), C.c_float()
    lib().dplo_minmax(x.ctypes.data, x.size, C.byref(lo), C.byref(hi))
    return np.float32(lo.value), np.float32(hi.value)


def abs_hist(x, bins, dmax):
    x = _f32(x)
    h = np.zeros(int(bins), np.int64)
    st = lib().dplo_abs_hist(x.ctypes.data, x.size, int(bins), float(np.float32(dmax)), h.ctypes.data)
    if st == 1:
        raise ValueError("supplied range is not finite")
    if st == 2:
        raise ValueError("Too many bins for data range.")
    return h


def octav_scale(x, unsigned=1):
    x = _f32(x)
    scratch = np.empty(2 * max(x.size, 1), np.float32)
    return np.float32(lib().dplo_octav_scale(x.ctypes.data, x.size, int(unsigned), scratch.ctypes.data))


def batch(arrays, algo, bins=2048, threads=0):
    """Runs minmax (+ hist / OCTAV) over a list of fp32 arrays, OpenMP-parallel over tensors.
    Returns (threads_used, mins, maxs, s or None, hist or None)."""
    arrays = [_f32(a) for a in arrays]
    n = len(arrays)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
    counts = np.array([a.size for a in arrays], np.int64)
    mins, maxs, s = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.float32)
    code = {"minmax": 0, "hist": 1, "mse": 2}[algo]
    hist = np.zeros((n, bins), np.int64) if code == 1 else np.zeros((1, 1), np.int64)
    used = lib().dplo_batch(ptrs, counts.ctypes.data, n, code, int(bins), int(threads), mins.ctypes.data,
                            maxs.ctypes.data, s.ctypes.data, hist.ctypes.data)
    return used, mins, maxs, (s if code == 2 else None), (hist if code == 1 else None)
