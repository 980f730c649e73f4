"""ctypes binding of the C ABI (include/dipoorlet_hip.h) exported by csrc/libdipoorlet_hip.so.

There is NO CPU fallback: if the library is missing or a call fails this raises.  PyTorch only
supplies device memory and streams; the signatures carry raw pointers and sizes.
"""
import ctypes as C
import os

# torch must be loaded first: its bundled libamdhip64 then satisfies this library's dependency, so the
# kernels run on the SAME HIP runtime that owns torch's streams and allocations.  Loading our library
# first pulls in /opt/rocm's copy as a second runtime, which cannot see the device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPL_LIB: another build of the same sources (kernel-tuning variants, scripts/variant_*.sh); never a different code path
LIB_PATH = os.environ.get("DPL_LIB") or os.path.join(_HERE, "csrc", "libdipoorlet_hip.so")

ABI_VERSION = 21
MAX_BINS = 16384


class Span(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("count", C.c_uint64), ("seg", C.c_uint32), ("slot", C.c_uint32)]


class WorkItem(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("count", C.c_uint32), ("seg", C.c_uint32), ("slot", C.c_uint32),
                ("reserved", C.c_uint32)]


class HistRange(C.Structure):
    _fields_ = [("first", C.c_float), ("last", C.c_float), ("step", C.c_float), ("inv", C.c_float),
                ("zero_bin", C.c_uint32), ("status", C.c_uint32), ("dmax", C.c_float), ("exact_div", C.c_uint32)]


class OctavState(C.Structure):
    _fields_ = [("sum", C.c_double), ("cnt_gt", C.c_uint64), ("cnt_le", C.c_uint64), ("min_enc", C.c_uint32),
                ("max_enc", C.c_uint32), ("nan_seen", C.c_uint32), ("done", C.c_uint32), ("s", C.c_float),
                ("unsigned_div", C.c_float), ("iters", C.c_uint32), ("mode", C.c_uint32), ("n_elems", C.c_uint64),
                ("len0", C.c_uint32), ("len1", C.c_uint32), ("cur", C.c_uint32), ("reserved", C.c_uint32)]


class RoundStepParams(C.Structure):
    _fields_ = [("lr", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step", C.c_int32), ("adam", C.c_int32), ("clamp", C.c_int32), ("reserved", C.c_int32),
                ("grad_scale", C.c_float), ("reg_beta", C.c_float), ("reg_lambda", C.c_float),
                ("reserved2", C.c_float)]


class FakeQuantParams(C.Structure):
    """dpl_fake_quant_params: one tensor's quantisation parameters for dpl_fake_quant_items."""
    _fields_ = [("d_scale", C.c_void_p), ("d_zero_point", C.c_void_p), ("n_channels", C.c_int64), ("inner", C.c_int64),
                ("qlo", C.c_int32), ("qhi", C.c_int32)]


class OctavOnereadJob(C.Structure):
    """dpl_octav_oneread_job (include/dipoorlet_hip.h): one batch of the exact-tail OCTAV form."""
    _fields_ = [("d_slices", C.c_void_p), ("n_slices", C.c_int64), ("d_pair_slice0", C.c_void_p),
                ("d_pair_spans", C.c_void_p), ("d_pair_base", C.c_void_p), ("d_pair_order", C.c_void_p),
                ("n_pairs", C.c_int64), ("n_tensors", C.c_int64), ("n_small", C.c_int64), ("n_multi", C.c_int64),
                ("d_items", C.c_void_p), ("n_items", C.c_int64), ("d_block_begin", C.c_void_p), ("n_blocks", C.c_int64),
                ("d_seg_ptrs", C.c_void_p), ("d_states", C.c_void_p), ("d_lh", C.c_void_p), ("d_pred", C.c_void_p),
                ("d_list0", C.c_void_p), ("d_list1", C.c_void_p), ("d_pair_base_full", C.c_void_p), ("d_clist0", C.c_void_p),
                ("d_clist1", C.c_void_p), ("d_rescue_bm", C.c_void_p), ("d_missed", C.c_void_p), ("d_resc", C.c_void_p),
                ("d_vis", C.c_void_p), ("write_epoch", C.c_int32), ("reset_epoch", C.c_int32), ("dynamic_sym", C.c_int32),
                ("max_iters", C.c_int32), ("compaction_inline", C.c_int32), ("reserved", C.c_int32)]


class OctavWorkspaceSizes(C.Structure):
    """dpl_octav_workspace_sizes: bytes of every part of the exact-tail form's workspace (dpl_octav_plan_sizes)."""
    _fields_ = [("tables_bytes", C.c_uint64), ("history_bytes", C.c_uint64), ("state_bytes", C.c_uint64), ("rescue_bytes", C.c_uint64),
                ("list_bytes", C.c_uint64), ("fallback_bytes", C.c_uint64), ("result_bytes", C.c_uint64),
                ("n_pairs", C.c_int64), ("n_slices", C.c_int64), ("n_multi", C.c_int64), ("n_small", C.c_int64)]


assert C.sizeof(RoundStepParams) == 64 and C.sizeof(OctavOnereadJob) == 240 and C.sizeof(FakeQuantParams) == 40
assert C.sizeof(Span) == 24 and C.sizeof(WorkItem) == 24 and C.sizeof(HistRange) == 32 and C.sizeof(OctavState) == 80

_P, _I64, _I32, _U64, _DBL = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_double

# name -> (restype, argtypes); mirrors include/dipoorlet_hip.h one to one
SIGNATURES = {
    "dpl_abi_version": (C.c_int, []),
    "dpl_last_error": (C.c_char_p, []),
    "dpl_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "dpl_stream_priority_range": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dpl_stream_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "dpl_stream_destroy": (C.c_int, [_P]),
    "dpl_build_work_items": (_I64, [_P, _I64, _U64, _P, _I64]),
    "dpl_build_balanced_items": (_I64, [_P, _I64, _I64, _P, _I64, _P]),
    "dpl_minmax_init": (C.c_int, [_P, _P, _P, _I64, _P]),
    "dpl_minmax_accumulate": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "dpl_minmax_finalize": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "dpl_minmax_encode": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P]),
    "dpl_hist_prepare": (C.c_int, [_P, _P, _I64, C.c_int, _P, _P]),
    "dpl_abs_hist_accumulate": (C.c_int, [_P, _I64, _P, _I64, _P, _P, C.c_int, _P, _P]),
    "dpl_hist_percentile": (C.c_int, [_P, _P, _P, _I64, C.c_int, _DBL, _P, _P]),
    "dpl_octav_init": (C.c_int, [_P, _I64, C.c_int, _P]),
    "dpl_octav_run_compact": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "dpl_octav_run": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, C.c_int, C.c_int, _P]),
    "dpl_octav_run_bracket": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int,
                                        C.c_int, _P]),
    "dpl_octav_slice_cap": (C.c_uint32, []),
    "dpl_octav_small_pair": (C.c_uint32, []),
    "dpl_build_octav_slices": (_I64, [_P, _I64, _P, _I64, _P]),
    "dpl_octav_oneread_prepare": (C.c_int, [_P, _P]),
    "dpl_octav_oneread_stream": (C.c_int, [_P, _P]),
    "dpl_octav_oneread_finish": (C.c_int, [_P, _P]),
    "dpl_octav_oneread_compaction": (C.c_int, [_P, _P]),
    "dpl_octav_run_oneread": (C.c_int, [_P, _P]),
    "dpl_octav_list_cap": (C.c_uint32, [_U64]),
    "dpl_octav_plan_create": (_P, [_P, _I64, _I64, _I64]),
    "dpl_octav_plan_destroy": (None, [_P]),
    "dpl_octav_plan_sizes": (C.c_int, [_P, _P]),
    "dpl_octav_plan_upload": (C.c_int, [_P, _P, _P]),
    "dpl_octav_fallback_layout": (_I64, [_P, _I64, _P]),
    "dpl_octav_plan_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int, C.c_int, _P]),
    "dpl_test_hook_exact_fail_every": (C.c_int, [C.c_int]),
    "dpl_test_hook_rescue_fail_every": (C.c_int, [C.c_int]),
    "dpl_octav_finalize": (C.c_int, [_P, _I64, _P, _P]),
    "dpl_rowwise_minmax": (C.c_int, [_P, _I64, _I64, _P, _P, _P]),
    "dpl_fake_quant": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "dpl_fake_quant_pre": (C.c_int, [_I32, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _P]),
    "dpl_fake_quant_items": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _P, _P]),
    "dpl_cos_accumulate": (C.c_int, [_P, _P, _I64, _P, _I64, _P]),
    "dpl_channel_diff_sum": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "dpl_cos_items_accumulate": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _P, _P]),
    "dpl_gemm_small_workspace": (_U64, [_I64, _I64, _I64]),
    "dpl_gemm_small": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, C.c_float, C.c_float, _P, _P]),
    "dpl_round_init": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "dpl_round_quant": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, C.c_int, C.c_int, _P, _P]),
    "dpl_round_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, C.POINTER(RoundStepParams), _P, _P,
                                 _P, _P, _P]),
    "dpl_round_sched_advance": (C.c_int, [_P, _I32, _DBL, _DBL, _DBL, _P]),
    "dpl_sparse_quant": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, C.c_int, _P, _P]),
    "dpl_sparse_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_int, _P, _P]),
    "dpl_l2_loss": (C.c_int, [_P, _P, _I64, C.c_int, C.c_float, _DBL, _P, _P, _P]),
    "dpl_acti_drop_fwd": (C.c_int, [_P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "dpl_acti_drop_bwd": (C.c_int, [_P, _P, _I64, C.c_float, _P, _P]),
}

_lib = None


class DipoorletHipError(RuntimeError):
    pass


def lib():
    """The loaded shared library (cached).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DipoorletHipError(
                f"{LIB_PATH} is missing: build it with `python -m dipoorlet_amd.csrc.build` "
                "(or __graft_entry__.build()).  dipoorlet_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.dpl_abi_version() != ABI_VERSION:
            raise DipoorletHipError(f"ABI mismatch: library {l.dpl_abi_version()} != binding {ABI_VERSION}")
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        raise DipoorletHipError(f"{what} failed ({status}): {lib().dpl_last_error().decode()}")


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_uint64(0)
    st = lib().dpl_device_info(name, 256, C.byref(cus), C.byref(mem))
    return st, name.value.decode(), cus.value, mem.value


def _span_array(spans):
    spans = list(spans)
    arr = (Span * max(len(spans), 1))()
    for i, (seg, off, cnt, slot) in enumerate(spans):
        arr[i] = Span(off, cnt, seg, slot)
    return arr, len(spans)


def build_balanced_items(spans, n_blocks):
    """HOST: spans = iterable of (seg, offset, count, slot) -> (WorkItem array, n_items, block_begin array):
    n_blocks contiguous equal shares of the concatenated element stream."""
    arr, ns = _span_array(spans)
    n = lib().dpl_build_balanced_items(C.addressof(arr), ns, n_blocks, None, 0, None)
    if n < 0:
        check(int(n), "dpl_build_balanced_items")
    out = (WorkItem * max(n, 1))()
    bb = (C.c_uint32 * (n_blocks + 1))()
    n2 = lib().dpl_build_balanced_items(C.addressof(arr), ns, n_blocks, C.addressof(out), n, C.addressof(bb))
    assert n2 == n
    return out, int(n), bb


def build_octav_slices(spans):
    """HOST: spans (one per (image, tensor) pair, slots 0 .. n-1) -> (WorkItem array, n_slices, pair_slice0 uint32 [n, 2]),
    largest pairs first, or None when a pair is too large for the one-read form (more than 64 slices)."""
    arr, ns = _span_array(spans)
    n = lib().dpl_build_octav_slices(C.addressof(arr), ns, None, 0, None)
    if n == -3:
        return None
    if n < 0:
        check(int(n), "dpl_build_octav_slices")
    out = (WorkItem * max(n, 1))()
    ps = (C.c_uint32 * (2 * max(ns, 1)))()
    n2 = lib().dpl_build_octav_slices(C.addressof(arr), ns, C.addressof(out), n, C.addressof(ps))
    assert n2 == n
    return out, int(n), ps


def build_work_items(spans, chunk_elems):
    """HOST: spans = iterable of (seg, offset, count, slot) -> ctypes array of WorkItem."""
    spans = list(spans)
    arr = (Span * max(len(spans), 1))()
    for i, (seg, off, cnt, slot) in enumerate(spans):
        arr[i] = Span(off, cnt, seg, slot)
    n = lib().dpl_build_work_items(C.addressof(arr), len(spans), chunk_elems, None, 0)
    if n < 0:
        check(int(n), "dpl_build_work_items")
    out = (WorkItem * max(n, 1))()
    n2 = lib().dpl_build_work_items(C.addressof(arr), len(spans), chunk_elems, C.addressof(out), n)
    assert n2 == n
    return out, int(n)
