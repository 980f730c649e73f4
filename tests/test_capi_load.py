"""CPU: the C-ABI library builds, loads and exports every symbol include/dipoorlet_hip.h declares;
host-only entry points behave.  No device compute here."""
import ctypes as C
import os
import re

import pytest

from dipoorlet_amd import _hip
from dipoorlet_amd.csrc import build as hipbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hipbuild.build()
    return _hip.lib()


def test_header_symbols_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "dipoorlet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dpl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dpl_abi_version() == _hip.ABI_VERSION


def test_struct_layouts_match_header():
    assert C.sizeof(_hip.Span) == 24
    assert C.sizeof(_hip.WorkItem) == 24
    assert C.sizeof(_hip.HistRange) == 32
    assert C.sizeof(_hip.OctavState) == 80


def test_build_work_items_host(lib):
    spans = [(0, 0, 5000, 0), (1, 16, 1024, 1), (2, 0, 0, 2), (3, 7, 2049, 3)]
    arr, n = _hip.build_work_items(spans, 2048)
    got = [(arr[i].seg, arr[i].offset, arr[i].count, arr[i].slot) for i in range(n)]
    assert got == [(0, 0, 2048, 0), (0, 2048, 2048, 0), (0, 4096, 904, 0), (1, 16, 1024, 1),
                   (3, 7, 2048, 3), (3, 2055, 1, 3)]
    # every element covered exactly once
    for seg, off, cnt, slot in spans:
        cov = sorted((o, c) for s, o, c, sl in got if s == seg)
        pos = off
        for o, c in cov:
            assert o == pos
            pos += c
        assert pos == off + cnt
    with pytest.raises(_hip.DipoorletHipError):
        _hip.build_work_items(spans, 1000)  # not a multiple of 1024


def test_build_balanced_items_host(lib):
    import random
    rnd = random.Random(4)
    for trial in range(30):
        spans = [(i, rnd.choice([0, 16, 4096]), rnd.choice([0, 1, 1000, 2048, 25088, 401408, 802816 * 16]), 100 + i)
                 for i in range(rnd.randint(1, 12))]
        nb = rnd.choice([1, 2, 7, 64, 512])
        arr, n, bb = _hip.build_balanced_items(spans, nb)
        items = [(arr[i].seg, arr[i].offset, arr[i].count, arr[i].slot) for i in range(n)]
        assert bb[0] == 0 and bb[nb] == n and all(bb[i] <= bb[i + 1] for i in range(nb))
        # every element of every span exactly once, in order, slots preserved
        for seg, off, cnt, slot in spans:
            pos = off
            for s_, o, c, sl in items:
                if s_ == seg:
                    assert o == pos and sl == slot and c > 0
                    pos += c
            assert pos == off + cnt
        # shares are balanced to within one aligned piece per span boundary
        total = sum(c for _, _, c, _ in spans)
        share = [sum(items[k][2] for k in range(bb[b], bb[b + 1])) for b in range(nb)]
        assert sum(share) == total
        if total >= nb * 8192:
            assert max(share) <= total / nb + 1024 * (len(spans) + 1)
        # cuts inside a span are 4 KiB aligned relative to the span start
        for s_, o, c, sl in items:
            base = [sp for sp in spans if sp[0] == s_][0][1]
            assert (o - base) % 1024 == 0


def test_build_octav_slices_host(lib):
    """HOST side of the one-read OCTAV form: every pair cut into ceil(count / cap) equal slices on multiples of 4 elements,
    largest pairs first, pair_slice0 = the pair's contiguous slice range; a pair beyond 64 slices refuses the form."""
    import random
    rnd = random.Random(9)
    cap = _hip.lib().dpl_octav_slice_cap()
    assert cap % 4096 == 0 and _hip.lib().dpl_octav_list_cap(cap) == cap // 16 + 16384 and _hip.lib().dpl_octav_list_cap(20480) >= 20480
    assert _hip.lib().dpl_octav_small_pair() == 20480
    for trial in range(20):
        n = rnd.randint(1, 40)
        spans = [(i % 5, 1000 * i, rnd.choice([0, 1, 3, 777, 20480, 20481, 401408, cap, cap + 1, 802816, 3 * cap + 5]), i)
                 for i in range(n)]
        arr, ns, ps = _hip.build_octav_slices(spans)
        items = [(arr[i].seg, arr[i].offset, arr[i].count, arr[i].slot, arr[i].reserved) for i in range(ns)]
        sizes_seen = []
        for seg, off, cnt, slot in spans:
            lo, hi = ps[2 * slot], ps[2 * slot + 1]
            want = 0 if cnt == 0 else -(-cnt // cap)
            assert hi - lo == want
            pos = off
            for k in range(lo, hi):
                s_, o, c, sl, res = items[k]
                assert (s_, o, sl, res) == (seg, pos, slot, want) and 0 < c <= cap
                assert (o - off) % 4 == 0
                if k + 1 < hi:
                    assert c % 4 == 0 and c == items[lo][2]          # equal slices; only the last one takes the remainder
                pos += c
            assert pos == off + cnt
            if want:
                sizes_seen.append((lo, cnt))
        order = [c for _, c in sorted(sizes_seen)]
        assert order == sorted(order, reverse=True)                   # largest pairs first
    assert _hip.build_octav_slices([(0, 0, 65 * cap, 0)]) is None


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from dipoorlet_amd import ops
    with pytest.raises(_hip.DipoorletHipError):
        ops.rowwise_minmax(torch.zeros(4, 4))


def test_header_compiles_as_plain_c_and_cpp():
    """include/dipoorlet_hip.h is the drop-in boundary: it must be consumable by a C compiler (cgo / JNI / ctypes
    style FFI generators) as well as from C++."""
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dipoorlet_hip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", hdr], check=True)
    subprocess.run(["g++", "-fsyntax-only", "-x", "c++", "-Wall", "-Werror", hdr], check=True)


def test_oneread_job_struct_layout_matches_the_header(tmp_path):
    """dpl_octav_oneread_job crosses the boundary by pointer: the ctypes mirror must have the C compiler's layout — every
    field's offset and the total size, taken from a C program that includes the header."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    fields = [f[0] for f in _hip.OctavOnereadJob._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dipoorlet_hip.h"\nint main(void) {\n'
                   + "".join(f'  printf("{f} %zu\\n", offsetof(dpl_octav_oneread_job, {f}));\n' for f in fields)
                   + '  printf("sizeof %zu\\n", sizeof(dpl_octav_oneread_job));\n'
                   + '  printf("state %zu %zu %zu\\n", sizeof(dpl_octav_state), offsetof(dpl_octav_state, mode), offsetof(dpl_octav_state, len));\n'
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(line.split(None, 1) for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for f in fields:
        assert int(out[f]) == getattr(_hip.OctavOnereadJob, f).offset, f
    assert int(out["sizeof"]) == C.sizeof(_hip.OctavOnereadJob)
    assert out["state"].split() == [str(C.sizeof(_hip.OctavState)), str(_hip.OctavState.mode.offset), str(_hip.OctavState.len0.offset)]


def test_integration_md_job_struct_matches_the_binding():
    """INTEGRATION.md §B writes dpl_octav_oneread_job out field by field for a maintainer who binds the C ABI by hand: the
    listing must be the binding's (which test_job_struct_layout holds to the C compiler's)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = text[text.index("class Job(ctypes.Structure):"):text.index("assert ctypes.sizeof(Job)")]
    listed = re.findall(r'\("(\w+)", (P|I64|I32|F32)\)', block)
    kinds = {"P": C.c_void_p, "I64": C.c_int64, "I32": C.c_int32, "F32": C.c_float}
    assert [(n, kinds[k]) for n, k in listed] == [(f[0], f[1]) for f in _hip.OctavOnereadJob._fields_]
    assert "sizeof(Job) == %d" % C.sizeof(_hip.OctavOnereadJob) in text


def test_octav_plan_host_side(tmp_path):
    """dpl_octav_plan_* (HOST): sizes follow from the spans alone — list regions of dpl_octav_list_cap(n) values, whole-pair
    regions only in the compaction route's lists, histogram rows only for the slices of multi-slice pairs — and the ctypes
    mirror of dpl_octav_workspace_sizes has the C compiler's layout."""
    import shutil
    import subprocess
    L = _hip.lib()
    cap = L.dpl_octav_slice_cap()
    assert L.dpl_octav_list_cap(1000) == 1024 and L.dpl_octav_list_cap(20480) == 20480 and L.dpl_octav_list_cap(20481) == 17664 and L.dpl_octav_list_cap(802816) == 50176 + 16384
    assert all(L.dpl_octav_list_cap(n) % 32 == 0 and L.dpl_octav_list_cap(n) <= (n + 31) // 32 * 32 for n in (1, 31, 33, 4097, 10**6))
    T, B = 3, 2
    elems = [1000, 802816, 2 * cap + 4096]                    # a small pair, a single-slice pair, a pair of three slices
    arr, ns = _hip._span_array([(t, b * e, e, b * T + t) for b in range(B) for t, e in enumerate(elems)])
    plan = L.dpl_octav_plan_create(C.addressof(arr), ns, T, 64)
    assert plan
    z = _hip.OctavWorkspaceSizes()
    assert L.dpl_octav_plan_sizes(plan, C.byref(z)) == 0
    per3 = (((elems[2] + 2) // 3) + 3) & ~3
    want_list = B * (L.dpl_octav_list_cap(1000) + L.dpl_octav_list_cap(802816) + 3 * L.dpl_octav_list_cap(per3))
    assert z.list_bytes == 4 * want_list
    assert z.fallback_bytes == 2 * 4 * B * sum((e + 31) // 32 * 32 for e in elems)
    assert (z.n_pairs, z.n_slices, z.n_multi, z.n_small) == (B * T, B * 5, B, B)
    assert z.history_bytes == 4 * 2 * T * 64 and z.result_bytes == 4 * 3 * B * T
    assert z.rescue_bytes >= 8 * 2048 * (B * 3) + 8 * 3072 * B * T and z.rescue_bytes < 8 * 2048 * (B * 3) + 8 * 3072 * B * T + 4096 + 4 * 67 * B * T
    # a job bound to fake addresses: the tables' pointers fall inside the tables block, the lists where they were put
    job = _hip.OctavOnereadJob()
    base = 1 << 40
    st = L.dpl_octav_plan_bind(plan, base, base + (1 << 30), base + (2 << 30), base + (3 << 30), base + (4 << 30), base + (5 << 30), None,
                               base + (6 << 30), 9, 1, 20, C.byref(job))
    assert st == 0 and job.compaction_inline == 0 and job.d_clist0 is None
    assert (job.write_epoch, job.reset_epoch, job.dynamic_sym, job.max_iters) == (1, 0, 1, 20)
    for f in ("d_slices", "d_pair_slice0", "d_pair_spans", "d_pair_base", "d_pair_base_full", "d_pair_order", "d_items", "d_block_begin"):
        assert base <= getattr(job, f) < base + z.tables_bytes, f
    assert job.d_vis == base + (1 << 30) and job.d_states == base + (2 << 30) and job.d_rescue_bm == base + (3 << 30)
    assert job.d_list0 == base + (4 << 30) and job.d_list1 == base + (5 << 30) and job.d_seg_ptrs == base + (6 << 30)
    assert L.dpl_octav_plan_bind(plan, base, base, base, base, base, base, base + (7 << 30), base, 16, 0, 20, C.byref(job)) == 0
    assert job.compaction_inline == 1 and job.d_clist1 - job.d_clist0 == z.fallback_bytes // 2 and (job.write_epoch, job.reset_epoch) == (0, 1)
    L.dpl_octav_plan_destroy(plan)
    # more than 64 slices: no plan (the two-read form serves such a set)
    big, _ = _hip._span_array([(0, 0, 65 * cap, 0)])
    assert not L.dpl_octav_plan_create(C.addressof(big), 1, 1, 64)
    assert b"slices" in L.dpl_last_error()
    if shutil.which("gcc") is None:
        return
    fields = [f[0] for f in _hip.OctavWorkspaceSizes._fields_]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dipoorlet_hip.h"\nint main(void) {\n'
                   + "".join(f'  printf("{f} %zu\\n", offsetof(dpl_octav_workspace_sizes, {f}));\n' for f in fields)
                   + '  printf("sizeof %zu\\n", sizeof(dpl_octav_workspace_sizes));\n  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for f in fields:
        assert int(out[f]) == getattr(_hip.OctavWorkspaceSizes, f).offset, f
    assert int(out["sizeof"]) == C.sizeof(_hip.OctavWorkspaceSizes)


def test_octav_fallback_layout_host():
    """dpl_octav_fallback_layout: whole-pair regions (rounded up to 32 values) for the pairs a batch left on the compaction route
    (mode 1, not done), nothing for the others."""
    import numpy as np
    L = _hip.lib()
    n = 6
    st = (_hip.OctavState * (n + 1))()
    for i, (mode, done, elems) in enumerate([(2, 1, 1000), (1, 0, 1000), (3, 0, 50), (1, 1, 77), (1, 0, 33), (0, 0, 9)]):
        st[i].mode, st[i].done, st[i].n_elems = mode, done, elems
    base = np.full(n + 1, 99, np.uint64)
    assert L.dpl_octav_fallback_layout(C.addressof(st), n, base.ctypes.data) == 1024 + 64
    assert base.tolist() == [0, 0, 1024, 1024, 1024, 1088, 1088]
    assert L.dpl_octav_fallback_layout(None, n, base.ctypes.data) < 0


def _integration_md_example():
    """The python block of INTEGRATION.md §B that binds `-A mse` by hand (ctypes + torch for device memory, nothing of this
    package), as a namespace."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    i = text.index("import ctypes, torch\nP, I64, I32, F32, U64")
    code = text[i:text.index("```", i)]
    assert len([ln for ln in code[code.index("def octav_rows"):].splitlines() if ln.strip() and not ln.strip().startswith("#")]) <= 60
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_md_octav_example_compiles():
    ns = _integration_md_example()
    assert C.sizeof(ns["Job"]) == C.sizeof(_hip.OctavOnereadJob) and C.sizeof(ns["Sizes"]) == C.sizeof(_hip.OctavWorkspaceSizes)
    assert [f[0] for f in ns["Sizes"]._fields_] == [f[0] for f in _hip.OctavWorkspaceSizes._fields_]


@pytest.mark.gpu
def test_integration_md_octav_example_runs(golden_dir):
    """A maintainer's binding of `-A mse` — INTEGRATION.md §B's block, verbatim: dpl_octav_plan_* size and lay out the
    workspace, dpl_octav_run_oneread + dpl_octav_finalize run the batch — against the reference's own values
    (tests/golden/kernel_level.*: forward_net.py:315-330 under the Appendix-A stubs), trt and ti (dynamic_sym)."""
    import json

    import numpy as np
    import torch

    from _cases import make_tensor
    ns = _integration_md_example()
    with open(os.path.join(golden_dir, "kernel_level.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(golden_dir, "kernel_level.npz"))
    cases = meta["cases"]
    for dyn, deploy in ((0, "trt"), (1, "ti")):
        # one batch of two images over all the cases' tensors: image 0 = the golden tensor, image 1 = the same values reversed
        xs = [make_tensor(c["kind"], c["n"], c["seed"]) for c in cases]
        tensors = [torch.from_numpy(np.stack([x, x[::-1].copy()])).to("cuda") for x in xs]
        rows = ns["octav_rows"](_hip.LIB_PATH, tensors, dyn).cpu().numpy()
        for t, c in enumerate(cases):
            want = g[f"{c['key']}/octav_{deploy}"]
            for b in range(2):
                got = rows[b, t]
                assert abs(got[0] - want[0]) <= 1e-5 * max(1.0, abs(want[0])) or (np.isnan(got[0]) and np.isnan(want[0])), (c["key"], deploy, b, got, want)
                assert np.array_equal(got[1:], want[1:], equal_nan=True), (c["key"], deploy, b, got, want)
