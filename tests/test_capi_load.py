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
    assert cap % 4096 == 0 and _hip.lib().dpl_octav_sort_chunk() == 8192 and _hip.lib().dpl_octav_dir_row() % 8 == 0
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
