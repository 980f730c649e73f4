"""Pins the CPU oracle (oracle/np_oracle.py) to the golden vectors produced by the reference's own
code (tests/golden/gen_golden.py).  CPU only."""
import json
import os
import warnings

import numpy as np
import pytest

from _cases import MINI_NET, checksum, make_tensor, mini_net_activations
from oracle import np_oracle as O


@pytest.fixture(scope="module")
def kl(golden_dir):
    with open(os.path.join(golden_dir, "kernel_level.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(golden_dir, "kernel_level.npz"))


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_case_data_regenerates(kl):
    meta, _ = kl
    for c in meta["cases"]:
        assert checksum(make_tensor(c["kind"], c["n"], c["seed"])) == c["crc"], c["key"]


def test_minmax_bit_exact(kl):
    meta, g = kl
    for c in meta["cases"]:
        x = make_tensor(c["kind"], c["n"], c["seed"])
        lo, hi = O.minmax(x)
        ref = g[c["key"] + "/minmax"]
        if c["kind"] == "with_nan":
            assert np.isnan(ref).all() and np.isnan([lo, hi]).all()  # numpy max / min propagate NaN
        else:
            assert np.array_equal(_bits([lo, hi]), _bits(ref)), c["key"]


def test_abs_hist_bit_exact_and_percentile(kl):
    meta, g = kl
    for c in meta["cases"]:
        x = make_tensor(c["kind"], c["n"], c["seed"])
        gmin0, gmax0 = g[c["key"] + "/minmax"]
        if c["kind"] == "with_nan":  # the reference's histogram pass raises on the NaN range; so does the oracle
            assert c["hist_raises"]
            with pytest.raises(ValueError):
                O.abs_hist(x, 2048, O.hist_dmax(gmin0, gmax0))
            continue
        for bins in (2048, 1000):
            for scale in (1.0, 1.5):
                gmin, gmax = np.float32(gmin0 * np.float32(scale)), np.float32(gmax0 * np.float32(scale))
                tag = f"{c['key']}/hist_b{bins}_s{scale}"
                h = O.abs_hist(x, bins, O.hist_dmax(gmin, gmax))
                assert h.dtype == np.int64 and np.array_equal(h, g[tag]), tag
                for thr in (0.99999, 0.999):
                    clip = O.hist_percentile(g[tag], gmin, gmax, bins, thr)
                    assert np.array_equal(_bits(clip), _bits(g[f"{tag}_clip{thr}"])), (tag, thr)


def test_abs_hist_matches_numpy_histogram_directly():
    rng = np.random.default_rng(3)
    for bins in (2048, 1000, 777, 64):
        for _ in range(4):
            x = (rng.standard_normal(50000) * rng.uniform(0.01, 30)).astype(np.float32)
            dmax = np.float32(np.abs(x).max() * rng.choice([1.0, 1.3, 0.6]))
            ref, _ = np.histogram(np.abs(x), bins, (0, dmax))
            assert np.array_equal(O.abs_hist(x, bins, dmax), ref)
    x = np.array([0, 0, np.nan, 1, -1, 0.5], np.float32)
    ref, _ = np.histogram(np.abs(x), 2048, (0, np.float32(1)))
    assert np.array_equal(O.abs_hist(x, 2048, np.float32(1)), ref)
    assert ref.sum() == 5  # NaN dropped


def test_octav_scale(kl):
    meta, g = kl
    for c in meta["cases"]:
        x = make_tensor(c["kind"], c["n"], c["seed"])
        for deploy, dyn in (("trt", False), ("ti", True)):
            ref = g[f"{c['key']}/octav_{deploy}"]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                s = O.octav_scale(x, O.octav_unsigned(ref[1], dyn))
            assert np.array_equal(_bits(s), _bits(ref[0])) or (np.isnan(s) and np.isnan(ref[0])), c["key"]


def test_pipeline_statistics_and_clips(golden_dir):
    with open(os.path.join(golden_dir, "pipeline_level.json")) as f:
        pl = json.load(f)
    st = np.load(os.path.join(golden_dir, "pipeline_stats.npz"))
    N = pl["N"]
    acts = [dict(mini_net_activations(i)) for i in range(N)]
    names = [n for n, _, _ in MINI_NET]
    # per-image statistics seam
    for k in names:
        mins = [O.minmax(a[k])[0] for a in acts]
        maxs = [O.minmax(a[k])[1] for a in acts]
        assert np.array_equal(_bits(mins), _bits(st[f"{k}/min"]))
        assert np.array_equal(_bits(maxs), _bits(st[f"{k}/max"]))
        dmax = O.hist_dmax(np.min(mins), np.max(maxs))
        hs = np.stack([O.abs_hist(a[k], 2048, dmax) for a in acts])
        assert np.array_equal(hs, st[f"{k}/hist"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = [O.octav_scale(a[k], O.octav_unsigned(mn, True)) for a, mn in zip(acts, mins)]
        assert np.array_equal(_bits(s), _bits(st[f"{k}/octav_s_ti"]), equal_nan=False) or \
            np.array_equal(np.isnan(s), np.isnan(st[f"{k}/octav_s_ti"]))
    # registry algorithms, per rank, and the reference's JSON merge
    from dipoorlet_amd.platform_settings import platform_setting_table
    for run in pl["runs"]:
        W = run["world_size"]
        per_rank = []
        for r in range(W):
            b, e = O.shard_range(N, r, W)
            clip = {}
            for k in names:
                mins = [O.minmax(acts[i][k])[0] for i in range(b, e)]
                maxs = [O.minmax(acts[i][k])[1] for i in range(b, e)]
                if run["algo"] == "minmax":
                    clip[k] = O.clip_minmax(mins, maxs)
                elif run["algo"] == "hist":
                    gmin, gmax = np.min(mins), np.max(maxs)
                    h = sum(O.abs_hist(acts[i][k], run["bins"], O.hist_dmax(gmin, gmax)) for i in range(b, e))
                    clip[k] = O.hist_percentile(h, gmin, gmax, run["bins"], run["threshold"])
                else:
                    dyn = "dynamic_sym" in platform_setting_table[run["deploy"]]["qi_params"]
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        s = [O.octav_scale(acts[i][k], O.octav_unsigned(mn, dyn))
                             for i, mn in zip(range(b, e), mins)]
                        clip[k] = O.octav_clip(s, mins, maxs)
                ref = run["ranks"][r][k]
                assert np.array_equal(_bits(clip[k]), _bits(ref)), (run, r, k, clip[k], ref)
            per_rank.append({k: [float(v[0]), float(v[1])] for k, v in clip.items()})
        merged = O.reduce_clip_val(per_rank, run["algo"])
        for k in names:
            assert merged[k][0] == run["merged"][k][0] and merged[k][1] == run["merged"][k][1], (run, k)


def test_qparams_all_platforms(golden_dir):
    from dipoorlet_amd.platform_settings import platform_setting_table
    with open(os.path.join(golden_dir, "qparam_level.json")) as f:
        q = json.load(f)
    for row in q["rows"]:
        param = platform_setting_table[row["platform"]][row["param"]]
        lo = np.array(row["lo"]) if row["per_channel_in"] else row["lo"][0]
        hi = np.array(row["hi"]) if row["per_channel_in"] else row["hi"][0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scale, zp, qmin, qmax, sym = O.qparams(param, lo, hi)
        assert np.array_equal(_bits(scale), _bits(row["scale"])), row
        assert zp.astype(np.int64).tolist() == row["zp"], row
        assert [int(v) for v in qmin] == [int(v) for v in row["qmin"]], row
        assert [int(v) for v in qmax] == [int(v) for v in row["qmax"]], row
        assert sym == row["symmetric"], row


def test_weight_minmax_and_quant_acti(golden_dir):
    with open(os.path.join(golden_dir, "qparam_level.json")) as f:
        q = json.load(f)
    g = np.load(os.path.join(golden_dir, "qparam_level.npz"))
    assert q["weight_keys"] == ["conv.b", "conv.w", "deconv.w", "gemm.w"]  # 0-d BN scalar skipped
    for k in q["weight_keys"]:
        lo, hi = O.rowwise_minmax(g[f"w/{k}"], transpose=(k == "deconv.w"))
        assert np.array_equal(_bits(lo), _bits(g[f"wmin/{k}"]))
        assert np.array_equal(_bits(hi), _bits(g[f"wmax/{k}"]))
    x = g["qa/x"]
    for i in range(4):
        scale, qlo, qhi = g[f"qa/{i}/p"]
        assert np.array_equal(_bits(O.quant_acti(x, scale, qlo, qhi)), _bits(g[f"qa/{i}/y"]))
    # symmetric int8 Q/DQ (zp = 0) inside the clamp range coincides with quant_acti on [-127, 127]
    y = O.fake_quant_qdq(x, np.float32(0.05), 0, signed=True)
    ya = O.quant_acti(x, 0.05, -128, 127)
    assert np.array_equal(y, ya)  # values; Q/DQ yields +0.0 where quant_acti keeps -0.0


def test_c_oracle_bit_exact_against_goldens(kl):
    """The plain-C restatement (oracle/c_oracle.c) is held to the same reference-generated vectors."""
    from oracle import c_oracle as CO
    meta, g = kl
    for c in meta["cases"]:
        x = make_tensor(c["kind"], c["n"], c["seed"])
        gmin0, gmax0 = g[c["key"] + "/minmax"]
        if c["kind"] == "with_nan":
            assert np.isnan(CO.minmax(x)).all()
            with pytest.raises(ValueError):
                CO.abs_hist(x, 2048, O.hist_dmax(gmin0, gmax0))
            assert np.isnan(CO.octav_scale(x, 1))
            continue
        assert np.array_equal(_bits(CO.minmax(x)), _bits([gmin0, gmax0])), c["key"]
        for bins in (2048, 1000):
            for scale in (1.0, 1.5):
                gmin, gmax = np.float32(gmin0 * np.float32(scale)), np.float32(gmax0 * np.float32(scale))
                assert np.array_equal(CO.abs_hist(x, bins, O.hist_dmax(gmin, gmax)), g[f"{c['key']}/hist_b{bins}_s{scale}"])
        for deploy, dyn in (("trt", False), ("ti", True)):
            ref = g[f"{c['key']}/octav_{deploy}"]
            s = CO.octav_scale(x, O.octav_unsigned(ref[1], dyn))
            assert np.array_equal(_bits(s), _bits(ref[0])) or (np.isnan(s) and np.isnan(ref[0])), (c["key"], s, ref[0])
    xs = [make_tensor("relu", 50000, 1), make_tensor("normal", 7000, 2)]
    used, mins, maxs, s, hist = CO.batch(xs, "hist", 2048, threads=2)
    assert used >= 1 and np.array_equal(hist[0], O.abs_hist(xs[0], 2048, O.hist_dmax(mins[0], maxs[0])))
