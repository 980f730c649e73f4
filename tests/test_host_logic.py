"""CPU: host-side mirror of the reference interfaces — dispatcher, clip JSON exchange (byte-compatible
with the reference's files), scale/zero-point derivation, sharding, and the world_size-2 merge over gloo."""
import json
import os

import pytest
import types
import warnings

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _cases import MINI_NET, mini_net_activations
from oracle import np_oracle as O


def _args(**kw):
    a = types.SimpleNamespace(rank=0, world_size=1, deploy="trt", act_quant="hist", output_dir=None)
    a.__dict__.update(kw)
    return a


def test_dispatcher_semantics():
    from dipoorlet_amd.utils import dispatch_functool

    calls = []

    @dispatch_functool
    def disp(*a, **k):
        calls.append(("default", a, k))

    @disp.register("x")
    def fx(g, args, **k):
        return ("x", g, args, k)

    assert disp("x", 1, 2, store_stats=3) == ("x", 1, 2, {"store_stats": 3})
    assert disp("nope", 1, 2) is None and calls == [("default", (1, 2), {})]  # unknown key -> default
    assert disp.dispatch("x") is fx and "x" in disp.registry


def test_registry_keys_match_reference():
    from dipoorlet_amd.tensor_cali import tensor_cali_dispatcher
    assert set(tensor_cali_dispatcher.registry) == {"minmax", "hist", "mse"}
    assert tensor_cali_dispatcher("kl", None, None) is None  # "Calibration Algorithm Not Found!"


def test_clip_json_roundtrip_is_byte_compatible(golden_dir, tmp_path):
    """save -> per-rank files -> reduce -> load reproduces the reference's merged act_clip_val.json text."""
    from dipoorlet_amd.utils import load_clip_val, reduce_clip_val, save_clip_val
    with open(os.path.join(golden_dir, "pipeline_level.json")) as f:
        pl = json.load(f)
    for run in pl["runs"]:
        od = tmp_path / f"{run['algo']}_{run['deploy']}_{run['world_size']}_{run['bins']}"
        od.mkdir()
        a = _args(output_dir=str(od), act_quant=run["algo"], deploy=run["deploy"])
        for r, clips in enumerate(run["ranks"]):
            act = {k: [np.float32(v[0]), np.float32(v[1])] for k, v in clips.items()}
            save_clip_val(act, {}, a, act_fname=f"act_clip_val.json.rank{r}", weight_fname=f"weight_clip_val.json.rank{r}")
        reduce_clip_val(run["world_size"], a)
        assert (od / "act_clip_val.json").read_text() == run["merged_json_text"]
        act, wt = load_clip_val(a)
        for k, v in run["merged"].items():
            assert isinstance(act[k][0], np.float64) and act[k][0] == v[0] and act[k][1] == v[1]


def test_get_qnode_by_param_all_platforms(golden_dir):
    from dipoorlet_amd.platform_settings import platform_setting_table
    from dipoorlet_amd.quantize import get_qnode_by_param
    with open(os.path.join(golden_dir, "qparam_level.json")) as f:
        q = json.load(f)
    for row in q["rows"]:
        param = platform_setting_table[row["platform"]][row["param"]]
        if row["per_channel_in"]:
            rng = [np.array(row["lo"]), np.array(row["hi"])]
        else:
            rng = [np.float64(row["lo"][0]), np.float64(row["hi"][0])]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            node, qmin, qmax = get_qnode_by_param(param, "t", [5, 3], rng)
        assert np.array_equal(node.scale.view(np.uint32), np.array(row["scale"], np.float32).view(np.uint32)), row
        assert node.zero_point.astype(np.int64).tolist() == row["zp"], row
        assert [int(v) for v in np.ravel(qmin)] == [int(v) for v in row["qmin"]], row
        assert [int(v) for v in np.ravel(qmax)] == [int(v) for v in row["qmax"]], row
        assert node.symmetric == row["symmetric"] and node.per_channel == row["per_channel"]
        assert node.output == "t_dq" and node.q_name == "t_QuantizeLinear" and node.scale_name == "t_scale"
    # the documented latent wrap: snpe, range [-3, 1] -> zp 191 stored as int8 -65, read back as uint8 191
    node, qmin, qmax = get_qnode_by_param(platform_setting_table["snpe"]["qi_params"], "t", [1],
                                          [np.float64(-3.0), np.float64(1.0)])
    assert node.zero_point.tolist() == [-65] and node.zero_point_as_stored().tolist() == [191]
    assert (qmin, qmax) == ([-191], [64]) and node.saturation() == (0, 255)


def test_shard_range_matches_reference_split():
    from dipoorlet_amd.dist_helper import shard_range, slurm_master_addr
    for n in (8, 10, 1024, 7):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert got == [O.shard_range(n, r, w) for r in range(w)]
            assert got[-1][1] == (n // w) * w  # the remainder images are dropped, as in the reference
    assert slurm_master_addr("SH-IDC1-10-5-30-[12-15]") == "10.5.30.12"


# ------------------------------------------------------------------ world_size 2 over gloo (CPU)
def _worker(rank, world, port, tmp, golden_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dipoorlet_amd import dist_helper as dh
    dh.init_default()
    assert dist.get_backend() == "gloo"
    N, bins = 8, 2048
    names = [n for n, _, _ in MINI_NET]
    st, ed = dh.shard_range(N, rank, world)
    acts = [dict(mini_net_activations(i)) for i in range(st, ed)]
    # this rank's statistics (oracle stands in for the kernels: no GPU in this container)
    gmin = torch.tensor([min(O.minmax(a[k])[0] for a in acts) for k in names])
    gmax = torch.tensor([max(O.minmax(a[k])[1] for a in acts) for k in names])
    dh.merge_ranges(gmin, gmax, world)
    hist = torch.from_numpy(np.stack([
        sum(O.abs_hist(a[k], bins, O.hist_dmax(gmin[t].numpy(), gmax[t].numpy())) for a in acts)
        for t, k in enumerate(names)]))
    dh.merge_hist(hist, world)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rows = torch.tensor([[[float(O.octav_scale(a[k], 1)), float(O.minmax(a[k])[0]), float(O.minmax(a[k])[1])]
                              for k in names] for a in acts], dtype=torch.float32)
    rows = dh.gather_rows(rows, world)
    torch.save({"gmin": gmin, "gmax": gmax, "hist": hist, "rows": rows}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_merge_equals_world1_reference(golden_dir, tmp_path):
    world = 2
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(world, port, str(tmp_path), golden_dir), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for k in ("gmin", "gmax", "hist", "rows"):
        assert torch.equal(r0[k], r1[k]) or (k == "rows" and torch.allclose(r0[k], r1[k], equal_nan=True))
    # equals the reference's own world_size = 1 answer over all 8 images
    with open(os.path.join(golden_dir, "pipeline_level.json")) as f:
        pl = json.load(f)
    st = np.load(os.path.join(golden_dir, "pipeline_stats.npz"))
    names = [n for n, _, _ in MINI_NET]
    ref = {(r["algo"], r["deploy"], r["bins"], r["world_size"]): r for r in pl["runs"]}
    mm = ref[("minmax", "trt", 2048, 1)]["ranks"][0]
    hh = ref[("hist", "trt", 2048, 1)]["ranks"][0]
    oo = ref[("mse", "trt", 2048, 1)]["ranks"][0]
    for t, k in enumerate(names):
        assert [r0["gmin"][t].item(), r0["gmax"][t].item()] == mm[k]
        assert np.array_equal(r0["hist"][t].numpy(), st[f"{k}/hist"].sum(0))
        clip = O.hist_percentile(r0["hist"][t].numpy(), r0["gmin"][t].numpy(), r0["gmax"][t].numpy(), 2048, 0.99999)
        assert [float(clip[0]), float(clip[1])] == hh[k]
        rows = r0["rows"][:, t].numpy()
        clip = O.octav_clip(rows[:, 0], rows[:, 1], rows[:, 2])
        assert [float(clip[0]), float(clip[1])] == oo[k]


def _launcher_worker(kind, port, q):
    """Runs in a fresh process: environment as OpenMPI / SLURM would set it -> the process group the helpers build."""
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from dipoorlet_amd import dist_helper as dh
    if kind == "mpi":
        os.environ.update(OMPI_MCA_orte_hnp_uri="123.0;tcp://127.0.0.1:5555", OMPI_COMM_WORLD_SIZE="1",
                          OMPI_COMM_WORLD_RANK="0", MASTER_PORT=str(port))
        dh.init_from_mpi()
    else:
        os.environ.update(SLURM_JOB_ID=str(port - 24553), SLURM_NODELIST="HOST-AB-127-0-0-1", SLURM_NTASKS="1",
                          SLURM_PROCID="0")
        dh.init_from_slurm()
    q.put((os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], dist.get_rank(), dist.get_world_size(),
           dist.get_backend()))
    dist.destroy_process_group()


def test_mpi_and_slurm_bootstrap():
    """dist_helper.py:8-49 — rank / size / master derived from the launcher's environment (gloo here: no GPU)."""
    ctx = mp.get_context("spawn")
    for kind, port in (("mpi", 29871), ("slurm", 24553 + 4321)):
        q = ctx.Queue()
        p = ctx.Process(target=_launcher_worker, args=(kind, port, q))
        p.start()
        addr, mport, rank, world, backend = q.get(timeout=120)
        p.join(60)
        assert p.exitcode == 0
        assert (addr, rank, world, backend) == ("127.0.0.1", 0, 1, "gloo")
        assert mport == str(port)


def test_deploy_emitters_match_the_reference_files(tmp_path, golden_dir):
    """Every deploy.py emitter against the files the reference's own emitters wrote for the same graph and ranges
    (tests/golden/deploy_level.json, from gen_golden_deploy.py)."""
    import yaml
    from dipoorlet_amd.deploy import to_deploy
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.onnx_io import Node
    with open(os.path.join(golden_dir, "deploy_level.json")) as f:
        G = json.load(f)

    def graph():
        g = ONNXGraph()
        g.graph.node = [Node(op, i, o, name=n, attrs=a) for op, i, o, n, a in G["nodes"]]
        g.initializer = {k: np.asarray(v, np.float32) for k, v in G["weights"].items()}
        g.initializer.update({b: np.zeros(1, np.float32) for b in ("b1", "b3", "b4", "b5")})
        g.network_inputs, g.network_outputs = ["input"], ["output"]
        g.tensor_name_shape_map = {k: list(v) for k, v in G["shapes"].items()}
        g.update_model()
        return g

    def clips():
        act = {k: [np.float64(v[0]), np.float64(v[1])] for k, v in G["act_clip"].items()}
        wt = {k: [np.array(v[0]), np.array(v[1])] for k, v in G["weight_clip"].items()}
        return act, wt

    def close(a, b, path=""):
        if isinstance(b, dict):
            assert isinstance(a, dict) and list(a) == list(b), path      # same keys in the same order
            for k in b:
                close(a[k], b[k], path + "/" + str(k))
        elif isinstance(b, list):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                close(x, y, f"{path}[{i}]")
        elif isinstance(b, float):
            assert a == pytest_approx(b), (path, a, b)
        else:
            assert a == b, (path, a, b)
    for tag, deploy, wg in (("rv", "rv", False), ("stpu", "stpu", False), ("stpu_wg", "stpu", True), ("trt", "trt", False),
                            ("snpe", "snpe", False), ("ti", "ti", False), ("imx", "imx", False),
                            ("magicmind", "magicmind", False), ("atlas", "atlas", False)):
        out = tmp_path / tag
        os.makedirs(out)
        act, wt = clips()
        to_deploy(graph(), act, wt, types.SimpleNamespace(deploy=deploy, output_dir=str(out), stpu_wg=wg))
        want = {k.split("/", 1)[1]: v for k, v in G["files"].items() if k.startswith(tag + "/")}
        assert sorted(os.listdir(out)) == sorted(want)
        for name, text in want.items():
            got = open(out / name).read()
            if name.endswith(".json"):
                close(json.loads(got), json.loads(text), name)
            elif name.endswith(".yaml"):
                close(yaml.safe_load(got), yaml.safe_load(text), name)
            else:
                assert got == text, name
        if tag != "stpu_wg":            # without winograd nothing is tolerance-dependent: byte-identical files
            for name, text in want.items():
                assert open(out / name).read() == text, name


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-6, abs=1e-12)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_dry_run_gloo(world):
    """bench.py --gpus W --dry-run (W = 2 and the node's 8): the launcher (fresh child processes before anything touches a
    GPU), the rendezvous on 127.0.0.1 and the collectives of one hist sweep (all-reduce MIN / MAX of the ranges, SUM of the
    histograms) on stand-in CPU tensors over gloo — the multi-GPU plumbing the driver's N > 1 runs go through."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_run"] is True and line["n_gpus"] == world and line["backend"] == "gloo"
    assert line["hist_checksum"] == line["hist_checksum_expected"] == 123 * 2048 * world * (world + 1) // 2
    assert line["range"] == [-float(world), float(world)]    # MIN / MAX over the ranks' ranges
    assert line["world_size_seen_by_backend"] == world and len(line["per_rank_ok"]) == world and all(line["per_rank_ok"])


def test_dump_indent4_is_json_dump_byte_for_byte():
    """utils.dump_indent4 — the writer of act_clip_val.json / weight_clip_val.json — yields exactly json.dumps(obj, indent=4)
    (the reference's format, utils.py:313-323) for nested dicts / lists of numbers, non-finite values, escapes, empty containers."""
    import io
    import json
    import random

    import numpy as np

    from dipoorlet_amd.utils import dump_indent4
    leaves = [0.0, -0.0, 1, -7, 2 ** 70, True, False, None, float("nan"), float("inf"), float("-inf"), "a\"b\\c\né", "",
              np.float64(1.5), 1e22, 1e-7, 123456789.123456789]

    def rnd(depth=0):
        r = random.random()
        if depth > 3 or r < 0.3:
            return random.choice(leaves + [random.uniform(-1e6, 1e6), random.gauss(0, 1) * 10 ** random.randint(-30, 30)])
        if r < 0.55:
            return [random.gauss(0, 1) for _ in range(random.randint(0, 6))]
        if r < 0.8:
            return [rnd(depth + 1) for _ in range(random.randint(0, 4))]
        return {random.choice(["k", "conv.w", "x y", "é", ""]) + str(i): rnd(depth + 1) for i in range(random.randint(0, 4))}
    random.seed(1)
    for _ in range(3000):
        o = rnd()
        b = io.StringIO()
        dump_indent4(o, b)
        assert b.getvalue() == json.dumps(o, indent=4), o
    clip = {f"w{i}": [np.random.default_rng(i).standard_normal(64).astype(np.float32).tolist(), (1.0, float(i))] for i in range(8)}
    b = io.StringIO()
    dump_indent4(clip, b)
    assert b.getvalue() == json.dumps(clip, indent=4)
    b = io.StringIO()
    dump_indent4({1: [1.0]}, b)                      # (anything unusual: json.dump itself)
    assert b.getvalue() == json.dumps({1: [1.0]}, indent=4)


def test_bc_shard_is_a_balanced_partition_of_all_images():
    """weight_transform.bias_correction.bc_shard: the sharded `--bc` walk covers EVERY image exactly once (the reference corrects
    with all N, forward_net.py:50-52), shards differ by at most one image; the calibration sweeps' floor split
    (dist_helper.shard_range, forward_net.py:207-209) drops N mod W."""
    from dipoorlet_amd.dist_helper import shard_range
    from dipoorlet_amd.weight_transform.bias_correction import bc_shard
    for n in (0, 1, 7, 8, 33, 1024, 2047):
        for w in (1, 2, 3, 8):
            parts = [bc_shard(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[r][1] == parts[r + 1][0] for r in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
            assert sum(b - a for a, b in (shard_range(n, r, w) for r in range(w))) == (n // w) * w
