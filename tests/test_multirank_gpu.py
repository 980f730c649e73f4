"""GPU: the real multi-rank path — two processes, real kernels, statistics merged with collectives — on the ONE
GPU of the test box (gloo transport, because RCCL refuses two ranks on one device; the code path above the
backend string is the one the 8-GPU run takes).  The merged result must equal the reference's own
world_size = 1 answer over all images (tests/golden/pipeline_level.json)."""
import json
import os
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, calib_dir, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DPL_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from test_calibration_e2e import MiniGraph
    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.tensor_cali import tensor_cali_dispatcher
    dist_helper.init_default()
    res = {}
    for algo, deploy in (("minmax", "trt"), ("hist", "trt"), ("mse", "trt"), ("mse", "ti")):
        args = types.SimpleNamespace(input_dir=calib_dir, data_num=8, rank=rank, local_rank=0, world_size=world,
                                     bins=2048, threshold=0.99999, deploy=deploy, act_quant=algo,
                                     optim_transformer=False, merge="allreduce", calib_batch=3)
        clip = tensor_cali_dispatcher(algo, MiniGraph(), args)
        res[f"{algo}_{deploy}"] = {k: [float(v[0]), float(v[1])] for k, v in clip.items()}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_merge_to_the_world1_reference(tmp_path, golden_dir):
    from _cases import mini_net_activations
    calib = tmp_path / "calib"
    os.makedirs(calib / "input")
    for i in range(8):
        mini_net_activations(i)[0][1].tofile(calib / "input" / f"{i}.bin")
    port = 29700 + os.getpid() % 200
    mp.spawn(_worker, args=(2, port, str(calib), str(tmp_path)), nprocs=2, join=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0 == r1  # every rank ends with the statistics of the whole set
    with open(os.path.join(golden_dir, "pipeline_level.json")) as f:
        runs = {(r["algo"], r["deploy"], r["bins"], r["world_size"]): r for r in json.load(f)["runs"]}
    for key, (algo, deploy) in (("minmax_trt", ("minmax", "trt")), ("hist_trt", ("hist", "trt")),
                                ("mse_trt", ("mse", "trt")), ("mse_ti", ("mse", "ti"))):
        ref = runs[(algo, deploy, 2048, 1)]["ranks"][0]
        for name, v in ref.items():
            if algo == "mse":
                assert np.allclose(r0[key][name], v, rtol=1e-5, atol=1e-5), (key, name, r0[key][name], v)
            else:
                assert r0[key][name] == v, (key, name, r0[key][name], v)  # bit-exact


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` for real (not --dry-run): the self-launcher, the rendezvous, two ranks running the kernels of the hist
    sweep and of the mse sweep (exact-tail form through ops.OctavPipeline) on the box's one GPU, the merge collectives inside
    the timed region (gloo, because RCCL refuses two ranks on one device), max-over-ranks timing, rank 0's record line: the code
    path of the driver's N = 2 / 4 / 8 runs above the backend string."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DPL_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--mse-steps", "1",
                        "--e2e-images", "0", "--vit-images", "0", "--real-images", "0", "--fq-reps", "0", "--mse-jitter", "", "--pool", "5",
                        "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    line = json.loads(lines[-1])
    assert len(lines[-1]) < 2048 and "details" in json.loads(lines[-2])
    assert line["n_gpus"] == 2 and line["config"]["world_size_seen_by_backend"] == 2 and line["config"]["backend"] == "gloo"
    assert line["config"]["hist_checksum_ok"] is True and len(line["config"]["per_rank_images_per_s"]) == 2
    assert line["config"]["collectives_ms_per_sweep"] > 0
    assert line["roofline"]["mse"]["ok"] is True and line["value"] > 0


def test_bench_one_rank_over_rccl():
    """RCCL for real, as far as one GPU allows: `bench.py` as ONE rank of a torch.distributed.run-style launch with backend
    'nccl' — communicator set-up on the device, the barriers, the merge collectives of the hist sweep (min / max as encoded
    uint32 -> int64, int64 histograms) and of the mse sweep (all_gather of the [n, T, 3] rows), the max-over-ranks timing on a
    device tensor: every dtype / reduce-op pair the N = 2 / 4 / 8 runs hand to RCCL goes through its API here."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("DPL_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--mse-steps", "1",
                        "--e2e-images", "0", "--vit-images", "0", "--real-images", "0", "--fq-reps", "0", "--mse-jitter", "", "--pool", "5",
                        "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])       # (RCCL's version banner must not end up behind the record line)
    assert line["config"]["backend"] == "nccl (RCCL)" and line["config"]["world_size_seen_by_backend"] == 1
    assert line["config"]["hist_checksum_ok"] is True and line["roofline"]["mse"]["ok"] is True and line["value"] > 0


def _rccl_worker(rank, port):
    """Every collective the path hands to the backend, through RCCL itself on a one-rank group (the merge functions are called
    with world_size = 2 so that they do not short-circuit: over one rank every reduction returns its input)."""
    import torch.distributed as dist
    from dipoorlet_amd import dist_helper
    from dipoorlet_amd.weight_transform.bias_correction import _channel_mean_diff
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    # running min / max with a NaN tensor: fp32 MIN / MAX + int32 MAX (dist_helper.merge_ranges)
    gmin = torch.tensor([-1.5, float("nan"), 2.0, -0.0], device=dev)
    gmax = torch.tensor([3.0, 1.0, 2.0, float("inf")], device=dev)
    lo, hi = dist_helper.merge_ranges(gmin.clone(), gmax.clone(), 2)
    assert torch.equal(torch.isnan(lo), torch.tensor([False, True, False, False], device=dev)) and torch.isnan(hi[1])
    assert lo[0] == -1.5 and hi[0] == 3.0 and hi[3] == float("inf")
    # histograms: int64 SUM, counts above 2^31
    hist = torch.randint(0, 1 << 40, (123, 2048), dtype=torch.int64, device=dev, generator=g)
    assert torch.equal(dist_helper.merge_hist(hist.clone(), 2), hist)
    # OCTAV rows: the fused all-gather and the list form gather_rows falls back to
    rows = torch.randn(5, 123, 3, device=dev, generator=g)
    out = torch.empty_like(rows)
    dist.all_gather_into_tensor(out, rows)
    parts = [torch.empty_like(rows)]
    dist.all_gather(parts, rows)
    assert torch.equal(out, rows) and torch.equal(parts[0], rows) and torch.equal(dist_helper.gather_rows(rows, 1), rows)
    # --bc: one fp64 SUM of [C + 1] per node (Conv and Gemm outputs)
    for shape, is_conv in (((4, 16, 7, 7), True), ((4, 10), False)):
        fp = [torch.randn(shape, device=dev, generator=g) for _ in range(2)]
        q = [x + 0.01 * torch.randn(shape, device=dev, generator=g) for x in fp]
        one = _channel_mean_diff(fp, q, is_conv)
        merged = _channel_mean_diff(fp, q, is_conv, world_size=2, n_ch=shape[1])
        assert torch.equal(one, merged)
    # bench.py's timing exchange: fp64 MAX and a gather of per-rank rates on device tensors, between two barriers
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rates = [torch.zeros(1, dtype=torch.float64, device=dev)]
    dist.all_gather(rates, t)
    dist.barrier()
    torch.cuda.synchronize()
    assert t.item() == 1.25 and rates[0].item() == 1.25
    dist.destroy_process_group()


def test_every_collective_of_the_path_through_rccl():
    """No node with more than one GPU was ever available to the rounds: RCCL has never run at N > 1.  What one GPU can prove is
    that RCCL accepts every (dtype, reduce-op) pair and gather form the path uses — fp32 MIN / MAX, int32 MAX, int64 SUM, fp64
    SUM / MAX, all_gather_into_tensor, list all_gather, barrier — on device tensors, in this image, with the IPC mode the pool
    needs (_rccl_worker: a one-rank 'nccl' group)."""
    port = 29100 + os.getpid() % 200
    mp.spawn(_rccl_worker, args=(port,), nprocs=1, join=True)


def _bc_worker(rank, world, port, model, calib, out_dir, n, clips_from=None, batch=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DPL_DIST_BACKEND="gloo", DPL_DETERMINISTIC="1")
    # (two schedules, one answer: the library's deterministic algorithms — tests/conftest.py `two_forwards`; the variable is read
    # when dipoorlet_amd.executor is imported, in this fresh process)
    if clips_from is None:          # the whole CLI
        from dipoorlet_amd.__main__ import main
        rc = main(["-M", model, "-I", calib, "-N", str(n), "-A", "minmax", "-D", "trt", "-O", out_dir, "--calib_batch", str(batch),
                   "--bc", "--skip_profiling"])
        assert rc == 0
    else:                           # bias_correction alone, on the clip ranges another run wrote
        from dipoorlet_amd import dist_helper
        from dipoorlet_amd.graph import ONNXGraph
        from dipoorlet_amd.utils import load_clip_val
        from dipoorlet_amd.weight_transform.bias_correction import bias_correction
        dist_helper.init_default()
        os.makedirs(out_dir, exist_ok=True)
        args = types.SimpleNamespace(input_dir=calib, data_num=n, rank=rank, local_rank=0, world_size=world, deploy="trt",
                                     calib_batch=batch, output_dir=clips_from, skip_layers=[], merge="allreduce")
        a, w = load_clip_val(args)
        args.output_dir = out_dir
        bias_correction(ONNXGraph.load(model), a, w, args)
        torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _bc_setup(tmp_path, n):
    from dipoorlet_amd import models
    g = models.resnet18(seed=11, image=64)
    g.output_dir = str(tmp_path)
    model = g.save_onnx_model("model")
    os.makedirs(tmp_path / "calib" / "input")
    rng = np.random.default_rng(5)
    for i in range(n):
        rng.standard_normal(3 * 64 * 64).astype(np.float32).tofile(tmp_path / "calib" / "input" / f"{i}.bin")
    return g, model


def _bc_compare(g, tmp_path):
    """Both runs executed the same kernels on the same values image by image (deterministic library algorithms, same batch
    shapes): what differs is the ORDER in which the per-channel fp64 sums of (fp - q) were added up — over chunks on one rank,
    over chunks then over ranks (one all-reduce) on two — i.e. the last bit of a bias at most."""
    from dipoorlet_amd.graph import ONNXGraph
    g1 = ONNXGraph.load(str(tmp_path / "w1" / "update_bias_model.onnx"))
    g2 = ONNXGraph.load(str(tmp_path / "w2" / "update_bias_model.onnx"))
    checked, moved = 0, 0.0
    for node in g.graph.node:
        if node.op_type in ("Conv", "Gemm"):
            b0 = g.get_initializer(node.input[2])
            b1 = g1.get_initializer(node.input[2])
            b2 = g2.get_initializer(node.input[2])
            moved = max(moved, float(np.abs(b1 - b0).max()))
            ulps = np.abs(b1.astype(np.float64) - b2.astype(np.float64)) / np.spacing(np.maximum(np.abs(b1), np.float32(1e-30))).astype(np.float64)
            assert ulps.max() <= 1.0, (node.name, float(np.abs(b1 - b2).max()), float(ulps.max()))
            checked += 1
    assert checked >= 10 and moved > 1e-4       # (the correction did something)


@pytest.mark.two_forwards
def test_bias_correction_sharded_over_two_ranks_equals_one_rank(tmp_path):
    """`--bc` with the images sharded over two ranks (each walks its shard node-major, the per-channel fp64 sums are
    all-reduced per Conv / Gemm node; weight_transform/bias_correction.py) writes the biases the one-rank run writes
    (the reference's schedule: rank 0 over all images, weight_trans_base.py:21-29).  The whole CLI, N = 8: both runs execute
    the same batches of 4."""
    g, model = _bc_setup(tmp_path, 8)
    port = 29300 + os.getpid() % 200
    mp.spawn(_bc_worker, args=(1, port, model, str(tmp_path / "calib"), str(tmp_path / "w1"), 8), nprocs=1, join=True)
    mp.spawn(_bc_worker, args=(2, port + 1, model, str(tmp_path / "calib"), str(tmp_path / "w2"), 8), nprocs=2, join=True)
    _bc_compare(g, tmp_path)
    with open(tmp_path / "w2" / "weight_clip_val.json") as f1, open(tmp_path / "w1" / "weight_clip_val.json") as f2:
        assert set(json.load(f1)) == set(json.load(f2))


@pytest.mark.two_forwards
def test_bias_correction_balanced_split_covers_every_image(tmp_path):
    """N = 7 over two ranks: bias_correction's own split is balanced (3 + 4 images) and covers every image, where the
    calibration sweeps' floor split (forward_net.py:207-209) would drop the seventh — the reference corrects with all N
    (forward_net.py:50-52).  bias_correction alone, both runs on the clip ranges the one-rank CLI run wrote, one image per
    forward (--calib_batch 1) on both sides, so that 7 = 3 + 4 images go through the same kernels one by one: a rank that
    dropped or doubled an image would move every bias by a seventh of the correction; the runs agree to the last bit."""
    g, model = _bc_setup(tmp_path, 7)
    port = 29500 + os.getpid() % 200
    mp.spawn(_bc_worker, args=(1, port, model, str(tmp_path / "calib"), str(tmp_path / "w1"), 7, None, 1), nprocs=1, join=True)
    mp.spawn(_bc_worker, args=(2, port + 1, model, str(tmp_path / "calib"), str(tmp_path / "w2"), 7, str(tmp_path / "w1"), 1), nprocs=2, join=True)
    _bc_compare(g, tmp_path)


@pytest.mark.two_forwards
def test_bias_correction_with_an_empty_shard(tmp_path):
    """More ranks than images (N = 1 over two ranks: bc_shard gives rank 0 nothing): the rank without images still joins every
    per-node all-reduce and both ranks write the one-rank result."""
    g, model = _bc_setup(tmp_path, 1)
    port = 29600 + os.getpid() % 200
    mp.spawn(_bc_worker, args=(1, port, model, str(tmp_path / "calib"), str(tmp_path / "w1"), 1, None, 1), nprocs=1, join=True)
    mp.spawn(_bc_worker, args=(2, port + 1, model, str(tmp_path / "calib"), str(tmp_path / "w2"), 1, str(tmp_path / "w1"), 1), nprocs=2, join=True)
    _bc_compare(g, tmp_path)


def test_bench_four_ranks_on_one_gpu():
    """`bench.py --gpus 4` for real on the box's one GPU (gloo; --pool 3 keeps four ranks' resident batches small): rank-indexed
    seeds, the ranges / histogram all-reduces and the gather of the OCTAV rows at world size 4, four per-rank rates, the
    collectives' milliseconds of both sweeps in rank 0's line — what the driver's N = 4 run does above the backend string.
    With the DEFAULT side-object flags, as the driver passes none: over more than one rank the run measures the sharded path
    (hist + mse with their collectives) and nothing else — no ViT / 448 x 448 / jitter / one-stream / fake-quant objects, no CLI
    children, no CPU baseline on any rank (they characterise one GPU and are the N = 1 run's)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DPL_DIST_BACKEND"] = "gloo"
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--mse-steps", "1",
                        "--pool", "3"], env=env, capture_output=True, text=True, timeout=1200)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    line = json.loads(lines[-1])
    details = json.loads(lines[-2])["details"]
    assert line["n_gpus"] == 4 and line["config"]["world_size_seen_by_backend"] == 4 and len(line["config"]["per_rank_images_per_s"]) == 4
    assert line["config"]["hist_checksum_ok"] is True and line["config"]["collectives_ms_per_sweep"] > 0
    assert line["roofline"]["mse"]["ok"] is True and details["mse"]["collectives_ms_per_sweep"] > 0
    assert details["hist"]["hist_checksum"] == 4 * 1024 * 26598376
    for k in ("mse_lanes1", "mse_jitter", "fake_quant", "vit_mse", "mse_448", "mse_feature_maps", "e2e"):
        assert details[k] is None, k
    assert line["cpu_baseline"] is None and len(lines[-1]) < 2000
    assert wall < 240, wall      # (the side objects alone take minutes per rank)


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """Over RCCL a rank needs a GPU of its own: bench.py says so instead of hanging in the communicator's set-up."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DPL_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--mse-steps", "0",
                        "--e2e-images", "0", "--fq-reps", "0", "--cpu-seconds", "0", "--pool", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "need" in (r.stderr + r.stdout)
