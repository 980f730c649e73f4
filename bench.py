#!/usr/bin/env python3
"""Benchmark of the calibration hot path on MI355X.

A "step" = one pass of the hot path over one batch of B synthetic calibration images' activations
(ResNet-50 tensor multiset: 123 tensors, 26,598,376 fp32 elements = 106.39 MB per image), already
resident in HBM:
    -A hist  (default, BASELINE configs[1]): range pass (k_minmax) + histogram pass (k_abs_hist) over the
             batch = both reads the algorithm inherently needs (212.79 MB / image algorithmic);
    -A minmax: range pass only (106.39 MB / image);   -A mse: OCTAV (106.39 MB / image credited).
Default: B = 32, 32 steps = one whole N = 1024 calibration set per GPU.

Prints ONE JSON line (rank 0).  `value` is whole-job images/s; `roofline` is for the dominant kernel
(k_abs_hist for hist), its duration measured with HIP events on the launch stream inside the timed
region; `cpu_baseline` times the CPU oracle (numpy port of the reference arithmetic) on a bounded
sample of the same activations on the host cores of this box.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): images are sharded across ranks with
no data-path collective (weak scaling: every rank runs K steps of B images); the one real exchange of
the algorithm — all-reduce MIN/MAX of the ranges and SUM of the histograms over RCCL — runs once,
inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--batch", type=int, default=32, help="calibration images per step and per GPU")
    p.add_argument("--algo", choices=["hist", "minmax", "mse"], default="hist")
    p.add_argument("--bins", type=int, default=2048)
    p.add_argument("--pool", type=int, default=2, help="distinct resident batches cycled through")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables")
    p.add_argument("--chunk", type=int, default=0, help="work-item elements (0 = auto)")
    return p.parse_args()


def cpu_baseline(algo, bins, tensors, budget_s):
    """Time the CPU oracle on the host cores of this box over a bounded sample of the same activations.
    Checker code is being MEASURED here as the CPU side of the comparison, never shipped.

    Main figure: the plain-C restatement (oracle/c_oracle.c, bit-compatible with the reference's numpy
    arithmetic), OpenMP-parallel over the (image, tensor) arrays on all cores.  Also reported: the numpy
    restatement on one thread — what the reference's own Python does per image."""
    import warnings

    from oracle import c_oracle as CO
    from oracle import np_oracle as O
    B = tensors[0].shape[0]
    host = [t.cpu().numpy() for t in tensors]                       # [B, e] each
    arrays = [h[b] for b in range(B) for h in host]                 # B * T independent arrays
    threads = os.cpu_count() or 1
    CO.batch(arrays[:len(host)], algo, bins, threads)               # warm up (thread pool, page faults)
    done, t_used = 0, 0.0
    while t_used < budget_s * 0.75:
        t0 = time.perf_counter()
        used = CO.batch(arrays, algo, bins, threads)[0]
        t_used += time.perf_counter() - t0
        done += B
    # numpy, one thread, a few images
    n_np, t_np = 0, 0.0
    while t_np < budget_s * 0.25:
        xs = [h[n_np % B] for h in host]
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mm = [O.minmax(x) for x in xs]
            if algo == "hist":
                for x, (lo, hi) in zip(xs, mm):
                    O.abs_hist(x, bins, O.hist_dmax(lo, hi))
            elif algo == "mse":
                for x in xs:
                    O.octav_scale(x, 1)
        t_np += time.perf_counter() - t0
        n_np += 1
    c_rate, np_rate = done / t_used, n_np / t_np
    out = {"unit": "images/s", "kind": "port", "c_openmp_images_per_s": c_rate, "c_openmp_threads": int(used),
           "numpy_single_thread_images_per_s": np_rate}
    if c_rate >= np_rate:
        out.update(value=c_rate, cores=int(used),
                   sample=f"{done} images ({done // B} passes over {B} images' ResNet-50-shaped activations), -A {algo}, "
                          f"C oracle with OpenMP over (image, tensor) arrays, {t_used:.1f} s; host has {os.cpu_count()} cores")
    else:  # (a streaming min/max is memory-bound: one numpy thread beats the OpenMP fan-out on this host)
        out.update(value=np_rate, cores=1,
                   sample=f"{n_np} images of the same ResNet-50-shaped activations, -A {algo}, numpy oracle on one "
                          f"thread, {t_np:.1f} s; host has {os.cpu_count()} cores")
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group("nccl")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", torch.cuda.current_device())

    from dipoorlet_amd import _hip, ops
    from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
    st, devname, cus, mem = _hip.device_info()
    spec = resnet50_tensors()
    elems = [e for _, e, _ in spec]
    T, E = len(elems), sum(elems)
    B = a.batch
    pool = [synth_activations(spec, B, dev, seed=1234 + 1000 * rank + j) for j in range(max(1, a.pool))]
    plan = ops.TensorSetPlan(elems, B, dev, chunk_elems=a.chunk or None)
    acc = ops.CalibAccumulators(T, dev, a.bins)
    acc_rng = ops.CalibAccumulators(T, dev, a.bins)  # timed range pass writes here; `acc` keeps the global ranges

    # global ranges over everything this rank will see (pass 1 of the real algorithm), merged over ranks
    for tset in pool:
        acc.minmax_accumulate(plan, tset)
    gmin, gmax = acc.finalize_minmax()
    if use_dist:
        dist.all_reduce(gmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
        acc.set_minmax(gmin.clone(), gmax.clone())
    acc.hist_prepare()
    states = None

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def step(i, timed):
        ta = pool[i % len(pool)]
        tb = pool[(i + 1) % len(pool)]
        if a.algo == "minmax":
            if timed:
                ev[i][0].record()
            acc_rng.minmax_accumulate(plan, ta)
            if timed:
                ev[i][1].record()
        elif a.algo == "hist":
            acc_rng.minmax_accumulate(plan, ta)  # pass 1 work for this batch
            if timed:
                ev[i][0].record()
            acc.abs_hist_accumulate(plan, tb)  # pass 2 work (a different resident batch: no cache reuse)
            if timed:
                ev[i][1].record()
        else:
            if timed:
                ev[i][0].record()
            rows = ops.octav_batch(plan, ta, False, states)
            if timed:
                ev[i][1].record()
            return rows
        return None

    if a.algo == "mse":
        import ctypes
        states = torch.empty((plan.n_pairs + 1) * ctypes.sizeof(_hip.OctavState), dtype=torch.uint8, device=dev)
    for i in range(a.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    octav_rows = None
    for i in range(a.steps):
        octav_rows = step(i, True)
    if use_dist:  # the algorithm's one exchange step per run, inside the timed region
        if a.algo == "mse":
            from dipoorlet_amd.dist_helper import gather_rows
            gather_rows(octav_rows, world)       # per-image (s, min, max) rows of the last batch as a stand-in
        else:
            mn, mx = acc_rng.finalize_minmax()
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            if a.algo == "hist":
                dist.all_reduce(acc.hist, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kern_ms = sum(s.elapsed_time(e) for s, e in ev) / max(1, a.steps)
    bytes_per_img = {"hist": 8 * E, "minmax": 4 * E, "mse": 4 * E}[a.algo]
    kernel_bytes = 4 * E * B  # the dominant kernel reads the batch once
    achieved = kernel_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic = None
    tj = os.environ.get("DPL_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "traffic_latest.json"))
    if os.path.exists(tj):
        try:
            with open(tj) as f:
                tr = json.load(f)
            if tr.get("algo") == a.algo and tr.get("batch") == B:
                traffic = tr.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    images = a.steps * B * world
    out = {
        # BASELINE.json's metric; images/s is `value`, the achieved HBM GB/s of the dominant kernel is `roofline.achieved`
        "metric": "calibration images/sec (whole node) + achieved HBM GB/s, ResNet-50 ONNX N=%d, -A %s" % (a.steps * B, a.algo),
        "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ResNet-50 ONNX activation set (T={T} tensors, {E} fp32 elems/img), -A {a.algo}"
                               f" --bins {a.bins}, N={a.steps * B} images per GPU in batches of {B}",
                   "batch": B, "bins": a.bins, "algo": a.algo, "workgroups": plan.work("minmax" if a.algo == "minmax" else ("octav" if a.algo == "mse" else "hist"),
                                                     a.algo == "mse").n_blocks,
                   "chunk_elems": plan.chunk, "device": devname},
        "algorithmic_GBps_job": bytes_per_img * images / dt / 1e9,
        "roofline": {"bound": "hbm", "kernel": {"hist": "k_abs_hist", "minmax": "k_minmax",
                                               "mse": "octav_batch (k_octav_loghist + bracket + gather + exact)"}[a.algo],
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "bytes_per_launch": kernel_bytes, "avg_kernel_ms": kern_ms},
    }
    if rank == 0:
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(a.algo, a.bins, pool[0], a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
