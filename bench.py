#!/usr/bin/env python3
"""Benchmark of the calibration hot path on MI355X.

A "step" = ONE WHOLE CALIBRATION SWEEP of BASELINE.json's headline configuration over activations already resident in
HBM: ResNet-50 (123 tensors, 26,598,376 fp32 elements = 106.39 MB per image), `-A hist --bins 2048`, N = 1024 images per
GPU in 32 batches of 32:
    pass 1  running min / max of every tensor over the 32 batches      (k_minmax:   one read of the set)
            [N > 1 GPUs: all-reduce MIN / MAX of the ranges over RCCL]
    pass 2  |x| histograms against the global ranges                   (k_abs_hist: the second, inherent read)
            [N > 1 GPUs: all-reduce SUM of the [123, 2048] int64 histograms over RCCL]
    clip    percentile threshold per tensor                            (k_hist_percentile)
so `--steps K` times K sweeps (~35 ms each) and `metric` / `config.workload` say N = 1024 whatever K is.

The same JSON line carries, as `"mse"`, BASELINE configs[2]: the `-A mse` (OCTAV) sweep, N = 4096 images per GPU in 128
batches of 32 through ops.octav_batch (one-read form) + the per-tensor clip — timed the same way over `--mse-steps` sweeps.

`value` is whole-job images/s of the hist sweep; `roofline` is the hist kernel's (duration by HIP events on the launch
stream inside the timed region) and carries the other objects' headline scalars (`roofline.mse` = configs[2], `.mse_jitter`,
`.mse_feature_maps`, `.mse_vit`, `.fake_quant` = per mode [one launch per tensor over the set, ... over the tensors >= 50 MB,
the set in one launch], `.e2e` = per algorithm [calibration images/s of a fresh CLI process, images/s of the network forward]);
`cpu_baseline` times the CPU oracle (a port of the reference's
arithmetic) on a bounded sample of the same activations on the host cores of this box, rank 0, N = 1 only.
TWO lines are printed: `{"details": {...}}` (every object in full: workload strings, prediction statistics, the e2e split)
and then the record's line (< 2 KB).

Multi-GPU: launched by torch.distributed.run (one rank per GPU) — or, when started plainly with --gpus N > 1, this script
starts the N ranks itself as child processes (before anything touches the GPU) and relays rank 0's line.  Images are
sharded with no data-path collective (weak scaling: every rank sweeps its own N images); the algorithm's real exchanges
(above) run inside the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_HIST, N_MSE, BATCH = 1024, 4096, 32   # BASELINE.json configs[1], configs[2]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20, help="timed hist sweeps (N = 1024 images each)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--mse-steps", type=int, default=4, help="timed mse sweeps (N = 4096 images each); 0 skips the mse object")
    p.add_argument("--bins", type=int, default=2048)
    p.add_argument("--pool", type=int, default=17, help="distinct resident batches of 32 images cycled through (3.4 GB each); "
                   "17 = more than the exact-tail OCTAV form's threshold history remembers (2 epochs of 8 batches)")
    p.add_argument("--mse-jitter", default="0.03,0.1", help="extra one-sweep mse objects with per-image contrast jitter (comma list; '' = none)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget; 0 disables")
    p.add_argument("--e2e-images", type=int, default=1024, help="N of the end-to-end CLI object (real .onnx + .bin files, fresh process; 0 skips it)")
    p.add_argument("--real-images", type=int, default=2048, help="N of the mse objects over executor-produced ResNet-50 activations (0 skips them)")
    p.add_argument("--vit-images", type=int, default=256, help="N of the ViT-B/16 mse object (0 skips it)")
    p.add_argument("--big-images", type=int, default=256, help="N of the mse object over ResNet-50's shapes at 448 x 448 input: tensors above "
                   "one OCTAV slice, batches of 8 (0 skips it)")
    p.add_argument("--fq-reps", type=int, default=3, help="timed passes of the fake-quant object (0 skips it)")
    p.add_argument("--dry-run", action="store_true",
                   help="launcher / rendezvous / collective plumbing only, on CPU tensors (no kernels, no GPU): what the "
                        "world_size-2 gloo test on the build box runs")
    p.add_argument("--algo", choices=["hist", "mse"], default="hist",
                   help="which sweep is the headline `value` (the other one is still reported: mse as the `mse` object)")
    return p.parse_args()


def self_launch(a):
    """Started without a launcher but asked for N > 1 ranks: start them (fresh child processes, nothing here has touched
    the GPU), rendezvous on 127.0.0.1, relay rank 0's stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(max(abs(c) for c in codes))


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
    n_np, t_np = 0, 0.0
    while t_np < budget_s * 0.25:                                   # numpy, one thread, a few images
        xs = [h[n_np % B] for h in host]
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mm = [O.minmax(x) for x in xs]
            if algo == "hist":
                for x, (lo, hi) in zip(xs, mm):
                    O.abs_hist(x, bins, O.hist_dmax(lo, hi))
            else:
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
                          f"C oracle with OpenMP over (image, tensor) arrays, {t_used:.1f} s; host has {os.cpu_count()} cores "
                          f"(memory- / NUMA-bound: {threads} threads give {c_rate / max(np_rate, 1e-9):.1f} x one numpy thread)")
    else:
        out.update(value=np_rate, cores=1,
                   sample=f"{n_np} images of the same ResNet-50-shaped activations, -A {algo}, numpy oracle on one "
                          f"thread, {t_np:.1f} s; host has {os.cpu_count()} cores")
    return out


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)   # does not return

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run or by self_launch (also with one rank: same code path)
    backend = os.environ.get("DPL_DIST_BACKEND", "gloo" if a.dry_run else "nccl")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not a.dry_run:
            if backend == "nccl" and torch.cuda.device_count() < world:
                sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(RCCL refuses two ranks on one device; DPL_DIST_BACKEND=gloo runs them on one GPU for plumbing tests)")
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend)
    if a.dry_run:
        # the exchanges of one hist sweep on stand-in CPU tensors: same launcher, same rendezvous, same collectives, same line
        T, bins = 123, a.bins
        gmin = torch.full((T,), float(-rank - 1)), torch.full((T,), float(rank + 1))
        hist = torch.full((T, bins), rank + 1, dtype=torch.int64)
        if use_dist:
            dist.barrier()
            dist.all_reduce(gmin[0], op=dist.ReduceOp.MIN)
            dist.all_reduce(gmin[1], op=dist.ReduceOp.MAX)
            dist.all_reduce(hist, op=dist.ReduceOp.SUM)
            dist.barrier()
        ok = torch.tensor([1.0 if int(hist[0, 0]) == world * (world + 1) // 2 else 0.0])
        per_rank_ok = [torch.zeros(1) for _ in range(world)]
        if use_dist:
            dist.all_gather(per_rank_ok, ok)
        else:
            per_rank_ok = [ok]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "backend": backend if use_dist else None,
                              "world_size_seen_by_backend": dist.get_world_size() if use_dist else 1,
                              "per_rank_ok": [bool(x.item()) for x in per_rank_ok],
                              "range": [float(gmin[0][0]), float(gmin[1][0])], "hist_checksum": int(hist.sum().item()),
                              "hist_checksum_expected": T * bins * world * (world + 1) // 2}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", torch.cuda.current_device())

    from dipoorlet_amd import _hip, ops
    from dipoorlet_amd.dist_helper import gather_rows
    from dipoorlet_amd.synthetic import resnet50_tensors, synth_activations
    _, devname, _, _ = _hip.device_info()
    spec = resnet50_tensors()
    elems = [e for _, e, _ in spec]
    T, E = len(elems), sum(elems)
    B = BATCH
    # DPL_BENCH_JITTER (a tuning aid, not the headline workload): per-image contrast jitter of the synthetic activations
    jitter = float(os.environ.get("DPL_BENCH_JITTER", "0"))
    plan = ops.TensorSetPlan(elems, B, dev)
    # (resident sets are validated and pinned once, TensorSetPlan.bind: a launch over them costs no per-tensor checks)
    pool = [plan.bind(synth_activations(spec, B, dev, seed=1234 + 1000 * rank + j, image_jitter=jitter)) for j in range(max(1, a.pool))]
    n_pool = len(pool)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    # ------------------------------------------------------------------ -A hist --bins 2048, N = 1024 per GPU
    acc = ops.CalibAccumulators(T, dev, a.bins)
    n_hist_batches = N_HIST // B
    hist_ev = []

    coll_ev = []

    def hist_sweep(timed):
        acc.reset_minmax()
        for b in range(n_hist_batches):                                   # pass 1
            acc.minmax_accumulate(plan, pool[b % len(pool)])
        gmin, gmax = acc.finalize_minmax()
        if use_dist:                                                      # the algorithm's exchange after pass 1
            if timed:     # the collectives' own time (events on the stream the backend enqueues on: what the merge costs a sweep)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
            dist.all_reduce(gmin, op=dist.ReduceOp.MIN)
            dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
            if timed:
                c1.record()
                coll_ev.append((c0, c1))
            acc.set_minmax(gmin.clone(), gmax.clone())
        acc.hist_prepare()
        for b in range(n_hist_batches):                                   # pass 2
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            acc.abs_hist_accumulate(plan, pool[b % len(pool)])
            if timed:
                e1.record()
                hist_ev.append((e0, e1))
        if use_dist:                                                      # ... and after pass 2
            if timed:
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
            dist.all_reduce(acc.hist, op=dist.ReduceOp.SUM)
            if timed:
                c1.record()
                coll_ev.append((c0, c1))
        return acc.hist_percentile(0.99999)

    for _ in range(a.warmup):
        hist_sweep(False)
    fence()
    t0 = time.perf_counter()
    clip = None
    for _ in range(a.steps):
        clip = hist_sweep(True)
    fence()
    dt_local = time.perf_counter() - t0
    dt_hist = max_over_ranks(dt_local)
    if use_dist:   # every rank's own rate (images/s over its own clock), gathered for the line
        rates = [torch.zeros(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu") for _ in range(world)]
        dist.all_gather(rates, torch.tensor([N_HIST * a.steps / dt_local], dtype=torch.float64, device=rates[0].device))
        per_rank_rates = [float(r.item()) for r in rates]
    else:
        per_rank_rates = [N_HIST * a.steps / dt_local]
    hist_kern_ms = sum(s.elapsed_time(e) for s, e in hist_ev) / max(1, len(hist_ev))
    # RCCL's three all-reduces of a sweep (MIN + MAX of the ranges, SUM of the [T, bins] int64 histograms), milliseconds per sweep
    coll_ms = round(sum(s.elapsed_time(e) for s, e in coll_ev) / max(1, a.steps), 4) if coll_ev else None
    hist_checksum = int(acc.hist.sum().item())        # = elements x images x ranks when every rank's counts arrived
    clip_checksum = float(clip.double().abs().sum().item())

    # ------------------------------------------------------------------ -A mse (OCTAV), N = 4096 per GPU
    # A timed sweep is ONE COLD calibration run (forward_net.py:297-340: one independent pass over the shard): the plan forgets
    # what earlier sweeps learned (octav_reset) inside the timed region, so the first batches run without a prediction as they
    # do in a fresh process.  The pool holds more distinct batches than the prediction remembers (2 epochs of
    # ops._ONEREAD_EPOCH batches), so no batch is ever predicted from itself.
    mse, mse_jitter, vit_mse, mse_real, mse_big, mse_lanes1 = None, {}, None, {}, None, None
    if a.mse_steps > 0:
        import ctypes
        form = ops._default_form()
        pipeline = os.environ.get("DPL_OCTAV_PIPELINE", "1") != "0" and form == "tail"
        pipe = ops.OctavPipeline(False, dev) if pipeline else None
        pipe1 = ops.OctavPipeline(False, dev, lanes=1) if pipeline else None    # the schedule forward_net_octav runs (below)
        min_pool = 2 * ops._ONEREAD_EPOCH + 1
        # DPL_BENCH_FAIL_EVERY=n (a tuning aid, not the headline workload): the C ABI's test hook makes every n-th pair's walk
        # report a missed prediction, to price the device-side rescue of such pairs
        inject = int(os.environ.get("DPL_BENCH_FAIL_EVERY", "0"))
        if inject > 0:
            _hip.lib().dpl_test_hook_exact_fail_every(inject)

        def run_mse(mpool, steps, jit, plan=plan, n_images=N_MSE, net="ResNet-50", pipe=pipe):
            mse_ev, mse_coll = [], []
            mpool = [plan.bind(x) for x in mpool]
            B, T, E = plan.batch, plan.T, sum(plan.elems)        # (the ViT object runs its own plan through the same code)
            n_mse_batches = n_images // B
            rows = torch.empty(n_images, T, 3, dtype=torch.float32, device=dev)
            states = torch.empty((plan.n_pairs + 1) * ctypes.sizeof(_hip.OctavState), dtype=torch.uint8, device=dev)

            def mse_sweep(timed):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                plan.octav_reset()         # a cold run: nothing learned from the previous sweep
                # pipe: two lane streams — the streaming kernel of batch i + 1 starts while batch i drains (the statistics
                # kernels alone on the chip: this object); pipe1: lanes = 1, one stream — what forward_net.forward_net_octav runs
                # between two network forwards (`mse_lanes1`); either way the rescue of batch i runs beside batch i + 1
                if pipe is not None:
                    outs = [pipe.submit(plan, mpool[b % len(mpool)]) for b in range(n_mse_batches)]
                    pipe.sync()
                    torch.cat(outs, out=rows)
                else:
                    for b in range(n_mse_batches):
                        rows[b * B:(b + 1) * B] = ops.octav_batch(plan, mpool[b % len(mpool)], False, states, form=form)
                if timed:
                    e1.record()
                    mse_ev.append((e0, e1))
                if use_dist and timed:
                    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    c0.record()
                allr = gather_rows(rows, world) if use_dist else rows          # the algorithm's exchange: per-image rows
                if use_dist and timed:
                    c1.record()
                    mse_coll.append((c0, c1))
                s_mean = allr[:, :, 0].mean(0)                                 # basic_algorithm.py:57-69 on the device
                lo = torch.maximum(allr[:, :, 1].amin(0), -s_mean)
                hi = torch.minimum(allr[:, :, 2].amax(0), s_mean)
                return torch.stack([lo, hi], 1)

            mse_sweep(False)   # warm-up of the allocator / code objects only: every timed sweep starts cold
            if pipe is not None:
                pipe.reset_stats()
            fence()
            t0 = time.perf_counter()
            mclip = None
            for _ in range(steps):
                mclip = mse_sweep(True)
            fence()
            dt_mse = max_over_ranks(time.perf_counter() - t0)
            mse_ms = sum(s.elapsed_time(e) for s, e in mse_ev) / max(1, len(mse_ev)) / n_mse_batches   # per batch, in the sweep
            mse_bytes = 4 * E * B          # credited: ONE read of the batch, whatever the form actually reads
            mse_ach = mse_bytes / (mse_ms * 1e-3) / 1e9 if mse_ms > 0 else 0.0
            # parity spot check against the CPU oracle (checker, after the timed region): 8 (image, tensor) pairs of the last batch
            last = mpool[(n_mse_batches - 1) % len(mpool)]
            ok, worst = True, 0.0
            if rank == 0:
                from oracle import np_oracle as O
                import numpy as np
                rs = np.random.RandomState(7)
                for t in [0, 1, 2] + list(rs.choice(T, 5, replace=False)):
                    img = int(rs.randint(B))
                    want = float(O.octav_scale(last[t][img].cpu().numpy(), 1))
                    got = float(rows[(n_mse_batches - 1) * B + img, t, 0].item())
                    err = abs(got - want) / max(abs(want), 1.0)
                    worst = max(worst, err)
                    ok = ok and err <= 1e-5
            obj = {"metric": f"calibration images/sec, {net} ONNX N={n_images}, -A mse", "value": n_images * world * steps / dt_mse,
                   "unit": "images/s", "steps": steps, "ms_per_step": dt_mse / steps * 1e3,
                   "workload": f"{net} activation set (T={T} tensors, {E} fp32 elems/img), -A mse (OCTAV per image and tensor), N={n_images} images per GPU in "
                               f"batches of {B}, form '{form}', every sweep a cold run, {len(mpool)} distinct resident batches"
                               + (f", per-image contrast jitter +-{jit:g}" if jit else ""),
                   "roofline": {"bound": "hbm", "kernel": f"OCTAV batch, form '{form}'" + (" (streaming kernel of batch i+1 beside the rescue of batch i)"
                                                                                   if pipe is not None else ""),
                                "achieved": mse_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": mse_ach / HBM_PEAK_GBPS,
                                "traffic": None, "bytes_per_launch": mse_bytes, "avg_batch_ms": mse_ms},
                   "clip_checksum": float(mclip.double().abs().sum().item()),
                   "sample_ok": bool(ok), "sample_worst_rel_err": worst,
                   # the all_gather of the [n, T, 3] rows (RCCL), milliseconds per sweep
                   "collectives_ms_per_sweep": round(sum(s.elapsed_time(e) for s, e in mse_coll) / max(1, len(mse_coll)), 4) if mse_coll else None}
            if pipe is not None and pipe.scratch_bytes(plan) is not None:
                # every device byte the form holds besides the activations: tables, threshold history, state / rescue blocks,
                # lists (and the compaction route's lists, had a batch asked for them)
                obj["scratch_bytes"] = int(pipe.scratch_bytes(plan))
                obj["scratch_over_batch_activations"] = round(pipe.scratch_bytes(plan) / (4.0 * E * B), 4)
            if inject > 0:
                obj["injected_miss_every"] = inject
            if pipe is not None:   # prediction misses (pairs finished by a re-read of the pair), timed sweeps only
                obj["prediction"] = {"batches": pipe.batches, "batches_with_a_miss": pipe.fallback_batches,
                                     "pairs_missed": pipe.fallback_pairs, "pairs_compaction": pipe.compaction_pairs,
                                     "pairs_per_batch": plan.n_pairs,
                                     "listed_share_of_elements": pipe.list_share, "listed_share_max": pipe.max_share,
                                     "source": "threshold history",
                                     "thresholds_raised_per_batch": pipe.raises / max(1, pipe.batches),
                                     # 1024-element tiles holding a non-zero value outside the 2^-18 .. 2^14 window (summed apart)
                                     "tiles_with_values_outside_window_share": pipe.tiles_reread / max(1, pipe.batches * B * sum((e + 1023) // 1024 for e in plan.elems))}
            return obj

        if len(pool) < min_pool and rank == 0:
            print(f"bench.py: --pool {len(pool)} < {min_pool}: batches repeat inside the prediction's memory", file=sys.stderr)
        mse = run_mse(pool, a.mse_steps, jitter)
        # Everything below characterises ONE GPU (other schedules, other activations, other shapes, the fake-quant kernels, the CLI end
        # to end, the CPU baseline) and is reported by the N = 1 run.  A run over several ranks measures the sharded path — the hist
        # sweep and the mse sweep with their collectives — and nothing else: N processes building ViT sessions, 448 x 448 pools and
        # fake-quantised forwards side by side would only lengthen the run and fight over the host's cores.
        side = world == 1
        if not side:
            a.mse_jitter, a.real_images, a.vit_images, a.big_images, a.fq_reps, pipe1 = "", 0, 0, 0, 0, None
        mse_lanes1 = run_mse(pool, max(1, a.mse_steps // 2), jitter, pipe=pipe1) if pipe1 is not None else None
        # the same sweep over images that differ in contrast (one sweep each): what a prediction from other images costs
        for jit in ([float(x) for x in a.mse_jitter.split(",") if x] if jitter == 0.0 else []):
            jp = [synth_activations(spec, B, dev, seed=99 + 1000 * rank + j, image_jitter=jit) for j in range(len(pool))]
            mse_jitter[f"{jit:g}"] = run_mse(jp, 1, jit)
            del jp
            torch.cuda.empty_cache()
        # The same sweep over activations the repo's executor PRODUCES for ResNet-50 (random weights, random images): real layer
        # statistics — spatially smooth feature maps, where a strided sample of 32 neighbours at a time carries little and the
        # tensors settle on the prediction from earlier batches — alike, and with every image's input scaled by its own factor
        if a.real_images > 0:
            from dipoorlet_amd import models
            rsess = models.resnet50().make_session()
            assert [int(e) for e in rsess.elems_per_image] == elems
            for jit in (0.0, 0.3):
                gen = torch.Generator(device=dev)
                gen.manual_seed(777 + rank)
                rpool = []
                for _ in range(min_pool):
                    x = torch.randn(B, 3, 224, 224, generator=gen, device=dev)
                    if jit:
                        x = x * (1.0 + jit * (2.0 * torch.rand(B, 1, 1, 1, generator=gen, device=dev) - 1.0))
                    rpool.append([t.reshape(B, -1) for t in rsess.run({"input": x})])
                mse_real[f"{jit:g}"] = run_mse(rpool, 1, jit, n_images=a.real_images, net="ResNet-50 (executor-produced activations)")
                del rpool
                torch.cuda.empty_cache()
            del rsess
        # BASELINE configs[4]'s workload on one GPU: ViT-B/16 activations produced by the repo's own graph executor (every node
        # output exposed: 557 tensors, LayerNorm / erf-GELU / attention-probability tensors among them; the attention logits are
        # scaled up so that most probabilities lie below the OCTAV window, as in a trained network), N = 256 in batches of 8
        if a.vit_images > 0:
            from dipoorlet_amd import models
            vsess = models.vit_b16(seed=5, attn_gain=10.0).make_session()
            VB = 8
            vplan = ops.TensorSetPlan([int(e) for e in vsess.elems_per_image], VB, dev)
            gen = torch.Generator(device=dev)
            gen.manual_seed(4242 + rank)
            vpool = [[t.reshape(VB, -1) for t in vsess.run({"input": torch.randn(VB, 3, 224, 224, generator=gen, device=dev)})]
                     for _ in range(min_pool)]
            vit_mse = run_mse(vpool, 1, 0.0, plan=vplan, n_images=a.vit_images, net="ViT-B/16")
            del vpool, vsess
            torch.cuda.empty_cache()

        # Tensors above one OCTAV slice (1 044 480 elements per image and tensor: the packed histogram's 20-bit counts): ResNet-50's
        # shapes at 448 x 448 input — every tensor four times its size, 216 of a batch's 984 pairs of 2 .. 4 slices — in batches of 8:
        # streamed slice by slice, walked by the merge kernel
        if a.big_images > 0:
            bspec = [(n, 4 * e, k) for n, e, k in spec]
            bplan = ops.TensorSetPlan([e for _, e, _ in bspec], 8, dev)
            bpool = [synth_activations(bspec, 8, dev, seed=4321 + 1000 * rank + j) for j in range(min_pool)]
            mse_big = run_mse(bpool, 1, 0.0, plan=bplan, n_images=a.big_images, net="ResNet-50 at 448 x 448 (tensors above one slice)")
            del bpool
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the fake-quant forward (quantize.py:197-239)
    # One batch of the ResNet-50 activation set through k_fake_quant_*: per tensor (what the reference's activation Q/DQ nodes
    # do) and per channel (axis 1, the weights' granularity, on the same data): 4 B read + 4 B written per element; one launch
    # per tensor as the graph walk issues them, timed by HIP events on the launch stream around the whole sequence (123
    # launches of 0.1 .. 103 MB: the small ones are launch-bound) and around the large tensors alone.
    fake_quant = None
    if world > 1:
        a.fq_reps = 0
    if a.fq_reps > 0:
        from dipoorlet_amd.synthetic import resnet50_tensor_shapes
        shapes = resnet50_tensor_shapes()
        xs = pool[0]
        ybuf = torch.empty(B * max(elems), dtype=torch.float32, device=dev)
        qp = []
        for x, (c, h, w) in zip(xs, shapes):
            amax = x.abs().amax().clamp_min(1e-6)
            cmax = x.view(B, c, h * w).abs().amax((0, 2)).clamp_min(1e-6)
            qp.append(((amax / 127.0).reshape(1), torch.zeros(1, dtype=torch.int32, device=dev),
                       (cmax / 127.0).contiguous(), torch.zeros(c, dtype=torch.int32, device=dev)))
        fq = {}
        xv = [x.view(B, c, h * w) for x, (c, h, w) in zip(xs, shapes)]
        yv = [ybuf[:x.numel()].view(B, c, h * w) for x, (c, h, w) in zip(xs, shapes)]
        big = [i for i, e in enumerate(elems) if 4 * e * B >= 50e6]          # the tensors of >= 50 MB per batch
        for mode in ("per_tensor", "per_channel"):
            tot, tot_big = [], []
            for rep in range(a.fq_reps + 1):
                def launch(i):
                    s1, z1, sc, zc = qp[i]
                    if mode == "per_tensor":
                        ops.fake_quant(xv[i], s1, z1, -128, 127, out=yv[i])
                    else:
                        ops.fake_quant(xv[i], sc, zc, -128, 127, axis=1, out=yv[i])
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
                for i in range(T):        # the whole set, launch after launch on one stream (gaps between launches included)
                    launch(i)
                ev[1].record()
                ev[2].record()
                for i in big:
                    launch(i)
                ev[3].record()
                torch.cuda.synchronize()
                if rep > 0:   # (the first pass warms up)
                    tot.append(ev[0].elapsed_time(ev[1]))
                    tot_big.append(ev[2].elapsed_time(ev[3]))
            ms, ms_big = sum(tot) / len(tot), sum(tot_big) / len(tot_big)
            gbps = 8 * E * B / (ms * 1e-3) / 1e9
            gbps_big = 8 * B * sum(elems[i] for i in big) / (ms_big * 1e-3) / 1e9
            fq[mode] = {"ms_per_batch": ms, "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS, "launches": T,
                        "tensors_of_50MB_and_more": {"launches": len(big), "ms": ms_big, "achieved": gbps_big,
                                                     "frac": gbps_big / HBM_PEAK_GBPS}}
        # ... and the whole set in ONE launch (dpl_fake_quant_items: what a caller that holds every tensor of a forward uses)
        ys = plan.bind([torch.empty_like(x) for x in xs])
        fq["set_launch"] = {}
        for mode in ("per_tensor", "per_channel"):
            prm = [((q[0], q[1], 1, -128, 127) if mode == "per_tensor" else (q[2], q[3], h * w, -128, 127))
                   for q, (c, h, w) in zip(qp, shapes)]
            fset = ops.FakeQuantSet(plan, prm)
            fset(xs, out=ys)              # (warms up)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(a.fq_reps):  # back to back on the launch stream, as a forward issues them
                fset(xs, out=ys)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.fq_reps
            gbps = 8 * E * B / (ms * 1e-3) / 1e9
            fq["set_launch"][mode] = {"ms_per_batch": ms, "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS, "launches": 1}
        del ys
        # ... and where the PRODUCT runs it: the activation Q/DQ nodes of a fake-quantised ResNet-50 forward (quantize.quant_graph
        # for -D trt, executor.GraphSession; the weights' Q/DQ are folded at session build), one launch per node between the
        # network's own kernels, at the CLI's default batch: bytes of all Q/DQ nodes of one forward / their summed GPU time
        # (HIP events around each node on the launch stream while the stream is kept busy: kernel durations, not launch gaps)
        try:
            import types as _types
            from dipoorlet_amd import executor as _ex, models as _models
            from dipoorlet_amd.forward_net import DEFAULT_BATCH as _PB
            from dipoorlet_amd.quantize import quant_graph as _quant_graph
            from dipoorlet_amd.tensor_cali import find_clip_val_minmax_weight as _wranges
            gfp = _models.resnet50()
            sfp = gfp.make_session()
            gen = torch.Generator(device=dev)
            gen.manual_seed(99)
            xin = torch.randn(_PB, 3, 224, 224, generator=gen, device=dev)
            clipv = {n: [float(t.amin()), float(t.amax())] for n, t in zip(sfp.tensor_names, sfp.run({"input": xin}))}
            clipv.update(_wranges(gfp, None, session=sfp))
            del sfp
            gq, _ = _quant_graph(gfp, clipv, _types.SimpleNamespace(deploy="trt", skip_layers=[]))
            sq = gq.make_session()
            out_name = gq.network_outputs[0]
            orig, orig_fused = _ex._OPS["FakeQuant"], _ex.fused_fake_quant

            def timed_forward(fwd):
                """HIP events around every Q/DQ launch of `a.fq_reps` forwards (and around the forwards): (bytes of the Q/DQ nodes per
                forward — 8 per element, 12 where the residual Add is read too —, their summed milliseconds, forward ms, nodes)."""
                evs = []

                def timed(call, nbytes):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    y = call()
                    e1.record()
                    evs.append((e0, e1, nbytes))
                    return y
                fwd()
                fwd()
                _ex._OPS["FakeQuant"] = lambda sess, node, x: timed(lambda: orig(sess, node, x), 8 * x.numel())
                _ex.fused_fake_quant = lambda sess, node, pre, *xs: timed(lambda: orig_fused(sess, node, pre, *xs),
                                                                           (12 if pre == "add_relu" else 8) * xs[0].numel())
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                try:
                    f0.record()
                    for _ in range(a.fq_reps):
                        fwd()
                    f1.record()
                    torch.cuda.synchronize()
                finally:
                    _ex._OPS["FakeQuant"], _ex.fused_fake_quant = orig, orig_fused
                return (sum(n for _, _, n in evs) / a.fq_reps, sum(e0.elapsed_time(e1) for e0, e1, _ in evs) / a.fq_reps,
                        f0.elapsed_time(f1) / a.fq_reps, len(evs) // a.fq_reps)
            # every tensor exposed (profiling's per-layer pass: ReLU, Add, Q/DQ are separate launches) ...
            ub, ums, ufwd, unodes = timed_forward(lambda: sq.run({"input": xin}))
            # ... and only the output asked for (the walks of --bc / update_bn / AdaRound, a caller of the quantised network): a ReLU or
            # Add + ReLU whose only reader is a Q/DQ pair runs inside k_fake_quant<PRE> (executor.relu_fusion)
            nbytes, ms, fwd_ms, nodes = timed_forward(lambda: sq.run_named({"input": xin}, [out_name]))
            fused, skipped = sq.fusion([out_name])
            # what an (e0, e1) pair measures with NOTHING between the two records, on a stream that is kept busy the same way: the
            # events' own packets — reported beside the raw figure, never subtracted from it
            empty = []
            big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            for _ in range(64):
                big.add_(1.0)                              # (1 GB read + written: the stream stays busy)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
                p1.record()
                empty.append((p0, p1))
            torch.cuda.synchronize()
            pair_us = 1e3 * sorted(p0.elapsed_time(p1) for p0, p1 in empty)[len(empty) // 2]
            del big
            # (a Q/DQ node of this forward moves 106 MB on average: 18 us at 6 TB/s + the 2 - 3 us any launch takes to fill and
            # drain the chip; the nodes are a chain — conv, [add,] [relu +] Q/DQ, conv — so they cannot share a launch)
            fq["product_forward"] = {"batch": _PB, "form": "ReLU / Add + ReLU fused into the Q/DQ kernel (run_named: only the output asked for)",
                                     "nodes": nodes, "bytes": nbytes, "ms": ms,
                                     "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                     "forward_ms": fwd_ms, "share_of_the_quantised_forward": ms / fwd_ms,
                                     "fused_pairs": {"relu": sum(1 for p, _ in fused.values() if p == "relu"),
                                                     "add_relu": sum(1 for p, _ in fused.values() if p == "add_relu"),
                                                     "launches_saved_per_forward": len(skipped)},
                                     "every_tensor_exposed": {"nodes": unodes, "bytes": ub, "ms": ums, "frac": ub / (ums * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                              "forward_ms": ufwd},
                                     "forward_ms_saved": 1.0 - fwd_ms / ufwd,
                                     # (the kernels' own durations, rocprofv3 --kernel-trace over the same forwards:
                                     # scripts/fq_forward_prof.sh, profiles/r06/kernel_stats_fq_forward.md)
                                     "empty_event_pair_us": pair_us}
            del sq, gq, gfp, xin
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001  (a side object: the line goes out without it)
            fq["product_forward"] = {"error": repr(e)[:300]}
        fake_quant = {"workload": f"ResNet-50 activation set, one batch of {B} images, fused QuantizeLinear -> DequantizeLinear, int8 grid",
                      "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "bytes_per_batch": 8 * E * B, **fq}
        del ybuf, xs, qp, xv, yv

    # ------------------------------------------------------------------ end to end: real .onnx + .bin files through the CLI
    # (forward_net.py:192-281 is what this replaces.)  ResNet-50 written as an .onnx by the repo's graph layer, N = 1024 raw fp32
    # .bin images under a temporary directory, `python -m dipoorlet_amd ... -A hist -D trt --calib_batch 32` in a FRESH child
    # process (HIP context, MIOpen algorithm search, cold file cache all inside its wall time); the split is the child's own
    # (--timing_json: host .bin ingest seconds, GPU seconds of the network forward — library work — and of the statistics)
    e2e = None
    cpu_sample = [t.cpu() for t in pool[0]] if (rank == 0 and world == 1 and a.cpu_seconds > 0) else None
    if a.e2e_images > 0 and rank == 0 and world == 1:
        pool = None            # the child keeps its own 109 GB of activations resident between its two passes
        pipe = pipe1 = None
        plan = None
        torch.cuda.empty_cache()
        import shutil
        import tempfile
        import numpy as np
        from dipoorlet_amd import models
        d = tempfile.mkdtemp(prefix="dpl_e2e_")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        env["MASTER_PORT"] = str(29600 + os.getpid() % 300)

        def write_images(sub, n):
            os.makedirs(os.path.join(d, sub, "input"), exist_ok=True)
            rs = np.random.default_rng(0)
            base = rs.standard_normal((64, 3 * 224 * 224)).astype(np.float32)
            have = len(os.listdir(os.path.join(d, sub, "input")))
            for i in range(have, n):       # distinct images from 64 base draws (scaled): file I/O is what matters here
                (base[i % 64] * np.float32(1.0 + 0.01 * (i // 64))).tofile(os.path.join(d, sub, "input", f"{i}.bin"))

        def cli(model, sub, n, algo, tag, extra=()):
            """One fresh CLI process (python -m dipoorlet_amd ...): {command, process wall, calibration images/s, the child's split}."""
            tj = os.path.join(d, f"timing_{tag}.json")
            cmd = [sys.executable, "-m", "dipoorlet_amd", "-M", os.path.join(d, model), "-I", os.path.join(d, sub), "-N", str(n), "-A", algo,
                   "-D", "trt", "-O", os.path.join(d, "out_" + tag), "--skip_profiling", "--timing_json", tj, *extra]
            time.sleep(2.0)        # (the driver is still releasing the previous process' HBM: a fresh process right behind one that
            torch.cuda.empty_cache()   # held 100 GB pays seconds for its first allocations — not what a user's run sees)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            wall = time.perf_counter() - t0
            if r.returncode != 0 or not os.path.exists(tj):
                return {"error": (r.stderr or r.stdout)[-600:], "returncode": r.returncode}
            with open(tj) as f:
                tm = json.load(f)
            with open(os.path.join(d, "out_" + tag, "act_clip_val.json")) as f:
                n_clips = len(json.load(f))
            tm.pop("forward_batches_ms", None)
            return {"command": "python -m dipoorlet_amd -M %s -I %s -N %d -A %s -D trt --skip_profiling %s" % (model, sub, n, algo, " ".join(extra)),
                    "process_wall_s": wall, "images_per_s_process": n / wall,
                    "images_per_s_calibration": n / tm["tensor_calibration_wall_s"], "tensors": n_clips, "split": tm}
        try:
            g = models.resnet50()
            g.output_dir = d
            g.save_onnx_model("r50")
            del g
            write_images("calib", a.e2e_images)
            cli("r50.onnx", "calib", min(64, a.e2e_images), "minmax", "warm")   # (untimed: the box's page cache sees the libraries and the files)
            e2e = cli("r50.onnx", "calib", a.e2e_images, "hist", "hist")
            if "error" not in e2e:
                # the same files through `-A mse` (BASELINE configs[2]'s algorithm), then configs[2]'s own N = 4096
                e2e["mse"] = cli("r50.onnx", "calib", a.e2e_images, "mse", "mse")
                if a.e2e_images >= 1024:
                    write_images("calib", N_MSE)
                    e2e["mse_4096"] = cli("r50.onnx", "calib", N_MSE, "mse", "mse4096")
                # configs[4]'s network on one GPU: ViT-B/16 (557 exposed tensors, 133 M elements per image), -A mse, N = 256
                if a.vit_images > 0:
                    g = models.vit_b16(seed=5, attn_gain=10.0)
                    g.output_dir = d
                    g.save_onnx_model("vit")
                    del g
                    cli("vit.onnx", "calib", 16, "minmax", "vitwarm")      # (untimed, as for ResNet-50: the BLAS library's files — which a
                    # ResNet-50 process no longer opens — and the model file enter the box's page cache)
                    e2e["vit_mse"] = cli("vit.onnx", "calib", min(a.vit_images, 256), "mse", "vit")
        finally:
            shutil.rmtree(d, ignore_errors=True)

    # ------------------------------------------------------------------ the line
    kernel_bytes = 4 * E * B                                   # k_abs_hist reads the batch once
    achieved = kernel_bytes / (hist_kern_ms * 1e-3) / 1e9 if hist_kern_ms > 0 else 0.0
    # HBM bytes per launch by the PMC counters (scripts/profile_gpu.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    # this very script).  Quoted only when the record was measured on the kernel sources this run uses (sha over csrc/), else null.
    def traffic_record():
        tj = os.environ.get("DPL_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "r06", "traffic.json"))
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from summarize_prof import source_sha
            with open(tj) as f:
                tr = json.load(f)
            if tr.get("source_sha") == source_sha() and tr.get("batch") == B:
                return tr["kernels"]
        except Exception:
            pass
        return None
    tk = traffic_record()
    traffic = tk["k_abs_hist"]["hbm_bytes_per_launch"] if tk and "k_abs_hist" in tk else None
    # (`traffic` is NOT counted in this run: it is the PMC record of scripts/profile_gpu.sh over this script, replayed when its
    # source hash matches the kernels this run uses — the line names the file)
    traffic_from = os.path.relpath(os.environ.get("DPL_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "r06", "traffic.json")), ROOT) if tk else None

    def mse_batch_traffic():
        """HBM bytes of ONE mse batch: every k_octav_* kernel of the profiled run, per launch of the streaming kernel."""
        if not tk:
            return None
        ks = {k: v for k, v in tk.items() if k.startswith("k_octav_")}
        main = "k_octav_tail"
        if main not in ks:
            return None
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ks.values()) / ks[main]["launches"]
    if mse is not None:
        mse["roofline"]["traffic"] = mse_batch_traffic()
    if mse_lanes1 is not None:
        mse_lanes1["roofline"]["kernel"] = "OCTAV batch on ONE stream (OctavPipeline(lanes=1): forward_net.forward_net_octav's schedule; the rescue beside the next batch)"
    images = N_HIST * world * a.steps
    hist_rate = images / dt_hist
    headline_mse = a.algo == "mse" and mse is not None

    def brief(o, short=False):
        """The scalars of an mse object the record must carry (the whole object goes to the `details` line; short: the fraction,
        the milliseconds and the spot check only)."""
        if not o:
            return None
        r, p = o["roofline"], o.get("prediction") or {}
        b = {"frac": round(r["frac"], 4), "ms": round(r["avg_batch_ms"], 4), "ok": o["sample_ok"]}
        if short:
            return b
        if r.get("traffic"):
            b["traffic_ratio"] = round(r["traffic"] / r["bytes_per_launch"], 4)
        if p:
            b.update(listed=round(p["listed_share_of_elements"], 4), rescued=round(p["pairs_missed"] / max(1, p["batches"]), 1))
        if o.get("scratch_over_batch_activations") is not None:
            b["scratch"] = o["scratch_over_batch_activations"]
        return b
    hist_roof = {"bound": "hbm", "kernel": "k_abs_hist", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_from": traffic_from, "bytes_per_launch": kernel_bytes,
                 "avg_kernel_ms": hist_kern_ms}
    full = {
        "hist": {"images_per_s": hist_rate, "ms_per_step": dt_hist / a.steps * 1e3, "roofline": hist_roof,
                 "algorithmic_GBps_job": 8 * E * images / dt_hist / 1e9, "hist_checksum": hist_checksum,
                 "hist_checksum_expected": E * N_HIST * world, "clip_checksum": clip_checksum, "collectives_ms_per_sweep": coll_ms},
        "mse": mse, "mse_lanes1": mse_lanes1, "mse_jitter": mse_jitter or None, "fake_quant": fake_quant, "vit_mse": vit_mse, "mse_448": mse_big,
        "mse_feature_maps": mse_real or None, "e2e": e2e,
    }
    # The record's line: BASELINE.json's metric on its headline configuration, every headline scalar inside `roofline` / `config`
    # (the driver keeps those objects whole), under 2 KB.  Everything else (workload strings, prediction statistics, the e2e split)
    # is the `details` line printed BEFORE it.
    roof = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (mse["roofline"] if headline_mse else hist_roof).items()}
    roof["mse"] = brief(mse)                                                     # BASELINE configs[2]: -A mse, N = 4096, images alike
    roof["mse_lanes1"] = brief(mse_lanes1)                                       # ... on one stream, as forward_net_octav schedules it
    roof["mse_jitter"] = {k: brief(v, short=True) for k, v in mse_jitter.items()} or None    # ... with per-image contrast jitter
    roof["mse_feature_maps"] = {k: brief(v) for k, v in mse_real.items()} or None   # ... executor-produced ResNet-50 activations
    roof["mse_vit"] = brief(vit_mse)                                             # configs[4]'s workload on one GPU
    roof["mse_448"] = brief(mse_big)                                             # tensors above one OCTAV slice (ResNet-50 at 448 x 448)
    if fake_quant:
        # per mode: [frac of one launch per tensor over the set, ... over the tensors of >= 50 MB, ... of the set in ONE launch]
        roof["fake_quant"] = {m: [round(fake_quant[m]["frac"], 4), round(fake_quant[m]["tensors_of_50MB_and_more"]["frac"], 4),
                                  round(fake_quant["set_launch"][m]["frac"], 4)] for m in ("per_tensor", "per_channel")}
        # the 4th: the Q/DQ nodes of a fake-quantised ResNet-50 forward through the product's executor (per tensor, -D trt)
        if "frac" in fake_quant.get("product_forward", {}):
            roof["fake_quant"]["product_forward"] = round(fake_quant["product_forward"]["frac"], 4)
            # the same nodes by the KERNELS' durations (rocprofv3 kernel trace, scripts/fq_forward_prof.sh): quoted from the committed
            # record while the kernel sources are the ones it was taken on (the live figure above brackets every node with HIP events)
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from summarize_prof import source_sha as _sha
                with open(os.path.join(ROOT, "profiles", "r06", "fq_forward.json")) as f:
                    _fq = json.load(f)
                if _fq.get("source_sha") == _sha():
                    roof["fake_quant"]["product_forward_kernel_trace"] = round(_fq["frac_of_8TBps"], 4)
                    fake_quant["product_forward"]["kernel_trace"] = {"frac": _fq["frac_of_8TBps"], "us_per_forward": _fq["us_per_forward"],
                                                                     "forward_gpu_time_saved": _fq.get("gpu_time_saved"),
                                                                     "from": "profiles/r06/fq_forward.json"}
            except Exception:   # noqa: BLE001
                pass
    if e2e and "error" not in e2e:
        # per run: [images/s of calibration (fresh CLI process over .bin files), images/s of the network forward in steady state]
        def pair(o):
            return [round(o["images_per_s_calibration"]), round(o["split"].get("forward_steady_images_per_s", 0.0))]
        roof["e2e"] = {"hist": pair(e2e)}
        for k in ("mse", "mse_4096", "vit_mse"):
            if k in e2e and "error" not in e2e[k]:
                roof["e2e"][k] = pair(e2e[k])
        sp = e2e["split"]
        # the headline run's wall against its GPU work: tensor_calibration_wall_s - (forward_gpu_s + statistics_gpu_s)
        roof["e2e"]["hist_host_s"] = round(sp["tensor_calibration_wall_s"] - sp["forward_gpu_s"] - sp["statistics_gpu_s"], 3)
        # the statistics kernels INSIDE the product (one stream, between two network forwards): one read of every image's activation
        # set / the GPU seconds the child's HIP events give them, as a fraction of 8 TB/s — [hist: two reads, mse, mse N = 4096]
        def stats_frac(o, reads):
            sp_ = o["split"]
            return round(reads * 4.0 * E * sp_["images"] / max(sp_["statistics_gpu_s"], 1e-9) / 1e9 / HBM_PEAK_GBPS, 4)
        roof["e2e"]["hist_stats_frac"] = stats_frac(e2e, 2)
        for k in ("mse", "mse_4096"):
            if k in e2e and "error" not in e2e[k]:
                roof["e2e"][k + "_stats_frac"] = stats_frac(e2e[k], 1)
    out = {
        # BASELINE.json's metric; images/s is `value`, the achieved HBM GB/s of the dominant kernel is `roofline.achieved`
        "metric": "calibration images/sec (whole node) + achieved HBM GB/s, ResNet-50 ONNX N=%d, -A %s"
                  % ((N_MSE, "mse") if headline_mse else (N_HIST, "hist")),
        "value": round(mse["value"] if headline_mse else hist_rate, 1), "unit": "images/s", "n_gpus": world,
        "steps": a.mse_steps if headline_mse else a.steps, "warmup": a.warmup,
        "ms_per_step": round(mse["ms_per_step"] if headline_mse else dt_hist / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"ResNet-50 activation set (T={T}, {E} fp32 elems/img), -A mse (OCTAV), N={N_MSE} per GPU in batches of {B}, cold sweeps"
                                if headline_mse else
                                f"ResNet-50 ONNX activation set (T={T}, {E} fp32 elems/img), -A hist --bins {a.bins}, N={N_HIST} per GPU in "
                                f"{n_hist_batches} batches of {B}: range pass + histogram pass + percentile clip"),
                   "device": devname, "hist_checksum_ok": hist_checksum == E * N_HIST * world,
                   "world_size_seen_by_backend": dist.get_world_size() if use_dist else 1,
                   "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if use_dist else None,
                   "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if (use_dist and backend == "nccl") else None,
                   "per_rank_images_per_s": [round(r) for r in per_rank_rates],
                   "collectives_ms_per_sweep": coll_ms},
        "roofline": roof,
    }
    if rank == 0:
        if world == 1 and a.cpu_seconds > 0:
            cb = cpu_baseline(a.algo, a.bins, cpu_sample, a.cpu_seconds)
            full["cpu_baseline"] = dict(cb)
            out["cpu_baseline"] = {"value": round(cb["value"], 2), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": cb["sample"].split(" (")[0] + f" of the same set, -A {a.algo}, C oracle + OpenMP (memory-bound: "
                                             f"{cb['c_openmp_images_per_s'] / max(cb['numpy_single_thread_images_per_s'], 1e-9):.1f} x one numpy thread)",
                                   "numpy_1_thread": round(cb["numpy_single_thread_images_per_s"], 2)}
        else:
            out["cpu_baseline"] = None
        # RCCL prints a version banner through C stdio, which — stdout being a pipe — would leave its buffer only at exit, BEHIND the
        # record line: flush C's buffers first so that the record line stays the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        print(json.dumps({"details": full}), flush=True)
        line = json.dumps(out)
        # (the driver keeps the last 2 000 characters of stdout as `tail`: should the line ever outgrow that, the side objects go
        # first — they are all in the details line above —, never the contract's fields)
        for k in ("mse_448", "mse_feature_maps", "mse_jitter", "fake_quant", "mse_vit", "mse_lanes1", "e2e"):
            if len(line) < 1990:
                break
            out["roofline"].pop(k, None)
            line = json.dumps(out)
        print(line, flush=True)
    if use_dist:
        dist.destroy_process_group()
    if rank != 0:          # (only rank 0's stdout carries lines; nothing a library still holds may follow them on the others' either)
        try:
            sys.stdout.flush()
        except Exception:   # noqa: BLE001
            pass


if __name__ == "__main__":
    main()
