"""`python -m dipoorlet_amd -M model.onnx -I calib_dir -N 1024 -A hist -D trt` — the reference's CLI
(dipoorlet/__main__.py:23-161) over the MI355X calibration core.

Same flags; same phases where they are in scope: load model -> tensor calibration (sharded over ranks)
-> per-rank clip JSON -> rank-0 reduce -> load -> profiling (cosine similarity of the fake-quantised
model, optional) -> weight transforms (--bc, --we, --update_bn, --adaround, --brecq [--drop], --sparse) -> platform deploy file.
Extra flags: --calib_batch, --resident_gb, --merge {allreduce,reference}, --skip_profiling, --timing_json.
"""
import argparse
import copy
import os
import sys
import time

import torch.distributed as dist

from . import dist_helper
from .deploy import to_deploy
from .graph import ONNXGraph
from .tensor_cali import tensor_calibration
from .utils import MARKS, load_clip_val, logger, mark, reduce_clip_val, save_clip_val, setup_logger


def build_parser():
    p = argparse.ArgumentParser(prog="dipoorlet_amd")
    p.add_argument("-M", "--model", help="onnx model")
    p.add_argument("-I", "--input_dir", help="calibration data", required=True)
    p.add_argument("-O", "--output_dir", help="output data path")
    p.add_argument("-N", "--data_num", help="num of calibration pics", type=int, required=True)
    for flag in ("--we", "--bc", "--update_bn", "--adaround", "--brecq", "--drop", "--savefp", "--stpu_wg",
                 "--skip_prof_layer", "--slurm", "--mpirun", "--sparse", "--optim_transformer"):
        p.add_argument(flag, default=False, action="store_true")
    p.add_argument("-A", "--act_quant", choices=["minmax", "hist", "mse"], default="mse",
                   help="minmax / hist: bit-exact clip ranges; mse (OCTAV): within 1e-5 * max(1, |ref|) of the reference's, repeating "
                        "to about 1e-6 relative from run to run (DPL_OCTAV_FORM=bracket: the bit-stable two-read form)")
    p.add_argument("-D", "--deploy", choices=["trt", "stpu", "magicmind", "rv", "atlas", "snpe", "ti", "imx"],
                   required=True)
    p.add_argument("--bins", default=2048, type=int)  # the reference omits type= and crashes on a CLI value
    p.add_argument("--threshold", default=0.99999, type=float)
    p.add_argument("--ada_bs", type=int, default=64)
    p.add_argument("--ada_epoch", type=int, default=5000)
    p.add_argument("--skip_layers", default=[], type=str, nargs="+")
    p.add_argument("--sparse_rate", type=float, default=0.5)
    p.add_argument("--pattern", choices=["unstruction", "nv24"], default="unstruction")
    p.add_argument("--model_type", choices=["unet"], default=None)
    p.add_argument("--quant_format", default="QDQ", type=str, choices=["QOP", "QDQ"])
    # MI355X-side knobs
    p.add_argument("--calib_batch", type=int, default=None,
                   help="calibration images per forward (default: by the graph's size — about 8 GB of exposed activations per batch, at most 64); "
                        "--bc, --update_bn, profiling and AdaRound / BRECQ walk the set in chunks of this size too (16 when not given)")
    p.add_argument("--resident_gb", type=float, default=160.0, help="HBM budget for keeping pass-1 activations")
    p.add_argument("--merge", choices=["allreduce", "reference"], default="allreduce")
    p.add_argument("--skip_profiling", default=False, action="store_true")
    p.add_argument("--timing_json", default=None, help="rank 0 writes where the calibration time went (host .bin ingest, GPU "
                                                        "forward, GPU statistics, wall) to this file")
    p.add_argument("--keep_bn", default=False, action="store_true",
                   help="do not fold BatchNormalization into the preceding Conv / Gemm (the reference always simplifies; "
                        "note: --update_bn also keeps the BN nodes, which it re-estimates — unlike the reference, whose "
                        "onnxsim pass has fused them before --update_bn runs)")
    return p


def main(argv=None):
    """One rank of a calibration run.  A rank that fails must not leave its peers parked at the next barrier (the
    reference's ranks hang until the launcher is killed, __main__.py:105-110): the error is logged and the process exits
    non-zero at once, which makes torch.distributed.run / mpirun / srun take the whole group down."""
    try:
        return _main(argv)
    except SystemExit:
        _join_helpers()
        raise
    except BaseException as e:   # noqa: BLE001
        import traceback
        traceback.print_exc()
        logger.error("rank %s failed: %s", os.environ.get("RANK", "0"), e)
        sys.stdout.flush()
        sys.stderr.flush()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            os._exit(1)   # no atexit / destructor may wait on a collective the peers will never join
        _join_helpers()
        raise


def _join_helpers():
    """The warm-up threads (executor.warm_libraries, GraphSession.prewarm_convs) are done within tenths of a second; a run that
    fails before its first forward must not exit underneath them."""
    ex = sys.modules.get(__package__ + ".executor")
    if ex is not None:
        ex.join_helpers()


def _process_age_s():
    """Seconds since this process was created (interpreter start-up and imports included)."""
    try:
        with open("/proc/self/stat") as f:
            start_ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / os.sysconf("SC_CLK_TCK")
    except Exception:   # noqa: BLE001
        return None


def _main(argv=None):
    t_enter = time.time()
    mark("main:enter")
    if os.environ.get("DPL_SWITCH_INTERVAL"):      # (a tuning aid: the interpreter's thread switch interval, seconds)
        sys.setswitchinterval(float(os.environ["DPL_SWITCH_INTERVAL"]))
    age_at_enter = _process_age_s()       # interpreter + imports (torch, the package) up to here
    args = build_parser().parse_args(argv)
    if args.quant_format == "QOP":
        raise SystemExit("--quant_format QOP (onnxruntime's QOperator export, dipoorlet/utils.py:415-435) is not built: "
                         "use the default QDQ format")
    if args.slurm:
        dist_helper.init_from_slurm()
    elif args.mpirun:
        dist_helper.init_from_mpi()
    else:
        dist_helper.init_default()
    t_dist = time.time()
    mark("main:group_and_device")
    rank, world = dist.get_rank(), dist.get_world_size()
    import torch
    if os.environ.get("DPL_DETERMINISTIC") == "1":
        # MIOpen's default convolution kernels are not bit-reproducible from call to call (1e-6 relative on an activation): a run
        # that has to repeat bit for bit asks the library for its deterministic algorithms (slower; the first use compiles them)
        torch.backends.cudnn.deterministic = True
    warm = None
    if torch.cuda.is_available():
        # the HIP context and the libraries' first calls: on a helper thread from here on, beside the model load and the
        # session build (executor.warm_libraries; its own clock reports the context's seconds)
        from . import executor
        dev0 = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
        executor.warm_libraries(dev0, blas=False)
        warm = executor._WARM
        try:    # the op types of the model file, without its tensors (milliseconds): a transformer's matrix products go to the BLAS
            # library, whose first call (0.2 s) then starts now, beside the model load.  (A convolutional network's one or two run
            # on ops.gemm_small; the session starts this thread if they turn out not to.)
            from . import onnx_io
            ops_seen = onnx_io.scan_op_types(args.model)
            if ops_seen.get("MatMul", 0) + ops_seen.get("Gemm", 0) > 4:
                executor.warm_blas(dev0)
        except Exception:   # noqa: BLE001  (best effort: the load below reports what is wrong with the file)
            pass
    t_ctx = time.time()
    if args.output_dir is None:
        args.output_dir = os.path.join(os.path.abspath(os.path.dirname(args.model)), "results")
    if args.model_type is not None:      # __main__.py:71-73 (the onnxruntime transformer optimiser step is not run:
        args.optim_transformer = True    # the graph is executed as it is)
        args.skip_prof_layer = True
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        setup_logger(args)
    dist.barrier()
    start = time.time()
    onnx_graph = ONNXGraph.load(args.model, args.output_dir, args.deploy, args.model_type)
    if not args.keep_bn and not args.update_bn:     # __main__.py:101 — onnxsim's Conv + BN fusion
        n_folded = onnx_graph.fold_batchnorm()
        if n_folded and rank == 0:
            logger.info("Folded {} BatchNormalization nodes into their producers.".format(n_folded))
    args.rank, args.world_size = rank, world
    args.local_rank = rank % max(1, torch.cuda.device_count())
    if rank == 0:
        logger.info("Do tensor calibration...")
    t_cal = time.time()
    mark("main:calibration_starts")
    prof_path = os.environ.get("DPL_PROFILE_HOST")     # (a tuning aid: cProfile of this rank's calibration phase, top entries to that file)
    if prof_path:
        import cProfile
        import pstats
        prof = cProfile.Profile()
        act_clip_val, weight_clip_val = prof.runcall(tensor_calibration, onnx_graph, args)
        with open(prof_path, "w") as f:
            pstats.Stats(prof, stream=f).sort_stats("cumtime").print_stats(60)
    else:
        act_clip_val, weight_clip_val = tensor_calibration(onnx_graph, args)
    if args.timing_json and rank == 0:
        import json
        from .forward_net import CalibrationRun
        mark("main:calibration_done")
        tm = CalibrationRun.last.timing() if CalibrationRun.last is not None else {}
        # seconds after main() was entered at which each point of a fresh process was first passed (warm:* = the helper thread)
        tm["timeline_s"] = {k: round(v - t_enter, 4) for k, v in sorted(MARKS.items(), key=lambda kv: kv[1])}
        tm.update(tensor_calibration_wall_s=time.time() - t_cal, load_model_wall_s=t_cal - start, act_quant=args.act_quant,
                  calib_batch=getattr(CalibrationRun.last, "batch", args.calib_batch), world_size=world,
                  # the fixed costs of a fresh process, itemised: interpreter + imports, process group, HIP context (the
                  # first batch's MIOpen algorithm search is forward_first_batch_gpu_s above)
                  startup={"interpreter_and_imports_s": age_at_enter,
                           "process_group_init_s": dist_helper.TIMES.get("group", t_dist - t_enter),
                           "process_group_backend": dist.get_backend(),
                           # first touch of the HIP runtime + the context (seconds longer right after a process that held
                           # most of the HBM has exited: the driver is still releasing it)
                           "hip_context_s": dist_helper.TIMES.get("device", 0.0) + (t_ctx - t_dist) + ((warm or {}).get("context_s") or 0.0),
                           "library_warm_threads_s": {k: round(v, 4) for k, v in (warm or {}).items() if k in ("kernels_s", "blas_s")},
                           "until_calibration_starts_s": t_cal - t_enter})
        with open(args.timing_json, "w") as f:
            json.dump(tm, f)
        CalibrationRun.last = None     # (a run pins its session, accumulators and every plan's scratch)
    tensor_range = copy.deepcopy(act_clip_val)
    save_clip_val(act_clip_val, weight_clip_val, args, act_fname=f"act_clip_val.json.rank{rank}",
                  weight_fname=f"weight_clip_val.json.rank{rank}")
    dist.barrier()
    if rank == 0:
        reduce_clip_val(world, args, already_merged=(args.merge != "reference"))
    dist.barrier()
    act_clip_val, weight_clip_val = load_clip_val(args)
    from .weight_transform import weight_calibration
    if rank == 0:
        logger.info("Weight transform...")
    graph_after_wt, graph_ori, act_clip_val, weight_clip_val = weight_calibration(onnx_graph, act_clip_val,
                                                                                  weight_clip_val, args)
    dist.barrier()
    if not args.skip_profiling:
        from .profiling import (quantize_profiling_multipass, quantize_profiling_transformer, show_model_profiling_res,
                                show_model_ranges, weight_need_perchannel)
        if rank == 0:
            logger.info("Profiling...")
        prof = quantize_profiling_transformer if args.model_type is not None else quantize_profiling_multipass   # :141-146
        layer_cos, model_cos, qnodes = prof(graph_after_wt, graph_ori, act_clip_val, weight_clip_val, args)
        if rank == 0:
            show_model_profiling_res(graph_after_wt, layer_cos, model_cos, qnodes, args)
            show_model_ranges(graph_after_wt, act_clip_val, weight_clip_val, args)
            weight_need_perchannel(graph_after_wt, args)
    if rank == 0:
        logger.info("Deploy to " + args.deploy + "...")
        to_deploy(graph_after_wt, act_clip_val, weight_clip_val, args)
        logger.info("Total time cost: {} seconds.".format(int(time.time() - start)))
    dist.barrier()
    _ = tensor_range
    _join_helpers()
    return 0


if __name__ == "__main__":
    sys.exit(main())
