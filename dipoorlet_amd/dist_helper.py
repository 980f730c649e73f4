"""Process-group bootstrap and the cross-rank merge of calibration statistics.

Bootstrap mirrors dipoorlet/dist_helper.py:8-49 and __main__.py:57-64 (env -> MASTER_ADDR/PORT, RANK,
WORLD_SIZE; backend 'nccl' = RCCL on ROCm; rank -> device = rank % device_count) with one addition the
reference lacks: with no GPU visible it falls back to the 'gloo' backend instead of dividing by zero
(BASELINE configs[0] is a CPU plumbing case).

Merge: the reference exchanges per-rank clip JSON files on a shared filesystem and rank 0 averages
them (__main__.py:121-128, utils.py:326-345).  Here the *statistics* are merged with three small
collectives over RCCL/xGMI, each on one packed buffer, so every rank ends with the statistics of the
whole calibration set and the result equals the reference's own world_size = 1 answer:
    ranges      all_reduce(MIN) on [T] fp32 mins, all_reduce(MAX) on [T] fp32 maxes   (after pass 1)
    histograms  all_reduce(SUM) on [T, bins] int64                                     (after pass 2)
    OCTAV       all_gather of the per-image [n, T, 3] (s, min, max) rows
"""
import os
import re

import torch
import torch.distributed as dist


def _backend():
    """'nccl' (= RCCL on ROCm) whenever a GPU is visible, as the reference hard-codes (__main__.py:62); 'gloo'
    otherwise.  DPL_DIST_BACKEND overrides (used to run two ranks on ONE GPU in tests: RCCL refuses that)."""
    forced = os.environ.get("DPL_DIST_BACKEND")
    if forced:
        return forced
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # one rank: every merge is the identity and only barriers remain; RCCL's communicator set-up (0.3 - 1 s of a fresh
        # process) buys nothing, gloo's is milliseconds
        return "gloo"
    return "nccl" if torch.cuda.is_available() and torch.cuda.device_count() > 0 else "gloo"


TIMES = {}   # host seconds of the bootstrap's two parts (reported by --timing_json): "group" = the process group,
             # "device" = the first touch of the HIP runtime (torch.cuda.is_available + set_device)


def _bind_device():
    import time
    t0 = time.perf_counter()
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        torch.cuda.set_device(dist.get_rank() % torch.cuda.device_count())
    TIMES["device"] = time.perf_counter() - t0


def init_default():
    """__main__.py:61-64 — env:// rendezvous as set up by torch.distributed.run."""
    import time
    t0 = time.perf_counter()
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if int(os.environ["WORLD_SIZE"]) == 1 and not os.environ.get("DPL_DIST_BACKEND"):
            # one rank: an in-process store — no TCP rendezvous (a port still in TIME_WAIT from the previous run of a script
            # cost a fresh process 1.3 s of bind retries)
            dist.init_process_group(backend="gloo", store=dist.HashStore(), rank=0, world_size=1)
        else:
            dist.init_process_group(backend=_backend())
    TIMES["group"] = time.perf_counter() - t0
    _bind_device()


def init_from_mpi():
    """dist_helper.py:8-23 — OpenMPI launch: rank/size from OMPI_COMM_WORLD_*, master from the HNP URI."""
    if "MASTER_ADDR" not in os.environ:
        m = re.search(r".*tcp://((\d{1,3}\.){3}\d{1,3})[:,].*", os.environ["OMPI_MCA_orte_hnp_uri"])
        os.environ["MASTER_ADDR"] = m.group(1)
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ["WORLD_SIZE"] = os.environ["OMPI_COMM_WORLD_SIZE"]
    os.environ["RANK"] = os.environ["OMPI_COMM_WORLD_RANK"]
    dist.init_process_group(backend=_backend())
    _bind_device()


def slurm_master_addr(node_list):
    """dist_helper.py:32-41 — first host of SLURM_NODELIST ('prefix-a-b-c-d' style names -> a.b.c.d)."""
    if "[" in node_list:
        beg = node_list.find("[")
        p1 = node_list.find("-", beg)
        p2 = node_list.find(",", beg)
        p1 = 1000 if p1 < 0 else p1
        p2 = 1000 if p2 < 0 else p2
        node_list = node_list[:min(p1, p2)].replace("[", "")
    return node_list[8:].replace("-", ".")


def init_from_slurm():
    """dist_helper.py:26-49."""
    job_id = int(os.environ["SLURM_JOB_ID"])
    os.environ["MASTER_PORT"] = str(24553 + job_id % 10000)
    os.environ["MASTER_ADDR"] = slurm_master_addr(os.environ["SLURM_NODELIST"])
    os.environ["WORLD_SIZE"] = os.environ["SLURM_NTASKS"]
    os.environ["RANK"] = os.environ["SLURM_PROCID"]
    dist.init_process_group(backend=_backend())
    _bind_device()


def shard_range(data_num, rank, world_size):
    """forward_net.py:207-209 — contiguous floor split; the last data_num % world_size images are unused."""
    rank_num = data_num // world_size
    return rank * rank_num, min((rank + 1) * rank_num, data_num)


# ---------------------------------------------------------------------------------- statistic merges
def _active(world_size):
    return world_size > 1 and dist.is_available() and dist.is_initialized()


def merge_ranges(gmin, gmax, world_size):
    """In place: element-wise MIN of mins and MAX of maxes over ranks.  NaN (a tensor that saw a NaN on
    any rank) must win like it does in numpy: NaN is mapped to -inf / +inf for the collective and back."""
    if not _active(world_size):
        return gmin, gmax
    nan = (torch.isnan(gmin) | torch.isnan(gmax)).to(torch.int32)
    lo = torch.where(nan.bool(), torch.full_like(gmin, float("-inf")), gmin)
    hi = torch.where(nan.bool(), torch.full_like(gmax, float("inf")), gmax)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(nan, op=dist.ReduceOp.MAX)
    gmin.copy_(torch.where(nan.bool(), torch.full_like(lo, float("nan")), lo))
    gmax.copy_(torch.where(nan.bool(), torch.full_like(hi, float("nan")), hi))
    return gmin, gmax


def merge_hist(hist, world_size):
    """In place: SUM of the [T, bins] int64 histograms over ranks (one 2 MB buffer for ResNet-50)."""
    if _active(world_size):
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def gather_rows(rows, world_size):
    """[n, T, 3] per-image OCTAV rows of this rank -> [world*n, T, 3] in rank (= image) order."""
    if not _active(world_size):
        return rows
    out = torch.empty((world_size * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    try:
        dist.all_gather_into_tensor(out, rows.contiguous())
    except (RuntimeError, NotImplementedError):  # a backend without the fused form (gloo with device tensors)
        parts = [torch.empty_like(rows) for _ in range(world_size)]
        dist.all_gather(parts, rows.contiguous())
        out = torch.cat(parts)
    return out
