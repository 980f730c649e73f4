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
