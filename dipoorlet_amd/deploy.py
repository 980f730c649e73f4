"""Platform emitters: clip ranges -> the vendor files the reference writes (dipoorlet/deploy/*).

Pure host-side dict -> file code, no GPU work; kept so a run ends with the same artefacts.  File names,
JSON structure (indent=4) and value formulas follow the reference emitters cited per function.  The two
large emitters (rv: deploy_rv.py:23-178, stpu: deploy_stpu.py:23-222) walk vendor-specific layer tables
and are not reproduced (out of scope); asking for them logs the reference's warning.
"""
import json
import os

import numpy as np

from .platform_settings import platform_setting_table
from .utils import dispatch_functool, logger


@dispatch_functool
def deploy_dispatcher(*args, **kwargs):
    logger.warning("Deploy Platform Not Found!")


def _dump(obj, args, fname):
    with open(os.path.join(args.output_dir, fname), "w") as f:
        json.dump(obj, f, indent=4)


@deploy_dispatcher.register("trt")
def gen_trt_range(graph, clip_val, args, **kwargs):
    """deploy_trt.py:7-16 — {"blob_range": {tensor: max(-lo, hi)}} -> trt_clip_val.json."""
    for k in clip_val:
        clip_val[k] = max(-float(clip_val[k][0]), float(clip_val[k][1]))
    _dump({"blob_range": clip_val}, args, "trt_clip_val.json")


@deploy_dispatcher.register("snpe")
def gen_snpe_encodings(graph, clip_val, args, **kwargs):
    """deploy_snpe.py:7-34 — activation_encodings for every non-initializer node input and network output."""
    enc = {}

    def entry(t):
        lo, hi = float(clip_val[t][0]), float(clip_val[t][1])
        return [{"bitwidth": 8, "min": lo, "max": max(max(0.0, hi), lo + 0.01)}]
    for node in graph.graph.node:
        for t in node.input:
            if t != "" and t not in graph.initializer:
                enc[t] = entry(t)
    for t in graph.network_outputs:
        enc[t] = entry(t)
    _dump({"activation_encodings": enc, "param_encodings": {}}, args, "snpe_encodings.json")


@deploy_dispatcher.register("ti")
def gen_ti_json(graph, clip_val, args, **kwargs):
    """deploy_ti.py:7-19 — ti_blob_range.txt (name lo hi) and ti_blob_range.json."""
    with open(os.path.join(args.output_dir, "ti_blob_range.txt"), "w") as f:
        for k, v in clip_val.items():
            f.write("{} {} {}\n".format(k, v[0], v[1]))
    for k, v in clip_val.items():
        clip_val[k] = [float(x) for x in v]
    _dump({"blob_range": clip_val}, args, "ti_blob_range.json")


@deploy_dispatcher.register("imx")
def gen_imx_range(graph, clip_val, args, **kwargs):
    """deploy_imx.py:8-26 — power-of-two scales, '.bias' entries dropped -> imx_scale.json."""
    for k in [k for k in clip_val if k.endswith(".bias")]:
        del clip_val[k]
    for k in clip_val:
        scale = np.array(np.max(np.abs(clip_val[k]), axis=0)) / [2 ** 7 - 1]
        scale = np.where(scale == 0, 1., scale)
        clip_val[k] = (2 ** np.round(np.log2(scale))).tolist()
    _dump({"blob_range": clip_val}, args, "imx_scale.json")


@deploy_dispatcher.register("magicmind")
def gen_magicmind_proto(graph, clip_val, args, **kwargs):
    """deploy_magicmind.py:9-20 — {"blob_range": {t: {"min","max"}}} -> magicmind_quant_param.json."""
    out = {k: {"min": float(np.min(v[0])), "max": float(np.max(v[1]))} for k, v in clip_val.items()}
    _dump({"blob_range": out}, args, "magicmind_quant_param.json")


@deploy_dispatcher.register("atlas")
def gen_atlas_quant_param(graph, clip_val, args, **kwargs):
    """deploy_atlas.py:10-32 — per quantised layer input: scale = (max(0,hi)-min(0,lo))/255 (0 -> 1),
    offset = round(-lo'/scale) - 128."""
    res = {}
    for node in graph.graph.node:
        if node.op_type in platform_setting_table["atlas"]["quant_nodes"]:
            t = node.input[0]
            lo, hi = min(0, clip_val[t][0]), max(0, clip_val[t][1])
            step = (hi - lo) / 255.
            if step == 0.0:
                step = 1.0
            res[t] = {"scale": step, "offset": int(round(-lo / step) - 128)}
    _dump(res, args, "atlas_quant_param.json")


def to_deploy(graph, act_clip_val, weight_clip_val, args, **kwargs):
    """deploy_base.py:13-19."""
    if platform_setting_table[args.deploy]["deploy_weight"]:
        clip_val = act_clip_val.copy()
        clip_val.update(weight_clip_val)
    else:
        clip_val = act_clip_val
    deploy_dispatcher(args.deploy, graph, clip_val, args, **kwargs)
