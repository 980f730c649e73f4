"""Platform emitters: clip ranges -> the vendor files the reference writes (dipoorlet/deploy/*).

Pure host-side dict -> file code, no GPU work; kept so a run ends with the same artefacts.  File names,
JSON structure (indent=4) and value formulas follow the reference emitters cited per function (all eight
platforms; the rv / stpu tables are checked against files the reference's own emitters wrote).
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


# ------------------------------------------------------------------------------------------------ rv (Rockchip)
def _affine_u8(rng):
    """deploy_rv.py:11-20 — asymmetric 8-bit step / zero point of a range that is first widened to contain 0."""
    lo, hi = min(0, np.min(rng[0])), max(0, np.max(rng[1]))
    step = (hi - lo) / 255.
    if step == 0.0:
        step = 1.0 / 255.
    return {"scale": [float(step)], "zero_point": [int(round(-lo / step))]}


def _zero_spanning(rng):
    return max(0., float(np.max(rng[1]))), min(0., float(np.min(rng[0])))


def _feeds_sigmoid(graph, node):
    nxt = graph.get_tensor_consumer(node.output[0])
    return len(nxt) == 1 and not isinstance(nxt[0], str) and nxt[0].op_type == "Sigmoid"


def _concat_ranges_to_inputs(graph, clip_val):
    """deploy_rv.py:29-33 — every input of a Concat takes the Concat output's range."""
    for node in graph.graph.node:
        if node.op_type == "Concat":
            for t in node.input:
                clip_val[t][0], clip_val[t][1] = clip_val[node.output[0]][0], clip_val[node.output[0]][1]


def _rv1126_table(graph, clip_val):
    """deploy_rv.py:25-107 — '@<node>:out<i>' / ':weight' / ':bias' entries, asymmetric_affine u8 (bias i32 with
    scale = weight step * input step); a Relu shares its entry with its producer."""
    from .platform_settings import LAYER_HAS_WEIGHT
    q = {}

    def tensor_entry(rng):
        hi, lo = _zero_spanning(rng)
        e = {"dtype": "asymmetric_affine", "method": "layer", "max_value": [hi], "min_value": [lo], "qtype": "u8"}
        e.update(_affine_u8(rng))
        return e
    for name in graph.network_inputs:
        e = tensor_entry([clip_val[name][0], clip_val[name][1]])
        e["max_value"], e["min_value"] = [max(0., float(clip_val[name][1]))], [min(0., float(clip_val[name][0]))]
        q[f"@{name}:out0"] = e
    for node in graph.graph.node:
        if _feeds_sigmoid(graph, node):
            continue
        if node.op_type in LAYER_HAS_WEIGHT:
            for idx, t in enumerate(node.input[1:]):
                if idx == 0:
                    q[f"@{node.name}:weight"] = tensor_entry(clip_val[t])
                elif idx == 1:
                    q[f"@{node.name}:bias"] = {
                        "dtype": "asymmetric_affine", "method": "layer", "max_value": [], "min_value": [],
                        "zero_point": [0],
                        "scale": [_affine_u8(clip_val[node.input[1]])["scale"][0] * _affine_u8(clip_val[node.input[0]])["scale"][0]],
                        "qtype": "i32"}
                else:
                    print("We meet unsupported node{}, skip.".format(node.name))
        key = None
        for idx, t in enumerate(node.output):
            key = f"@{node.name}:out{idx}"
            q[key] = tensor_entry(clip_val[t])
        if node.op_type == "Relu":
            prev = graph.get_tensor_producer(node.input[0])
            for k in q:
                if prev.name in k and "out" in k:
                    q[k] = q[key]
    return {"customized_quantize_layers": {}, "quantize_parameters": q}


def _rk3568_table(graph, clip_val):
    """deploy_rv.py:109-175 — entries keyed by tensor name ('<node>_W' / '<node>_b' for parameters) with plain
    min / max lists; biases symmetric; a Relu's input takes the Relu output's entry."""
    from .platform_settings import LAYER_HAS_WEIGHT
    q = {}
    for name in graph.network_inputs:
        q[name] = {"max": [max(0., float(clip_val[name][1]))], "min": [min(0., float(clip_val[name][0]))]}
    for node in graph.graph.node:
        if _feeds_sigmoid(graph, node):
            continue
        if node.op_type in LAYER_HAS_WEIGHT:
            for idx, t in enumerate(node.input[1:]):
                if idx == 0:
                    hi, lo = _zero_spanning(clip_val[t])
                    q[f"{node.name}_W"] = {"max": [hi], "min": [lo]}
                elif idx == 1:
                    m = float(max(abs(np.max(clip_val[node.input[2]])), abs(np.min(clip_val[node.input[2]]))))
                    q[f"{node.name}_b"] = {"max": [m], "min": [-m]}
                else:
                    print("We meet unsupported node{}, skip.".format(node.name))
        key = None
        for t in node.output:
            hi, lo = _zero_spanning(clip_val[t])
            key = t
            q[key] = {"max": [hi], "min": [lo]}
        if node.op_type == "Relu":
            q[node.input[0]] = q[key]
    return {"custom_quantize_layers": {}, "quantize_parameters": q}


@deploy_dispatcher.register("rv")
def gen_rv_yaml(graph, clip_val, args, **kwargs):
    """deploy_rv.py:23-178 — rv_quantized_param.{yaml,json} (RV1126 form) and rk_quantized_param.{yaml,json}
    (RK3568 form)."""
    import yaml
    _concat_ranges_to_inputs(graph, clip_val)
    for stem, table in (("rv_quantized_param", _rv1126_table(graph, clip_val)),
                        ("rk_quantized_param", _rk3568_table(graph, clip_val))):
        with open(os.path.join(args.output_dir, stem + ".yaml"), "w") as f:
            f.write(yaml.dump(table))
        _dump(table, args, stem + ".json")


# ------------------------------------------------------------------------------------------------ stpu
def _float_exponent(v):
    """deploy_stpu.py:103-115 — the biased fp32 exponent e with 2^(e-127) <= v < 2^(e-126), 0 for 0, clamped to
    [1, 254]."""
    if abs(v) == 0:
        return 0
    for e in range(1, 254):
        if 2 ** (e - 127) <= v < 2 ** (e - 126):
            return e
    return 1 if v < 2 ** (-126) else 254


def _conv_emin(i_vmax, w_vmax, o_vmax, n, r):
    """deploy_stpu.py:131-135."""
    return _float_exponent(max(n ** .5 * i_vmax * w_vmax, o_vmax)) - (12 - r)


def _winograd_weight_range(ker):
    """deploy_stpu.py:87-93 — range of G k G^T over all 3x3 kernels (F(2x2, 3x3) transform, un-normalised G)."""
    g = np.array([[2, 0, 0], [1, 1, 1], [1, -1, 1], [0, 0, 2]], dtype="float32")
    wu = np.einsum("ab,ijbc,dc->ijad", g, np.asarray(ker, np.float64), g)
    return max(wu.max(), 0), min(wu.min(), 0)


@deploy_dispatcher.register("stpu")
def gen_stpu_minmax(graph, clip_val, args, **kwargs):
    """deploy_stpu.py:23-222 — stpu_minmax.json: symmetric ranges per weight ('<node>_weights') and tensor, ReLU /
    Clip inputs sharing their output's range, optional winograd weight ranges (--stpu_wg), the accumulator exponent
    'emin' of Conv / ConvTranspose / Gemm / Upsample / Corr outputs, and bias scales."""
    from .platform_settings import LAYER_HAS_WEIGHT
    param = {}

    def sym(lo, hi):
        m = max(np.abs(lo), hi)
        return {"min": float(-m), "max": float(m)}
    for node in graph.graph.node:                                   # :38-46
        if node.op_type in LAYER_HAS_WEIGHT:
            param[node.name + "_weights"] = sym(np.min(clip_val[node.input[1]][0]), np.max(clip_val[node.input[1]][1]))
    for t in graph.network_inputs:                                  # :49-62
        param[t] = sym(clip_val[t][0], clip_val[t][1])
    for node in graph.graph.node:
        for t in node.output:
            param[t] = sym(clip_val[t][0], clip_val[t][1])
    for node in graph.graph.node:                                   # :65-68
        if node.op_type in ("Relu", "Clip"):
            param[node.input[0]] = param[node.output[0]].copy()
    if getattr(args, "stpu_wg", False):                             # :71-100
        for node in graph.graph.node:
            if (node.op_type == "Conv" and int(node.attrs.get("group", 1)) == 1
                    and list(node.attrs.get("kernel_shape", [])) == [3, 3]
                    and list(node.attrs.get("strides", [1, 1])) == [1, 1] and "layer_" + node.name not in param):
                param["layer_" + node.name] = {"wg": True}
                vmax, vmin = _winograd_weight_range(graph.get_initializer(node.input[1]))
                m = max(vmax, -vmin)
                param[node.name + "_weights"]["max"], param[node.name + "_weights"]["min"] = float(m), float(-m)
    for node in graph.graph.node:                                   # :152-208
        out = node.output[0]
        if node.op_type in ("Upsample", "DynamicUpsample"):
            param[out]["emin"] = _float_exponent(param[out]["max"]) - (22 - 2)
        elif node.op_type in ("Conv", "ConvTranspose", "Gemm"):
            if node.op_type == "Gemm":
                n = np.prod(graph.get_tensor_shape(node.input[0]))
            else:
                ws = graph.get_tensor_shape(node.input[1])
                n = ws[1] * ws[2] * ws[3]
            param[out]["emin"] = _conv_emin(param[node.input[0]]["max"], param[node.name + "_weights"]["max"],
                                            param[out]["max"], n, 2)
        elif node.op_type == "Corr":
            n = np.prod(graph.get_tensor_shape(node.input[0])) / node.attrs["groups"]
            param[out]["emin"] = _float_exponent(param[out]["max"] * n ** .5) - (12 - 4)
    for node in graph.graph.node:                                   # :211-222
        if node.op_type in ("Conv", "ConvTranspose", "Gemm") and len(node.input) == 3:
            w, i = param[node.name + "_weights"], param[node.input[0]]
            param[node.name + "_bias"] = {"alpha": (w["max"] - w["min"]) / (2 ** 8 - 2) * ((i["max"] - i["min"]) / (2 ** 8 - 2)),
                                          "zero_point": 0}
    _dump(param, args, "stpu_minmax.json")
