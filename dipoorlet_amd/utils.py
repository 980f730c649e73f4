"""Host-side plumbing shared by the calibration path: the dispatch registry (the reference's plugin
API), the clip-range JSON exchange, logging.

Mirrors, without sharing code with, dipoorlet/utils.py:281-303 (dispatch_functool) and :313-368
(save_clip_val / reduce_clip_val / load_clip_val): same call forms, same file names, byte-compatible
JSON (indent=4, {tensor_name: [lo, hi]}), same merge arithmetic.
"""
import json
import logging
import os

import numpy as np

from .platform_settings import platform_setting_table

logger = logging.getLogger("dipoorlet")

MARKS = {}     # name -> time.time() of the first pass through mark(name): a fresh process' timeline (--timing_json)


def mark(name):
    import time
    MARKS.setdefault(name, time.time())


class _Dispatcher:
    """`d = dispatch_functool(default)`; `@d.register(key)`; `d(key, *args, **kw)` calls the function
    registered under key, or `default(*args, **kw)` for an unknown key (utils.py:281-303)."""

    def __init__(self, default):
        self._default = default
        self.registry = {}
        self.__name__ = getattr(default, "__name__", "dispatcher")
        self.__doc__ = default.__doc__

    def dispatch(self, key):
        return self.registry.get(key, self._default)

    def register(self, key, func=None):
        if func is None:
            return lambda f: self.register(key, f)
        self.registry[key] = func
        return func

    def __call__(self, key, *args, **kw):
        return self.dispatch(key)(*args, **kw)


def dispatch_functool(func):
    return _Dispatcher(func)


# ------------------------------------------------------------------ clip-range JSON exchange
def _listify(v):
    return v.tolist() if hasattr(v, "tolist") else v


_FLOAT_REPR = float.__repr__
_STR = json.encoder.encode_basestring_ascii


def _json_text(o, pad, out):
    """Appends to `out` the text json.dumps(o, indent=4) yields for o at indentation `pad`."""
    if isinstance(o, dict):
        if not o:
            out.append("{}")
            return
        inner = pad + "    "
        sep = "{\n" + inner
        for k, v in o.items():
            if not isinstance(k, str):
                raise TypeError("non-string key")
            out.append(sep + _STR(k) + ": ")
            _json_text(v, inner, out)
            sep = ",\n" + inner
        out.append("\n" + pad + "}")
    elif isinstance(o, (list, tuple)):
        if not o:
            out.append("[]")
            return
        inner = pad + "    "
        try:    # a flat list of floats (a per-channel range: thousands of them) in one join
            body = (",\n" + inner).join(map(_FLOAT_REPR, o))
            if "n" in body:     # nan / inf: json spells them NaN / Infinity
                raise TypeError
            out.append("[\n" + inner + body + "\n" + pad + "]")
            return
        except TypeError:
            pass
        sep = "[\n" + inner
        for v in o:
            out.append(sep)
            _json_text(v, inner, out)
            sep = ",\n" + inner
        out.append("\n" + pad + "]")
    elif isinstance(o, str):
        out.append(_STR(o))
    elif o is None or o is True or o is False or isinstance(o, (int, float)):
        out.append(json.dumps(o))
    else:
        raise TypeError(type(o).__name__)


def dump_indent4(obj, f):
    """json.dump(obj, f, indent=4), byte for byte (the format of the reference's clip and profiling files, utils.py:313-323),
    without json's pure-Python encoder for indented output: a per-channel weight_clip_val.json holds ~10^5 numbers and json.dump
    spent 0.3 s of a 2.3 s run on the two files and their merged copies.  Falls back to json.dump for anything unusual."""
    out = []
    try:
        _json_text(obj, "", out)
    except TypeError:
        json.dump(obj, f, indent=4)
        return
    f.write("".join(out))


def save_clip_val(act_clip_val, weight_clip_val, args, act_fname="act_clip_val.json",
                  weight_fname="weight_clip_val.json"):
    """Writes {name: [lo, hi]} with indent=4.  Like the reference (utils.py:313-323) the dict values are
    converted to plain python numbers/lists IN PLACE."""
    for d in (act_clip_val, weight_clip_val):
        for k in d:
            d[k][0] = _listify(d[k][0])
            d[k][1] = _listify(d[k][1])
    with open(os.path.join(args.output_dir, act_fname), "w") as f:
        dump_indent4(act_clip_val, f)
    with open(os.path.join(args.output_dir, weight_fname), "w") as f:
        dump_indent4(weight_clip_val, f)


def load_clip_val(args, act_fname="act_clip_val.json", weight_fname="weight_clip_val.json"):
    """Activation ranges come back as np.float64 scalars; weight ranges as arrays, collapsed to scalars
    when the platform quantises weights per tensor (utils.py:348-368)."""
    with open(os.path.join(args.output_dir, act_fname)) as f:
        act = json.load(f)
    for k in act:
        act[k][0] = np.float64(act[k][0])
        act[k][1] = np.float64(act[k][1])
    per_channel = bool(platform_setting_table[args.deploy]["qw_params"].get("per_channel", False))
    with open(os.path.join(args.output_dir, weight_fname)) as f:
        wt = json.load(f)
    for k in wt:
        wt[k][0] = np.array(wt[k][0])
        wt[k][1] = np.array(wt[k][1])
        if not per_channel:
            wt[k][0] = np.float64(wt[k][0])
            wt[k][1] = np.float64(wt[k][1])
    return act, wt


def reduce_clip_val(rank_size, args, act_fname="act_clip_val.json", weight_fname="weight_clip_val.json", already_merged=False):
    """Rank-0 merge of the per-rank files `<fname>.rank<r>` (utils.py:326-345): minmax -> elementwise
    min / max; hist and mse -> sum over ranks of value / rank_size (rank order); weights from rank 0.
    already_merged: every rank already holds the clips of the WHOLE calibration set (statistics merged over RCCL): rank 0's
    files are the result as they are — averaging W identical values would only perturb the last fp64 bit (v / 6 * 6 != v)."""
    act, wt = load_clip_val(args, act_fname + ".rank0", weight_fname + ".rank0")
    if already_merged:
        save_clip_val(act, wt, args)
        return
    mean_mode = args.act_quant != "minmax"
    w = float(rank_size)
    if mean_mode:
        for v in act.values():
            v[0] /= w
            v[1] /= w
    for r in range(1, rank_size):
        with open(os.path.join(args.output_dir, f"{act_fname}.rank{r}")) as f:
            other = json.load(f)
        for k, v in other.items():
            if mean_mode:
                act[k][0] += v[0] / w
                act[k][1] += v[1] / w
            else:
                act[k] = [np.array(min(v[0], act[k][0])), np.array(max(v[1], act[k][1]))]
    save_clip_val(act, wt, args)


# ------------------------------------------------------------------ profiling result exchange (utils.py:371-412)
def save_profiling_res(layer_cosine_dict, model_cosine_dict, args, layer_res_fname="layer_res.json",
                       model_res_fname="model_res.json"):
    """utils.py:371-383 — this rank's cosine tables as `<fname>.rank<r>` (the layer table only without --model_type)."""
    rank = getattr(args, "rank", 0)
    if getattr(args, "model_type", None) is None:
        with open(os.path.join(args.output_dir, f"{layer_res_fname}.rank{rank}"), "w") as f:
            json.dump(layer_cosine_dict, f, indent=4)
    with open(os.path.join(args.output_dir, f"{model_res_fname}.rank{rank}"), "w") as f:
        json.dump(model_cosine_dict, f, indent=4)


def reduce_profiling_res(rank_size, args, layer_res_fname="layer_res.json", model_res_fname="model_res.json"):
    """utils.py:386-412 — mean over ranks of the per-rank layer cosines and of the network outputs' mean cosine (each rank
    weighted 1 / rank_size, summed in rank order), minimum over ranks of the outputs' worst cosine."""
    def read(fname, r):
        with open(os.path.join(args.output_dir, f"{fname}.rank{r}")) as f:
            return json.load(f)
    w = float(rank_size)
    layer = {}
    if getattr(args, "model_type", None) is None:
        layer = {k: v / w for k, v in read(layer_res_fname, 0).items()}
        for r in range(1, rank_size):
            for k, v in read(layer_res_fname, r).items():
                layer[k] += v / w
    model = {k: [v[0] / w, v[1]] for k, v in read(model_res_fname, 0).items()}
    for r in range(1, rank_size):
        for k, v in read(model_res_fname, r).items():
            model[k][0] += v[0] / w
            model[k][1] = min(model[k][1], v[1])
    return layer, model


def setup_logger(args=None, level=logging.INFO):
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("[%(asctime)s %(name)s](%(filename)s %(lineno)d): %(levelname)s %(message)s"))
        logger.addHandler(h)
    logger.setLevel(level)
    if args is not None and getattr(args, "output_dir", None):
        fh = logging.FileHandler(os.path.join(args.output_dir, "log.txt"))
        logger.addHandler(fh)
    return logger
