"""Per-backend quantisation settings (data only).

Same keys and values as the reference table (dipoorlet/platform_settings.py:1-184) because the
calibration hot path reads them verbatim: `'dynamic_sym' in qi_params` decides OCTAV's `unsigned`
factor (forward_net.py:319) and qi/qw params feed scale / zero-point derivation (quantize.py:111-194).
Expressed through a small builder instead of eight literal dicts.
"""

LAYER_HAS_WEIGHT = ["Conv", "Gemm", "ConvTranspose", "PRelu", "BatchNormalization"]

_BASIC = ["Relu", "Eltwise", "MaxPool", "Conv", "Gemm", "ConvTranspose", "PRelu", "AveragePool", "Concat",
          "Split", "Add", "Mul", "Abs", "Reciprocal", "Sigmoid"]


def _lin(symmetric, **extra):
    d = {"bit_width": 8, "type": "Linear", "symmetric": symmetric}
    d.update(extra)
    return d


def _platform(quant_nodes, qw, qi, net_out=False, deploy_weight=False, exclude=True):
    d = {"quant_nodes": list(quant_nodes), "qw_params": qw, "qi_params": qi,
         "quantize_network_output": net_out, "deploy_weight": deploy_weight}
    if exclude:
        d["deploy_exclude_layers"] = []
    return d


platform_setting_table = {
    "trt": _platform(["Relu", "MaxPool", "Conv", "Gemm", "ConvTranspose", "PRelu", "AveragePool", "Add",
                      "Sigmoid"], _lin(True, per_channel=True), _lin(True)),
    "stpu": _platform(_BASIC + ["Clip", "HardSigmoid"], _lin(True, per_channel=False), _lin(True),
                      deploy_weight=True),
    "magicmind": _platform(["Gemm", "Conv", "ConvTranspose", "MatMul"],
                           _lin(False, log_scale=False, per_channel=True), _lin(False, log_scale=False)),
    "rv": _platform(_BASIC, _lin(False, per_channel=False), _lin(False), net_out=True, deploy_weight=True),
    "atlas": _platform(["Conv", "Gemm", "AveragePool"], _lin(True, per_channel=True), _lin(False),
                       exclude=False),
    "snpe": _platform(_BASIC + ["Sigmoid"], _lin(False, per_channel=False), _lin(False), net_out=True),
    "ti": _platform(_BASIC, _lin(True, per_channel=False, log_scale=False),
                    _lin(True, dynamic_sym=True, log_scale=True)),
    "imx": _platform(_BASIC, _lin(True, per_channel=True, log_scale=True), _lin(True, log_scale=True),
                     net_out=True, deploy_weight=True),
}
