"""The rows round 1 left unpinned, held to what the REFERENCE itself computes (tests/golden/aux_level.* from
gen_golden_aux.py): cos_similarity (utils.py:273-278), update_conv_node_bias (bias_correction.py:9-31),
reduce_profiling_res (utils.py:386-412) and WHICH tensors quant_graph fake-quantises (quantize.py:20-108).

CPU tests: the oracle restatements and the host logic.  GPU tests (`-m gpu`): the kernels through the C ABI."""
import copy
import json
import os
import types

import numpy as np
import pytest

from _cases import AUX_GRAPH, aux_cos_pair, aux_stack
from oracle import np_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def aux():
    with open(os.path.join(HERE, "golden", "aux_level.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(HERE, "golden", "aux_level.npz"))


# ------------------------------------------------------------------------------------------------ CPU: oracle + host logic
def test_oracle_cos_similarity_golden(aux):
    meta, _ = aux
    for row in meta["cos"]:
        a, b = aux_cos_pair(row["case"])
        got = O.cos_similarity(a, b)
        assert np.float64(got) == np.float64(row["cos"]), (row, got)          # bit for bit (numpy fp32 arithmetic)


def test_oracle_bias_delta_golden(aux):
    meta, arr = aux
    for row in meta["bias"]:
        fp, q = aux_stack(row["case"], row["n"], row["C"], row["hw"])
        delta = O.bias_correction_delta(list(fp), list(q), row["op"] == "Conv")
        base = (np.arange(row["C"], dtype=np.float32) * np.float32(0.25) - np.float32(1.0)) if row["has_bias"] else 0
        want = arr[f"bias/{row['case']}"]
        got = base + delta
        assert got.dtype == want.dtype and np.array_equal(got, want), row


def _write_rank_files(od, run):
    for r, pr in enumerate(run["per_rank"]):
        if run["model_type"] is None:
            with open(os.path.join(od, f"layer_res.json.rank{r}"), "w") as f:
                json.dump(pr["layer"], f, indent=4)
        with open(os.path.join(od, f"model_res.json.rank{r}"), "w") as f:
            json.dump(pr["model"], f, indent=4)


def test_reduce_profiling_res_golden(aux, tmp_path):
    from dipoorlet_amd.utils import reduce_profiling_res, save_profiling_res
    meta, _ = aux
    for i, run in enumerate(meta["profiling"]):
        layers = [pr["layer"] for pr in run["per_rank"]] if run["model_type"] is None else None
        ol, om = O.reduce_profiling_res(layers, [copy.deepcopy(pr["model"]) for pr in run["per_rank"]])
        assert ol == run["layer"] and om == run["model"]                       # exact: same float operations, same order
        od = tmp_path / f"run{i}"
        od.mkdir()
        for r, pr in enumerate(run["per_rank"]):                                # through the package's own writer
            save_profiling_res(pr["layer"], pr["model"], types.SimpleNamespace(output_dir=str(od), rank=r,
                                                                                 model_type=run["model_type"]))
        layer, model = reduce_profiling_res(run["world"], types.SimpleNamespace(output_dir=str(od), model_type=run["model_type"]))
        assert layer == run["layer"] and model == run["model"]


def build_aux_graph():
    """AUX_GRAPH on this package's ONNXGraph."""
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.onnx_io import Node
    g = ONNXGraph()
    g.graph.node = [Node(n["op"], list(n["in"]), list(n["out"]), name=n["name"]) for n in AUX_GRAPH["nodes"]]
    g.initializer = {k: np.zeros((c, 1), np.float32) for k, c in AUX_GRAPH["initializers"].items()}
    g.network_inputs, g.network_outputs = list(AUX_GRAPH["inputs"]), list(AUX_GRAPH["outputs"])
    g.input = list(AUX_GRAPH["inputs"]) + list(AUX_GRAPH["initializers"])
    g.tensor_name_shape_map = {t: [1] for t in AUX_GRAPH["tensors"]}
    g.tensor_name_shape_map.update({k: [c, 1] for k, c in AUX_GRAPH["initializers"].items()})
    g.topologize_graph()
    g.set_index()
    return g


def test_quant_graph_selection_golden(aux):
    """The set (and order) of fake-quantised tensors, the re-wired node inputs, the node order and the new network
    outputs for a graph that exercises merge-ReLU, the TensorRT Add rule, dedupe, ConvTranspose weights, a two-input Mul,
    a PRelu on the network input, --skip_layers and quantize_network_output — against the reference's own run."""
    from dipoorlet_amd.quantize import quant_graph
    meta, _ = aux
    clip = {t: [np.float64(-1.0 - 0.1 * i), np.float64(2.0 + 0.1 * i)] for i, t in enumerate(AUX_GRAPH["tensors"])}
    for k, c in AUX_GRAPH["initializers"].items():
        clip[k] = [-np.ones(c), np.ones(c)]
    for run in meta["selection"]:
        args = types.SimpleNamespace(deploy=run["deploy"], skip_layers=run["skip_layers"])
        gq, qlist = quant_graph(build_aux_graph(), copy.deepcopy(clip), args)
        fq = [n for n in gq.graph.node if n.op_type == "FakeQuant"]
        assert [n.name for n in qlist] == run["quant_node_list"], run["deploy"]
        assert sorted(n.input[0] for n in fq) == sorted(run["quantized_in_order"]), run["deploy"]
        assert {n.name: list(n.input) for n in gq.graph.node if n.op_type != "FakeQuant"} == run["node_inputs"], run["deploy"]
        # node order: the reference inserts <t>_QuantizeLinear, <t>_DequantizeLinear where this package inserts one node
        want = [n for n in run["node_order"] if not n.endswith("_DequantizeLinear")]
        assert [n.name for n in gq.graph.node] == want, run["deploy"]
        assert gq.network_outputs == run["network_outputs"], run["deploy"]


# ------------------------------------------------------------------------------------------------ GPU: the kernels
@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_cos_kernels_vs_reference_golden(dev, aux):
    """k_cos_acc and the batched k_cos_items (fp64 sums) against the reference's float32 cos_similarity: the reference
    rounds every product and sum to fp32, so 2e-6 relative is the comparison the arithmetic allows; the dot == 0 branch
    and the all-zero tensor give exactly 0."""
    import torch
    from dipoorlet_amd import ops
    meta, _ = aux
    for row in meta["cos"]:
        a, b = aux_cos_pair(row["case"])
        ta, tb = torch.from_numpy(a.reshape(-1)).to(dev), torch.from_numpy(b.reshape(-1)).to(dev)
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        ops.cos_accumulate(ta, tb, acc)
        plan = ops.TensorSetPlan([ta.numel()], 1, dev)
        acc2 = ops.cos_per_image(plan, [ta.reshape(1, -1)], [tb.reshape(1, -1)])[0, 0]
        for s in (acc.cpu().numpy(), acc2.cpu().numpy()):
            got = 0.0 if s[0] == 0 else s[0] / np.sqrt(s[1]) / np.sqrt(s[2])
            if row["cos"] == 0.0:
                assert got == 0.0, row
            else:
                assert abs(got - row["cos"]) <= 2e-6 * abs(row["cos"]), (row, got)


@pytest.mark.gpu
def test_update_conv_node_bias_vs_reference_golden(dev, aux):
    """update_conv_node_bias on the device (k_channel_diff_sum, fp64 sums) against the reference's numpy float32 mean over
    prescribed fp / quantised stacks: Conv and Gemm, with an existing bias and with the add-a-bias branch (new
    initializer `<node>_bias`, appended to the node's inputs), the stacks fed in one chunk and in ragged chunks."""
    import torch
    from dipoorlet_amd.graph import ONNXGraph
    from dipoorlet_amd.onnx_io import Node
    from dipoorlet_amd.weight_transform.bias_correction import update_conv_node_bias
    meta, arr = aux
    for row in meta["bias"]:
        fp, q = aux_stack(row["case"], row["n"], row["C"], row["hw"])
        fp_t = torch.from_numpy(np.squeeze(fp, 1)).to(dev)
        q_t = torch.from_numpy(np.squeeze(q, 1)).to(dev)
        for cuts in ([row["n"]], [1, row["n"] - 1]):
            g = ONNXGraph()
            node = Node(row["op"], ["x", "w"] + (["b"] if row["has_bias"] else []), ["y"], name=f"node{row['case']}")
            g.graph.node = [node]
            g.initializer = {"w": np.zeros((row["C"], 1), np.float32)}
            if row["has_bias"]:
                g.initializer["b"] = np.arange(row["C"], dtype=np.float32) * np.float32(0.25) - np.float32(1.0)
            g.input = list(g.initializer)
            update_conv_node_bias(g, node, list(torch.split(fp_t, cuts)), list(torch.split(q_t, cuts)))
            assert list(node.input) == row["node_inputs_after"] and g.input[-1] == row["graph_input_appended"]
            got = g.get_initializer(row["bias_name"])
            want = arr[f"bias/{row['case']}"]
            assert got.dtype == np.float32 and got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7)
