"""Synthetic calibration activations with the exact tensor multiset of the BASELINE networks.

The reference exposes EVERY node output of the (BN-folded, onnxsim-simplified) graph plus the network
input as calibration tensors (forward_net.py:193-198, 220-235).  For torchvision-topology ResNet-50
that is 123 tensors / 26,598,376 fp32 elements per image; ResNet-18: 50 tensors / 5,897,704.
No checkpoints or datasets exist offline, so values are seeded random: pre-activation tensors
~ N(0, (1 + 0.1 t)^2), ReLU / pooling outputs clamped at 0 (~50 % exact zeros).
"""
import torch


def _bottleneck_net(blocks, widths):
    """(name, elems_per_image, kind) in topological order; kind in {'pre', 'relu'}."""
    out = [("input", 3 * 224 * 224, "pre")]
    out.append(("conv1", 64 * 112 * 112, "pre"))
    out.append(("relu1", 64 * 112 * 112, "relu"))
    out.append(("maxpool", 64 * 56 * 56, "relu"))
    sp = 56
    for li, (nb, w) in enumerate(zip(blocks, widths)):
        for bi in range(nb):
            osp = sp // 2 if (bi == 0 and li > 0) else sp
            p = f"layer{li + 1}.{bi}"
            out.append((p + ".conv1", w * sp * sp, "pre"))
            out.append((p + ".relu1", w * sp * sp, "relu"))
            out.append((p + ".conv2", w * osp * osp, "pre"))
            out.append((p + ".relu2", w * osp * osp, "relu"))
            out.append((p + ".conv3", 4 * w * osp * osp, "pre"))
            if bi == 0:
                out.append((p + ".downsample", 4 * w * osp * osp, "pre"))
            out.append((p + ".add", 4 * w * osp * osp, "pre"))
            out.append((p + ".relu3", 4 * w * osp * osp, "relu"))
            sp = osp
    c = 4 * widths[-1]
    out += [("avgpool", c, "relu"), ("flatten", c, "relu"), ("fc", 1000, "pre")]
    return out


def _basic_net(blocks, widths):
    out = [("input", 3 * 224 * 224, "pre"), ("conv1", 64 * 112 * 112, "pre"), ("relu1", 64 * 112 * 112, "relu"),
           ("maxpool", 64 * 56 * 56, "relu")]
    sp = 56
    for li, (nb, w) in enumerate(zip(blocks, widths)):
        for bi in range(nb):
            osp = sp // 2 if (bi == 0 and li > 0) else sp
            p = f"layer{li + 1}.{bi}"
            out.append((p + ".conv1", w * osp * osp, "pre"))
            out.append((p + ".relu1", w * osp * osp, "relu"))
            out.append((p + ".conv2", w * osp * osp, "pre"))
            if bi == 0 and li > 0:
                out.append((p + ".downsample", w * osp * osp, "pre"))
            out.append((p + ".add", w * osp * osp, "pre"))
            out.append((p + ".relu2", w * osp * osp, "relu"))
            sp = osp
    c = widths[-1]
    out += [("avgpool", c, "relu"), ("flatten", c, "relu"), ("fc", 1000, "pre")]
    return out


def resnet50_tensors():
    return _bottleneck_net((3, 4, 6, 3), (64, 128, 256, 512))


def resnet18_tensors():
    return _basic_net((2, 2, 2, 2), (64, 128, 256, 512))


def resnet50_tensor_elems():
    return [e for _, e, _ in resnet50_tensors()]


def resnet50_tensor_shapes():
    """(C, H, W) per image of every tensor of resnet50_tensors(), same order (the fully connected ones as (C, 1, 1))."""
    out = [(3, 224, 224), (64, 112, 112), (64, 112, 112), (64, 56, 56)]
    sp = 56
    for li, (nb, w) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for bi in range(nb):
            osp = sp // 2 if (bi == 0 and li > 0) else sp
            out += [(w, sp, sp), (w, sp, sp), (w, osp, osp), (w, osp, osp), (4 * w, osp, osp)]
            if bi == 0:
                out.append((4 * w, osp, osp))
            out += [(4 * w, osp, osp), (4 * w, osp, osp)]
            sp = osp
    out += [(2048, 1, 1), (2048, 1, 1), (1000, 1, 1)]
    assert [c * h * w for c, h, w in out] == resnet50_tensor_elems()
    return out


def synth_activations(spec, batch, device, seed=1234, image_jitter=0.0):
    """One batched tensor set on the device.  `spec`: list of (name, elems, kind) or of bare elems.
    image_jitter > 0 scales every image of every tensor by its own factor in [1 - jitter, 1 + jitter] (images of real
    calibration sets differ in contrast; used to stress the one-read OCTAV form's bin prediction)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = []
    for t, s in enumerate(spec):
        if isinstance(s, (tuple, list)):
            _, e, kind = s
        else:
            e, kind = int(s), ("relu" if t % 2 else "pre")
        x = torch.randn(batch, e, generator=g, device=device, dtype=torch.float32)
        x.mul_(1.0 + 0.1 * t)
        if image_jitter > 0.0:
            x.mul_(1.0 + image_jitter * (2.0 * torch.rand(batch, 1, generator=g, device=device) - 1.0))
        if kind == "relu":
            x.clamp_(min=0)
        out.append(x)
    return out
