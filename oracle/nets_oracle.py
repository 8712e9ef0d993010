"""oracle/nets_oracle.py - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement of the reference networks' forward pass, driven directly by a
state_dict (no nn.Module tree).  The arithmetic is the reference's own third-party arithmetic
(PyTorch ATen / oneDNN fp32 kernels behind torch.nn.functional - SURVEY.md section 8c), so on
identical weights it reproduces the reference up to oneDNN's blocking choices (checked against
the committed golden heat maps in tests/test_oracle_golden.py).

  resnet_dconv_forward  <- nets/pose_resnet_dconv.py:251-265 (_forward_impl), Bottleneck.forward :112-133,
                            _make_layer :201-228, _make_deconv_layer :230-249, final_layer :173-178
  resnet_duc_forward    <- nets/pose_resnet_duc.py (_forward_impl), _make_duc_layer :227-232, nets/commons.py:21-43
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

RESNET50_BLOCKS = (3, 4, 6, 3)
BN_EPS = 1e-5


def _bn(x, sd, prefix, training=False):
    # nn.BatchNorm2d(eps=1e-5, momentum=0.1); eval mode uses running stats
    if training:  # batch statistics; running stats (when present in sd) are updated in place like nn.BatchNorm2d does
        return F.batch_norm(x, sd.get(prefix + ".running_mean"), sd.get(prefix + ".running_var"), sd[prefix + ".weight"],
                            sd[prefix + ".bias"], True, 0.1, BN_EPS)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.1, BN_EPS)


def _se(x, sd, prefix):
    # nets/commons.py:4-18 (reduction=1: C -> C -> C)
    y = F.adaptive_avg_pool2d(x, 1)
    y = F.relu(F.conv2d(y, sd[prefix + ".fc.0.weight"], sd[prefix + ".fc.0.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[prefix + ".fc.2.weight"], sd[prefix + ".fc.2.bias"]))
    return x * y


def _bottleneck(x, sd, p, stride, training=False):
    # nets/pose_resnet_dconv.py:112-133 (stride sits on conv2: "ResNet v1.5", :84-88)
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1", training))
    w2 = sd[p + ".conv2.weight"]                     # (resnext*: groups = 32, :101 - read off the weight's shape)
    out = F.relu(_bn(F.conv2d(out, w2, stride=stride, padding=1, groups=out.shape[1] // w2.shape[1]), sd, p + ".bn2", training))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3", training)
    if (p + ".se.fc.0.weight") in sd:
        out = _se(out, sd, p + ".se")
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1", training)
    return F.relu(out + x)


def _res_basic_block(x, sd, p, stride, training=False):
    # nets/pose_resnet_dconv.py:61-80 (BasicBlock of resnet18 / resnet34: the stride sits on conv1)
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), sd, p + ".bn1", training))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], padding=1), sd, p + ".bn2", training)
    if (p + ".se.fc.0.weight") in sd:
        out = _se(out, sd, p + ".se")
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1", training)
    return F.relu(out + x)


def blocks_of(sd) -> tuple:
    """Bottlenecks per stage, read off the state_dict keys ((3, 4, 6, 3) for resnet50, (3, 4, 23, 3) resnet101, (3, 8, 36, 3) resnet152:
    nets/pose_resnet_dconv.py:306-339)."""
    return tuple(1 + max(int(k.split(".")[1]) for k in sd if k.startswith(f"layer{li}.")) for li in (1, 2, 3, 4))


def resnet_trunk(sd: Dict[str, torch.Tensor], x: torch.Tensor, training=False,
                 tap: Optional[Callable[[str, torch.Tensor], None]] = None, blocks=None):
    blocks = blocks or blocks_of(sd)
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1", training))
    x = F.max_pool2d(x, 3, 2, 1)
    if tap:
        tap("maxpool", x)
    block = _bottleneck if "layer1.0.conv3.weight" in sd else _res_basic_block
    for li, n in enumerate(blocks, start=1):
        for bi in range(n):
            x = block(x, sd, f"layer{li}.{bi}", 2 if (bi == 0 and li > 1) else 1, training)
        if tap:
            tap(f"layer{li}", x)
    return x


def resnet_dconv_forward(sd, x, training=False, tap=None):
    x = resnet_trunk(sd, x, training, tap)
    for n, idx in enumerate((0, 3, 6)):  # Sequential indices: deconv, bn, relu triples
        x = F.conv_transpose2d(x, sd[f"deconv_layers.{idx}.weight"], stride=2, padding=1, output_padding=0)
        x = F.relu(_bn(x, sd, f"deconv_layers.{idx + 1}", training))
        if tap:
            tap(f"deconv{n}", x)
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"])


def resnet_duc_forward(sd, x, training=False, tap=None):
    x = resnet_trunk(sd, x, training, tap)
    x = F.pixel_shuffle(x, 2)
    for n, idx in enumerate((1, 2)):
        x = F.conv2d(x, sd[f"duc_layers.{idx}.conv.weight"], padding=1)
        x = F.pixel_shuffle(F.relu(_bn(x, sd, f"duc_layers.{idx}.bn", training)), 2)
        if tap:
            tap(f"duc{n}", x)
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"], padding=1)


FORWARDS = {"resnet50_dconv": resnet_dconv_forward, "resnet50_duc": resnet_duc_forward}


def state_dict_shapes_resnet50(head: str, num_joints: int = 17, se: bool = False):
    """(key, shape, dtype) list of the reference's resnet50 state_dict without
    instantiating anything: lets the GPU box regenerate the synthetic weights with no reference."""
    out = []

    def conv(k, o, i, kh, kw):
        out.append((k + ".weight", (o, i, kh, kw), "torch.float32"))

    def bn(k, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{k}.{leaf}", (c,), "torch.float32"))
        out.append((f"{k}.num_batches_tracked", (), "torch.int64"))

    conv("conv1", 64, 3, 7, 7)
    bn("bn1", 64)
    inpl = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), RESNET50_BLOCKS), start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            conv(p + ".conv1", planes, inpl, 1, 1); bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3, 3); bn(p + ".bn2", planes)
            conv(p + ".conv3", planes * 4, planes, 1, 1); bn(p + ".bn3", planes * 4)
            if bi == 0:
                conv(p + ".downsample.0", planes * 4, inpl, 1, 1); bn(p + ".downsample.1", planes * 4)
                if se:
                    for f in ("0", "2"):
                        out.append((f"{p}.se.fc.{f}.weight", (planes * 4, planes * 4, 1, 1), "torch.float32"))
                        out.append((f"{p}.se.fc.{f}.bias", (planes * 4,), "torch.float32"))
            inpl = planes * 4
    if head == "dconv":
        cin = 2048
        for idx in (0, 3, 6):
            out.append((f"deconv_layers.{idx}.weight", (cin, 256, 4, 4), "torch.float32"))
            bn(f"deconv_layers.{idx + 1}", 256)
            cin = 256
        conv("final_layer", num_joints, 256, 1, 1)
    elif head == "duc":
        conv("duc_layers.1.conv", 1024, 512, 3, 3); bn("duc_layers.1.bn", 1024)
        conv("duc_layers.2.conv", 512, 256, 3, 3); bn("duc_layers.2.bn", 512)
        conv("final_layer", num_joints, 128, 3, 3)
    else:
        raise ValueError(head)
    out.append(("final_layer.bias", (num_joints,), "torch.float32"))
    return out


# ------------------------------------------------------------------------------------------------------------------
# HRNet: nets/pose_hrnet.py:419-454 (PoseHighResolutionNet.forward), :241-259 (HighResolutionModule.forward),
# :34-51 (BasicBlock), :71-92 (Bottleneck), :181-236 (fuse layers), :327-366 (transitions)
# ------------------------------------------------------------------------------------------------------------------
def _conv_bn(x, sd, conv_key, bn_key, stride=1, padding=0, relu=False, training=False):
    y = _bn(F.conv2d(x, sd[conv_key + ".weight"], stride=stride, padding=padding), sd, bn_key, training)
    return F.relu(y) if relu else y


def _hr_basic_block(x, sd, p, training=False):
    out = _conv_bn(x, sd, p + ".conv1", p + ".bn1", padding=1, relu=True, training=training)
    out = _conv_bn(out, sd, p + ".conv2", p + ".bn2", padding=1, training=training)
    return F.relu(out + x)


def hrnet_forward(sd, x, cfg, training=False, tap=None):
    extra = cfg["MODEL"]["EXTRA"]
    x = _conv_bn(x, sd, "conv1", "bn1", stride=2, padding=1, relu=True, training=training)
    x = _conv_bn(x, sd, "conv2", "bn2", stride=2, padding=1, relu=True, training=training)
    for k in range(4):
        x = _bottleneck(x, sd, f"layer1.{k}", 1, training)
    if tap:
        tap("layer1", x)
    ys, pre_n = [x], 1
    for si, st in enumerate((2, 3, 4)):
        sc = extra[f"STAGE{st}"]
        nb = sc["NUM_BRANCHES"]
        t = f"transition{si + 1}"
        xs = []
        for i in range(nb):
            if i < pre_n:
                if f"{t}.{i}.0.weight" in sd:
                    xs.append(_conv_bn(ys[i], sd, f"{t}.{i}.0", f"{t}.{i}.1", padding=1, relu=True, training=training))
                else:
                    xs.append(ys[i])
            else:
                v = ys[-1]
                for j in range(i + 1 - pre_n):
                    v = _conv_bn(v, sd, f"{t}.{i}.{j}.0", f"{t}.{i}.{j}.1", stride=2, padding=1, relu=True, training=training)
                xs.append(v)
        for m in range(sc["NUM_MODULES"]):
            multi = not (st == 4 and m == sc["NUM_MODULES"] - 1)
            base = f"stage{st}.{m}"
            for i in range(nb):
                for k in range(sc["NUM_BLOCKS"][i]):
                    xs[i] = _hr_basic_block(xs[i], sd, f"{base}.branches.{i}.{k}", training)
            fused = []
            for i in range(nb if multi else 1):
                y = None
                for j in range(nb):
                    f = f"{base}.fuse_layers.{i}.{j}"
                    if j == i:
                        term = xs[i]
                    elif j > i:
                        term = _conv_bn(xs[j], sd, f + ".0", f + ".1", training=training)
                        term = F.interpolate(term, scale_factor=2 ** (j - i), mode="nearest")
                    else:
                        term = xs[j]
                        for k in range(i - j):
                            term = _conv_bn(term, sd, f"{f}.{k}.0", f"{f}.{k}.1", stride=2, padding=1,
                                            relu=(k != i - j - 1), training=training)
                    y = term if y is None else y + term
                fused.append(F.relu(y))
            xs = fused
        ys, pre_n = xs, nb
        if tap:
            tap(f"stage{st}", ys[0])
    kf = extra["FINAL_CONV_KERNEL"]
    return F.conv2d(ys[0], sd["final_layer.weight"], sd["final_layer.bias"], padding=1 if kf == 3 else 0)
