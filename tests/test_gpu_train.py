"""GPU parity of the training step (ResNet50-DConv, fp32) against the oracle and the reference golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets_oracle, pose_oracle, train_oracle  # noqa: E402
from simple_pose_amd import synth  # noqa: E402
from simple_pose_amd.nets import pose_resnet_dconv  # noqa: E402
from simple_pose_amd.train import PoseTrainer  # noqa: E402

DEV = "cuda:0"


def _model(seed, head="dconv"):
    from simple_pose_amd.nets import pose_resnet_duc
    m = (pose_resnet_dconv if head == "dconv" else pose_resnet_duc).resnet50(pretrained=False, num_classes=17)
    sd = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50(head), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(DEV).train(), {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def _batch(B, H, W, seed):
    x = synth.input_images(B, seed, h=H, w=W)
    joints = synth.joints_batch(B, 17, seed=seed + 40, w=W // 4, h=H // 4)
    t, w = pose_oracle.encode_refine(joints, 2.0, (W // 4, H // 4))
    return x, t, w


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.sqrt((b * b).mean()) + 1e-30))


@pytest.mark.parametrize("B,H,W,head", [(2, 64, 64, "dconv"), (3, 96, 64, "dconv"), (2, 64, 64, "duc"), (3, 96, 64, "duc")])
def test_train_step_vs_oracle_small(B, H, W, head):
    model, sd = _model(7, head)
    arch = "resnet50_" + head
    x, t, w = _batch(B, H, W, 7)
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3)
    loss = tr.forward_backward(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    oloss, ograds, oheat = train_oracle.forward_backward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w), arch=arch)
    assert _rel(tr.last_heat.cpu().numpy(), oheat.numpy()) < 1e-3
    assert abs(loss.item() - float(oloss)) <= 1e-4 * abs(float(oloss))
    named = dict(model.named_parameters())
    # Gradients of this deep BN net at a tiny batch are chaotic at the 0.3 % level: torch-fp32 differs from torch-fp64 (and
    # from itself at another thread count) by 3-4e-3 in relative L2.  The bar is therefore relative L2 against the fp64
    # oracle, a few times torch-fp32's own deviation; the better-conditioned first case also meets a max-abs bar.
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in _model(7, head)[1].items()}
    _, g64, _ = train_oracle.forward_backward(sd64, torch.from_numpy(x).double(), torch.from_numpy(t).double(), torch.from_numpy(w).double(),
                                              arch=arch)
    l2 = sorted(((float((named[k].grad.cpu().double() - g64[k]).norm() / (g64[k].norm() + 1e-30)), k) for k in g64), reverse=True)
    l2_torch = sorted((float((ograds[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64), reverse=True)
    assert l2[0][0] < max(3e-2, 8 * l2_torch[0]), (l2[:6], l2_torch[:3])
    assert np.median([e for e, _ in l2]) < max(1e-2, 4 * np.median(l2_torch)), (np.median([e for e, _ in l2]), np.median(l2_torch))
    bufs = dict(model.named_buffers())
    for k in ("bn1.running_mean", "bn1.running_var", "layer2.0.downsample.1.running_var",
              "deconv_layers.7.running_mean" if head == "dconv" else "duc_layers.2.bn.running_var"):
        assert _rel(bufs[k].cpu().numpy(), sd[k].numpy()) < 1e-4, k
    # Adam
    tr.optimizer_step(1.0)
    params = {k: sd[k] for k in ograds}
    train_oracle.adam_step(params, ograds, {}, lr=1e-3)
    torch.cuda.synchronize()
    bad = [(float((named[k].detach().cpu() - params[k]).abs().max()), k) for k in params]
    # first Adam step moves every weight by ~lr * sign(g); where |g| ~ 0 the sign is numerically undefined
    frac_close = np.mean([float(((named[k].detach().cpu() - params[k]).abs() < 2e-4).float().mean()) for k in params])
    assert frac_close > 0.995, (frac_close, sorted(bad, reverse=True)[:5])


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 64)])
def test_train_step_with_selayer_vs_oracle(B, H, W):
    """`resnet50(reduction=True)` (SELayer in the first Bottleneck of every stage, nets/commons.py:4-18, pose_resnet_dconv.py:108-110,126-127;
    enabled by configs/dp_fast_pose.yaml) in train mode: loss, heat maps, every gradient incl. the gate's FC weights and biases, and
    one Adam step against the float64 oracle - same bars as the plain net (a tiny-batch BatchNorm net is chaotic at the 0.3 % level)."""
    m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
    sdn = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv", se=True), 7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
    model = m.to(DEV).train()
    sd = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    x, t, w = _batch(B, H, W, 7)
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3)
    assert sum(".se.fc." in n for n in tr.layers) == 8
    loss = tr.forward_backward(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    oloss, ograds, oheat = train_oracle.forward_backward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w), arch="resnet50_dconv")
    assert _rel(tr.last_heat.cpu().numpy(), oheat.numpy()) < 1e-3
    assert abs(loss.item() - float(oloss)) <= 1e-4 * abs(float(oloss))
    named = dict(model.named_parameters())
    sd64 = {k: (torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy())) for k, v in sdn.items()}
    _, g64, _ = train_oracle.forward_backward(sd64, torch.from_numpy(x).double(), torch.from_numpy(t).double(), torch.from_numpy(w).double(),
                                              arch="resnet50_dconv")
    assert set(g64) == set(named)
    l2 = sorted(((float((named[k].grad.cpu().double() - g64[k]).norm() / (g64[k].norm() + 1e-30)), k) for k in g64), reverse=True)
    l2_torch = sorted((float((ograds[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64), reverse=True)
    assert l2[0][0] < max(3e-2, 8 * l2_torch[0]), (l2[:6], l2_torch[:3])
    assert np.median([e for e, _ in l2]) < max(1e-2, 4 * np.median(l2_torch)), (np.median([e for e, _ in l2]), np.median(l2_torch))
    se = [(e, k) for e, k in l2 if ".se." in k]
    assert len(se) == 16 and max(e for e, _ in se) < max(3e-2, 8 * l2_torch[0]), se[:4]
    # the streamed step (weight gradients / optimizer on their own streams) on the same net: bit-reproducible, loss goes down
    losses = [tr.step(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV)).item() for _ in range(3)]
    assert losses[-1] < losses[0]


BASIC_NETS = [("resnet18", "dconv", False), ("resnet34", "duc", False), ("resnet18", "dconv", True)]


@pytest.mark.parametrize("arch,head,se", BASIC_NETS, ids=[f"{a}_{h}" + ("_se" if s else "") for a, h, s in BASIC_NETS])
def test_basic_block_nets_train_step_vs_oracle(arch, head, se):
    """The BasicBlock factories (resnet18 / resnet34, pose_resnet_dconv.py:38-80,282-304; heads from 512 channels; SELayer on the blocks with a
    projection shortcut) in train mode through the same tape as the Bottleneck nets (round 5: `tape.build_resnet_basic`): loss, heat maps,
    every gradient against the float64 oracle at the bars of the ResNet-50 tests, the streamed step bit-reproducible with a falling loss,
    and the bf16 step (bf16 activation gradients) running and learning."""
    from simple_pose_amd.nets import pose_resnet_duc
    B, H, W = 3, 96, 64
    mod = pose_resnet_dconv if head == "dconv" else pose_resnet_duc

    def make():
        m = getattr(mod, arch)(pretrained=False, num_classes=17, reduction=se)
        layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
        sdn = synth.conditioned_state_dict(layout, 11)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
        return m.to(DEV).train(), sdn
    model, sdn = make()
    sd = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    x, t, w = _batch(B, H, W, 11)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3)
    assert not any(n.endswith(".conv3") for n in tr.layers)                       # BasicBlocks: two 3x3 convs per block
    loss = tr.forward_backward(xs, ts, ws)
    torch.cuda.synchronize()
    okey = "resnet50_" + head                                                     # (the oracle's trunk reads the block type off the keys)
    oloss, ograds, oheat = train_oracle.forward_backward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w), arch=okey)
    assert _rel(tr.last_heat.cpu().numpy(), oheat.numpy()) < 1e-3
    assert abs(loss.item() - float(oloss)) <= 1e-4 * abs(float(oloss))
    named = dict(model.named_parameters())
    sd64 = {k: (torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy())) for k, v in sdn.items()}
    _, g64, _ = train_oracle.forward_backward(sd64, torch.from_numpy(x).double(), torch.from_numpy(t).double(), torch.from_numpy(w).double(), arch=okey)
    assert set(g64) == set(named)
    l2 = sorted(((float((named[k].grad.cpu().double() - g64[k]).norm() / (g64[k].norm() + 1e-30)), k) for k in g64), reverse=True)
    l2_torch = sorted((float((ograds[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64), reverse=True)
    assert l2[0][0] < max(3e-2, 8 * l2_torch[0]), (l2[:6], l2_torch[:3])
    assert np.median([e for e, _ in l2]) < max(1e-2, 4 * np.median(l2_torch)), (np.median([e for e, _ in l2]), np.median(l2_torch))
    runs = []
    for _ in range(2):
        m2, _ = make()
        tr2 = PoseTrainer(m2, in_h=H, in_w=W, lr=1e-3)
        runs.append([tr2.step(xs, ts, ws).item() for _ in range(4)])
    assert runs[0] == runs[1] and runs[0][-1] < runs[0][0]                        # bit-reproducible, learning
    m3, _ = make()
    tr3 = PoseTrainer(m3, in_h=H, in_w=W, lr=1e-3, dtype="bf16")
    assert tr3.g16 == (not se)                                                    # bf16 activation gradients on the plain nets
    lb = [tr3.step(xs, ts, ws).item() for _ in range(4)]
    assert abs(lb[0] - runs[0][0]) <= 2e-2 * abs(runs[0][0]) and lb[-1] < lb[0]


GROUPED_NETS = [("resnext50_32x4d", "dconv"), ("resnext50_32x4d", "duc"), ("resnext101_32x8d", "dconv")]


@pytest.mark.parametrize("arch,head", GROUPED_NETS, ids=[f"{a}_{h}" for a, h in GROUPED_NETS])
def test_grouped_nets_train_step_vs_oracle(arch, head):
    """The grouped factories (resnext50_32x4d / resnext101_32x8d, pose_resnet_dconv.py:97-101,342-368) in train mode (round 6): conv2 of every
    Bottleneck is nn.Conv2d(groups=32) - forward and input gradient as grouped implicit-GEMM launches on block-diagonal panels (the stride-2
    blocks: one launch per output phase), the weight gradient by sp_conv2d_wgrad_grouped.  Loss, heat maps and EVERY gradient against the
    float64 oracle at the bars of the ResNet-50 tests; the streamed step bit-reproducible with a falling loss; the bf16 step runs and learns."""
    from simple_pose_amd.nets import pose_resnet_duc
    B, H, W = 2, 64, 64
    mod = pose_resnet_dconv if head == "dconv" else pose_resnet_duc

    def make():
        m = getattr(mod, arch)(pretrained=False, num_classes=17)
        layout = [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()]
        sdn = synth.conditioned_state_dict(layout, 13)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
        return m.to(DEV).train(), sdn
    model, sdn = make()
    sd = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    x, t, w = _batch(B, H, W, 13)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3)
    grouped = [L for L in tr.layers.values() if L.groups > 1]
    assert len(grouped) == sum(model.BLOCKS) and all(L.groups == 32 and L.d_fwd.c_in_group == 64 for L in grouped)
    assert sum(len(L.d_dgrad) == 4 for L in grouped) == 3                           # the three stride-2 blocks: four output phases each
    loss = tr.forward_backward(xs, ts, ws)
    torch.cuda.synchronize()
    okey = "resnet50_" + head                                                     # (the oracle reads blocks and groups off the keys / shapes)
    oloss, ograds, oheat = train_oracle.forward_backward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w), arch=okey)
    assert _rel(tr.last_heat.cpu().numpy(), oheat.numpy()) < 1e-3
    assert abs(loss.item() - float(oloss)) <= 1e-4 * abs(float(oloss))
    named = dict(model.named_parameters())
    sd64 = {k: (torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy())) for k, v in sdn.items()}
    _, g64, _ = train_oracle.forward_backward(sd64, torch.from_numpy(x).double(), torch.from_numpy(t).double(), torch.from_numpy(w).double(), arch=okey)
    assert set(g64) == set(named)
    l2 = sorted(((float((named[k].grad.cpu().double() - g64[k]).norm() / (g64[k].norm() + 1e-30)), k) for k in g64), reverse=True)
    l2_torch = sorted((float((ograds[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64), reverse=True)
    assert l2[0][0] < max(3e-2, 8 * l2_torch[0]), (l2[:6], l2_torch[:3])
    assert np.median([e for e, _ in l2]) < max(1e-2, 4 * np.median(l2_torch)), (np.median([e for e, _ in l2]), np.median(l2_torch))
    gl2 = [(e, k) for e, k in l2 if k.endswith(".conv2.weight")]                  # the grouped weights themselves
    assert len(gl2) == len(grouped) and max(e for e, _ in gl2) < max(3e-2, 8 * l2_torch[0]), gl2[:4]
    if arch != "resnext50_32x4d" or head != "dconv":
        return                                                                    # (the step-level checks once: they do not depend on the head / depth)
    runs = []
    for _ in range(2):
        m2, _ = make()
        tr2 = PoseTrainer(m2, in_h=H, in_w=W, lr=1e-3)
        runs.append([tr2.step(xs, ts, ws).item() for _ in range(4)])
    assert runs[0] == runs[1] and runs[0][-1] < runs[0][0]                        # bit-reproducible (fixed-order folds, no atomics), learning
    m3, _ = make()
    tr3 = PoseTrainer(m3, in_h=H, in_w=W, lr=1e-3, dtype="bf16")
    assert tr3.g16
    lb = [tr3.step(xs, ts, ws).item() for _ in range(4)]
    assert abs(lb[0] - runs[0][0]) <= 2e-2 * abs(runs[0][0]) and lb[-1] < lb[0]


def _hrnet(seed):
    import functools
    import os
    from simple_pose_amd.nets.pose_hrnet import get_pose_net, hrnet_state_dict_shapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = get_pose_net(os.path.join(root, "simple_pose_amd", "nets", "hrnet_w32.yaml"), pretrained=None, joint_num=17)
    sdn = synth.conditioned_state_dict(hrnet_state_dict_shapes(net.cfg, 17), seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
    nets_oracle.FORWARDS["hrnet_w32"] = lambda sd, x, training=False: nets_oracle.hrnet_forward(sd, x, net.cfg, training=training)
    return net.to(DEV).train(), sdn


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 96, 64)])
def test_hrnet_train_step_vs_oracle(B, H, W):
    """PoseHighResolutionNet (nets/pose_hrnet.py:419-454) in train mode through PoseTrainer: stem, layer1 Bottlenecks, transitions, the
    BasicBlock branches of every HighResolutionModule, the fuse layers (1x1 conv + BN + nearest upsample + add; chains of stride-2 3x3
    convs; ReLU after the last term) and the final conv - loss, heat maps and all 1,148 parameter gradients against the float64 oracle
    (torch autograd through oracle/nets_oracle.hrnet_forward), bars as for the ResNets."""
    model, sdn = _hrnet(5)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sdn.items()}
    x, t, w = _batch(B, H, W, 7)
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3)
    loss = tr.forward_backward(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    oloss, ograds, oheat = train_oracle.forward_backward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w), arch="hrnet_w32")
    assert _rel(tr.last_heat.cpu().numpy(), oheat.numpy()) < 1e-3
    assert abs(loss.item() - float(oloss)) <= 1e-4 * abs(float(oloss))
    named = dict(model.named_parameters())
    sd64 = {k: (torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy())) for k, v in sdn.items()}
    _, g64, _ = train_oracle.forward_backward(sd64, torch.from_numpy(x).double(), torch.from_numpy(t).double(), torch.from_numpy(w).double(),
                                              arch="hrnet_w32")
    assert set(g64) == set(named)
    l2 = sorted(((float((named[k].grad.cpu().double() - g64[k]).norm() / (g64[k].norm() + 1e-30)), k) for k in g64), reverse=True)
    l2_torch = sorted((float((ograds[k].double() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64), reverse=True)
    assert l2[0][0] < max(3e-2, 8 * l2_torch[0]), (l2[:6], l2_torch[:3])
    assert np.median([e for e, _ in l2]) < max(1e-2, 4 * np.median(l2_torch)), (np.median([e for e, _ in l2]), np.median(l2_torch))
    bufs = dict(model.named_buffers())
    for k in ("bn1.running_mean", "stage3.1.branches.2.0.bn1.running_var", "stage4.2.fuse_layers.0.3.1.running_mean"):
        assert _rel(bufs[k].cpu().numpy(), sd[k].numpy()) < 1e-4, k


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hrnet_train_step_is_bit_reproducible_and_learns(dtype):
    """The streamed HRNet step (weight gradients and optimizer on their own streams) twice from the same state: identical parameters;
    the loss goes down; also through the autograd surface `model(x)` in train() mode."""
    res = []
    for _ in range(2):
        model, _ = _hrnet(6)
        tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3, dtype=dtype)
        x, t, w = _batch(4, 128, 96, 9)
        xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
        losses = [tr.step(xs, ts, ws).item() for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, tr.flat.data.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    assert res[0][0][-1] < res[0][0][0]
    if dtype == "fp32":
        model, _ = _hrnet(6)
        out = model(xs)
        assert out.requires_grad and out.shape == (4, 17, 32, 24)
        out.square().mean().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_selayer_train_step_is_bit_reproducible_and_bf16_tracks_fp32(dtype):
    """The SELayer step twice from the same state: identical parameters (deterministic kernels); bf16 compute stays within the bf16 bar of
    the plain net's loss."""
    res = []
    for _ in range(2):
        m = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17, reduction=True)
        sdn = synth.conditioned_state_dict(nets_oracle.state_dict_shapes_resnet50("dconv", se=True), 9)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
        tr = PoseTrainer(m.to(DEV).train(), in_h=128, in_w=96, lr=1e-3, dtype=dtype)
        x, t, w = _batch(4, 128, 96, 9)
        xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
        losses = [tr.step(xs, ts, ws).item() for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, tr.flat.data.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    assert res[0][0][-1] < res[0][0][0]


# pinned from gpurun_out/measured_parity.json, round 2: worst gradient slice 1.5e-2 of the per-element gradient scale (conv1.weight: the
# longest fp32 chain of the net, where the reference's own oneDNN-vs-fp64 spread is 0.3-1 %); 99.86 % of the sliced parameters within
# 2e-4 after one Adam step (lr 1e-3: a sign flip of a near-zero gradient moves a parameter by 2e-3); loss 7.8e-8 relative
# Round 4: the conv epilogues now add a tile's wave rows before the partial row is stored (another summation order of the BatchNorm sums):
# conv1.weight moved to 2.7e-2.  tests/measure_reference_spread.py: the reference's own fp32 step moves by 1.8e-2 ... 2.1e-2 on this metric
# when only its thread count changes (and sits 1.2e-2 ... 2.0e-2 from fp64) - this B = 2 BatchNorm net is chaotic at that level, so the bar
# is 2x the reference's own spread; the per-kernel tests (test_gpu_backward_kernels.py: every sum against float64 at 1e-6) are the sharp ones
# Round 5 (advisor): the wider bar is kept ONLY for the slices that moved past 2e-2 - the three at the far end of the backward chain (measured
# round 5: conv1.weight 2.81e-2, layer1.0.conv1.weight 2.44e-2, bn1.weight 2.02e-2; the next ones: layer2.0.conv2.weight 1.96e-2,
# layer2.0.downsample.0.weight 1.90e-2, layer4 / deconv0 1.5e-2, the head 1e-6); every other slice is back at 2e-2, so a regression of that size
# elsewhere fails again.  (The step is bit-reproducible for a given tile table, so these are not box-to-box noise.)
GRAD_SLICE_BAR = 2e-2
GRAD_SLICE_BAR_OF = {"conv1.weight": 4e-2, "layer1.0.conv1.weight": 4e-2, "bn1.weight": 4e-2}
ADAM_CLOSE_BAR = 0.995


def test_train_step_vs_reference_golden(golden, measured):
    """One step at 256x192, B=2, against the real reference (tests/golden/g6_train_step.npz)."""
    g = golden("g6_train_step.npz")
    model, _ = _model(0)
    x = synth.input_images(2, 0)
    t, w = pose_oracle.encode_refine(g["joints"], 2.0, (48, 64))
    tr = PoseTrainer(model, lr=1e-3)
    loss = tr.step(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    assert _rel(tr.last_heat.cpu().numpy(), g["heat_train"]) < 1e-3
    named = dict(model.named_parameters())
    over = []
    for key in [k for k in g.files if k.startswith("grad/")]:
        k = key[5:]
        ref = g[key]
        got = named[k].grad.cpu().numpy()[tuple(slice(0, s) for s in ref.shape)]
        scale = float(g["gradnorm/" + k]) / np.sqrt(named[k].numel())
        bar = GRAD_SLICE_BAR_OF.get(k, GRAD_SLICE_BAR)
        measured(f"grad_slice_err_over_scale/{k}", np.abs(got - ref).max() / scale, bar)
        if np.abs(got - ref).max() > bar * scale + 1e-12:
            over.append((k, float(np.abs(got - ref).max() / scale), bar))
    assert not over, over            # (every slice is measured before the verdict: one run names all of them)
    bufs = dict(model.named_buffers())
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_var", "deconv_layers.7.running_mean"):
        assert np.abs(bufs[k].cpu().numpy() - g["buf/" + k]).max() <= 1e-4 * max(1.0, np.abs(g["buf/" + k]).max()), k
    assert int(bufs["bn1.num_batches_tracked"]) == 1
    close = []
    for key in [k for k in g.files if k.startswith("param/")]:
        k = key[6:]
        ref = g[key]
        got = named[k].detach().cpu().numpy()[tuple(slice(0, s) for s in ref.shape)]
        close.append((np.abs(got - ref) < 2e-4).mean())
        measured(f"adam_param_max_abs_diff/{k}", np.abs(got - ref).max())
    measured("adam_fraction_within_2e-4", np.mean(close), ADAM_CLOSE_BAR)
    measured("loss_rel_err", abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), 1e-4)
    assert np.mean(close) > ADAM_CLOSE_BAR, close


# G6b (round 6): the bars the reference's OWN arithmetic allows at B = 8 (its 1-thread / 8-thread / fp64 evaluations of this very step, /tmp-free:
# `python tests/measure_reference_spread.py 8`): gradient norm of any parameter <= 1.1e-3 (median 5e-5), a sketch <= 8.4e-3 of the norm, relative
# L2 of a whole gradient <= 6.7e-3, the max over a slice 2-6e-2 of the rms.  A 1 % error in a dgrad epilogue moves every upstream norm by 1 %.
# Measured on the MI355X (round 6, `measured`): norm 1.17e-3, sketch 8.8e-3, slices <= 1.95e-2 (conv1.weight): every bar <= 3x its measurement.
G6B_NORM_BAR = 3.5e-3
G6B_SKETCH_BAR = 2.5e-2
G6B_SLICE_BAR = 5e-2


def test_train_step_vs_reference_golden_batch_8(golden, measured):
    """One fp32 step at 256x192, B = 8, against the real reference (tests/golden/g6b_train_step_b8.npz): loss, the fp64 norm and two
    sketches of the gradient of EVERY one of the 170 parameters (quantities the reference itself holds to 1e-3 across summation orders -
    the round-5 verdict's better-conditioned pin of the whole-net backward), G6's slices, running statistics, parameters after Adam."""
    from oracle.train_oracle import gradient_sketch_vector
    g = golden("g6b_train_step_b8.npz")
    B = int(g["batch"])
    model, _ = _model(int(g["seed"]))
    x = synth.input_images(B, int(g["seed"]))
    t, w = pose_oracle.encode_refine(g["joints"], 2.0, (48, 64))
    tr = PoseTrainer(model, lr=1e-3)
    loss = tr.step(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    measured("g6b/loss_rel_err", abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])), 1e-4)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    assert _rel(tr.last_heat.cpu().numpy()[:, :, ::4, ::4], g["heat_train_sub"]) < 1e-3
    named = dict(model.named_parameters())
    keys = [str(k) for k in g["keys"]]
    assert keys == list(named)
    worst_n, worst_s, over = (0.0, ""), (0.0, ""), []
    for i, k in enumerate(keys):
        gd = named[k].grad.double().cpu()
        n = float(g["gradnorm"][i])
        e = abs(float(gd.norm()) - n) / n
        worst_n = max(worst_n, (e, k))
        if e > G6B_NORM_BAR:
            over.append(("norm", k, e))
        for j in range(2):
            sk = float((gd.numpy() * gradient_sketch_vector(k, j, gd.shape)).sum())
            es = abs(sk - float(g["sketch"][i, j])) / n
            worst_s = max(worst_s, (es, k))
            if es > G6B_SKETCH_BAR:
                over.append(("sketch", k, es))
    measured("g6b/gradnorm_rel_err_max_over_170_params", worst_n[0], G6B_NORM_BAR)
    measured("g6b/sketch_err_over_norm_max_over_170_params", worst_s[0], G6B_SKETCH_BAR)
    assert not over, over[:10]
    for key in [k for k in g.files if k.startswith("grad/")]:
        k = key[5:]
        ref = g[key]
        got = named[k].grad.cpu().numpy()[tuple(slice(0, s) for s in ref.shape)]
        scale = float(g["gradnorm"][keys.index(k)]) / np.sqrt(named[k].numel())
        measured(f"g6b/grad_slice_err_over_scale/{k}", np.abs(got - ref).max() / scale, G6B_SLICE_BAR)
        assert np.abs(got - ref).max() <= G6B_SLICE_BAR * scale + 1e-12, k
    bufs = dict(model.named_buffers())
    for k in ("bn1.running_mean", "bn1.running_var", "layer3.5.bn3.running_var", "deconv_layers.7.running_mean"):
        assert np.abs(bufs[k].cpu().numpy() - g["buf/" + k]).max() <= 1e-4 * max(1.0, np.abs(g["buf/" + k]).max()), k
    close = []
    for key in [k for k in g.files if k.startswith("param/")]:
        k = key[6:]
        ref = g[key]
        got = named[k].detach().cpu().numpy()[tuple(slice(0, s) for s in ref.shape)]
        close.append((np.abs(got - ref) < 2e-4).mean())
    measured("g6b/adam_fraction_within_2e-4", np.mean(close), ADAM_CLOSE_BAR)
    assert np.mean(close) > ADAM_CLOSE_BAR, close


def test_training_is_deterministic_and_decreases_loss():
    model, _ = _model(3)
    x, t, w = _batch(4, 128, 96, 3)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3)
    losses = [tr.step(xs, ts, ws).item() for _ in range(6)]
    model2, _ = _model(3)
    tr2 = PoseTrainer(model2, in_h=128, in_w=96, lr=1e-3)
    losses2 = [tr2.step(xs, ts, ws).item() for _ in range(6)]
    assert losses == losses2                      # bitwise reproducible: no atomics anywhere in the step
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("grad_dtype,head", [("bf16", "dconv"), ("fp32", "dconv"), ("bf16", "duc")])
def test_bf16_train_step_behaves_like_the_amp_oracle(grad_dtype, head):
    """dtype="bf16" (BASELINE config 4's compute type), with the activation gradients kept in bf16 (the default: what autocast does) and in fp32.  bf16 perturbs the forward by ~0.4 %, and in this BN-heavy net with
    synthetic weights the gradient is extremely sensitive to that (torch's own CPU autocast-bf16 step differs from its fp32
    step by 0.5 % at the head up to ~60 % at the stem in relative L2).  So the checks are: loss within 2 % of fp32; the head
    gradients (not yet amplified) close to fp32; and layer by layer a deviation from fp32 no worse than 1.5x the deviation of
    the reference-style AMP oracle (oracle/train_oracle.py, amp_bf16=True)."""
    B, H, W = 4, 128, 96
    model, sd = _model(9, head)
    x, t, w = _batch(B, H, W, 9)
    tr = PoseTrainer(model, in_h=H, in_w=W, lr=1e-3, dtype="bf16", grad_dtype=grad_dtype)
    assert tr.g16 == (grad_dtype == "bf16")
    loss = tr.forward_backward(torch.from_numpy(x).to(DEV), torch.from_numpy(t).to(DEV), torch.from_numpy(w).to(DEV))
    torch.cuda.synchronize()
    xs, ts, ws = torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w)
    arch = "resnet50_" + head
    oloss, g32, _ = train_oracle.forward_backward({k: v.clone() for k, v in sd.items()}, xs, ts, ws, arch=arch)
    aloss, gamp, _ = train_oracle.forward_backward({k: v.clone() for k, v in sd.items()}, xs, ts, ws, arch=arch, amp_bf16=True)
    assert abs(loss.item() - float(oloss)) <= 2e-2 * abs(float(oloss)), (loss.item(), float(oloss))
    named = dict(model.named_parameters())

    def dev(g, k):
        return float((g.double() - g32[k].double()).norm() / (g32[k].double().norm() + 1e-30))

    mine = {k: dev(named[k].grad.cpu(), k) for k in g32}
    amp = {k: dev(gamp[k], k) for k in g32}
    last_bn = "deconv_layers.7.weight" if head == "dconv" else "duc_layers.2.bn.weight"
    assert mine["final_layer.weight"] < 2e-2 and mine["final_layer.bias"] < 2e-2 and mine[last_bn] < 3e-2, mine
    worse = [(k, mine[k], amp[k]) for k in g32 if mine[k] > 1.5 * amp[k] + 0.02]
    assert len(worse) <= 0.05 * len(g32), worse[:8]
    assert np.median(list(mine.values())) <= 1.25 * np.median(list(amp.values())) + 0.01
    assert all(p.dtype == torch.float32 and p.grad.dtype == torch.float32 for p in model.parameters())   # fp32 master state


def test_bf16_training_is_deterministic_and_decreases_loss():
    model, _ = _model(3)
    x, t, w = _batch(4, 128, 96, 3)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3, dtype="bf16")
    losses = [tr.step(xs, ts, ws).item() for _ in range(8)]
    model2, _ = _model(3)
    tr2 = PoseTrainer(model2, in_h=128, in_w=96, lr=1e-3, dtype="bf16")
    losses2 = [tr2.step(xs, ts, ws).item() for _ in range(8)]
    assert losses == losses2
    assert losses[-1] < losses[0]


# ---- N > 1: two ranks sharing the one GPU of the box (gloo carries the collectives) ---------------------------------------
def _ddp_worker(rank, world, port, sync_bn, bucket_mb, out_dir, fused=False, head="dconv"):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simple_pose_amd.sharding import rank_indices
        model, _ = _model(3 + rank, head)            # deliberately different initial weights per rank: the ctor broadcasts rank 0's
        x, t, w = _batch(4, 64, 64, 11)
        idx = rank_indices(4, rank, world)
        xs, ts, ws = (torch.from_numpy(v[idx]).to(DEV) for v in (x, t, w))
        tr = PoseTrainer(model, in_h=64, in_w=64, lr=1e-3, sync_bn=sync_bn, bucket_mb=bucket_mb)
        if fused:                                    # step(): all-reduce + Adam + re-pack per bucket on the optimizer stream
            loss = tr.step(xs, ts, ws).item()
            n_launched, scale = len(tr.buckets), 1.0 / world
            grad = (tr.flat.grad * scale).cpu().numpy()
        else:
            loss = tr.forward_backward(xs, ts, ws).item()
            n_launched = sum(wk is not None for wk in tr._works)
            scale = tr.all_reduce_grads()
            grad = (tr.flat.grad * scale).cpu().numpy()
            tr.optimizer_step(scale)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=loss, grad=grad, param=tr.flat.data.cpu().numpy(),
                 rm=tr.buffers["layer4.2.bn3.running_mean"].cpu().numpy(), rv=tr.buffers["bn1.running_var"].cpu().numpy(),
                 n_buckets=len(tr.buckets), n_launched=n_launched, n_bn_collectives=tr.collective_count)
    finally:
        dist.destroy_process_group()


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b * b).sum()) + 1e-30))


@pytest.mark.parametrize("sync_bn,fused,head", [(True, False, "dconv"), (False, False, "dconv"), (True, True, "dconv"), (True, False, "duc")])
def test_two_rank_step_matches_single_rank(sync_bn, fused, head, tmp_path):
    """DDP + SyncBatchNorm semantics (ddp...:89-93): 2 ranks x 2 images == 1 rank x 4 images when BN statistics are synced;
    without SyncBN both ranks still end with identical parameters (same averaged gradient, same Adam)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_worker, args=(2, port, sync_bn, 8.0, str(tmp_path), fused, head), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in (0, 1))
    assert int(r0["n_buckets"]) > 4 and int(r0["n_launched"]) >= int(r0["n_buckets"]) - 1      # buckets went out during backward
    np.testing.assert_array_equal(r0["grad"], r1["grad"])
    np.testing.assert_array_equal(r0["param"], r1["param"])
    # SyncBatchNorm exchanges per step: 56 BN layers, the conv1 / projection-shortcut pair of the 4 stage-opening bottlenecks shares one
    # forward message and the bn3 / shortcut pair one backward message -> 52 + 52 (was 56 + 56 plus a second pass over every z)
    # (DUC head: 55 BN layers - two DUC blocks instead of three deconvs -> 51 + 51)
    assert int(r0["n_bn_collectives"]) == ((104 if head == "dconv" else 102) if sync_bn else 0), int(r0["n_bn_collectives"])
    if not sync_bn:
        return
    np.testing.assert_array_equal(r0["rm"], r1["rm"])
    model, _ = _model(3, head)
    x, t, w = _batch(4, 64, 64, 11)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    tr = PoseTrainer(model, in_h=64, in_w=64, lr=1e-3)
    loss = tr.forward_backward(xs, ts, ws).item()
    grad = tr.flat.grad.cpu().numpy().copy()
    tr.optimizer_step(1.0)
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - loss) < 1e-5 * abs(loss)
    # same arithmetic up to the order of the per-channel sums; the net amplifies that (see the fp32-vs-fp64 note above)
    # (round 4: 2.07e-3 on the DUC head after the epilogues' wave rows are added in-launch - one more change of summation order; fp32 itself sits
    # 3e-3 from fp64 on this metric, which is where the bar belongs)
    assert _l2(r0["grad"], grad) < 4e-3, _l2(r0["grad"], grad)
    np.testing.assert_allclose(r0["rm"], tr.buffers["layer4.2.bn3.running_mean"].cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r0["rv"], tr.buffers["bn1.running_var"].cpu().numpy(), rtol=1e-4, atol=1e-7)
    agree = np.mean(np.sign(r0["param"] - _flat_init(3, head)) == np.sign(tr.flat.data.cpu().numpy() - _flat_init(3, head)))
    assert agree > 0.99, agree                     # first Adam step = lr * sign(grad): the update direction agrees


def _one_rank_rccl_worker(rank, port, out_dir):
    """A 1-rank RCCL group on the GPU: every collective of the SyncBatchNorm + DDP step is issued for real (nothing to exchange, so the
    results must equal the local step's), through torch.distributed and through the library's own communicator, eagerly and captured."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x, t, w = _batch(4, 64, 64, 11)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    res = {}
    for mode in ("local", "torch", "native", "native_graph"):
        model, _ = _model(3)
        tr = PoseTrainer(model, in_h=64, in_w=64, lr=1e-3, bucket_mb=8.0, collectives=mode != "local", native_comm=False)
        if mode != "local":
            tr.force_collectives, tr.sync_bn = True, True
        if mode.startswith("native"):
            tr._native_comm_wanted = True
            tr._open_native_comm()
            assert tr._comm is not None
        if mode == "native_graph":
            g = tr.capture(xs, ts, ws, warmup=2)
            losses = [g.step(xs, ts, ws).item() for _ in range(2)]
        else:
            losses = [tr.step(xs, ts, ws).item() for _ in range(4)][2:]
        torch.cuda.synchronize()
        res[mode] = (losses, tr.flat.data.cpu().numpy().copy(), tr.collective_count)
        tr.close()
    np.savez(os.path.join(out_dir, "one_rank.npz"), **{f"{m}_{k}": v for m, (l, p, c) in res.items() for k, v in (("loss", np.array(l)), ("param", p), ("n", np.array(c)))})
    dist.destroy_process_group()


def test_one_rank_rccl_collectives_leave_the_step_unchanged(tmp_path):
    """Preflight of the RCCL leg on the one GPU there is: with a 1-rank nccl group every SyncBatchNorm message (104 per step) and every
    gradient bucket really goes through RCCL - via torch.distributed (its own stream + events) and via sp_comm_allreduce_sum_f32 (on the
    step's own streams; also inside a captured step) - and the three ways give the same parameters after four steps bit for bit
    (the local-statistics step agrees to rounding)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_one_rank_rccl_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "one_rank.npz")
    assert int(r["local_n"]) == 0 and int(r["torch_n"]) == 104 and int(r["native_n"]) == 104 and int(r["native_graph_n"]) == 104
    for mode in ("native", "native_graph"):                      # the same SyncBatchNorm arithmetic, three ways of issuing the messages
        np.testing.assert_array_equal(r[f"{mode}_loss"], r["torch_loss"])
        np.testing.assert_array_equal(r[f"{mode}_param"], r["torch_param"])
    # against the local-statistics step: the SyncBatchNorm path forms mean / variance from exchanged sums (another summation order)
    np.testing.assert_allclose(r["torch_loss"], r["local_loss"], rtol=1e-3)
    assert _l2(r["torch_param"], r["local_param"]) < 1e-3


def _two_rank_rccl_worker(rank, port, out_dir):
    """Two ranks on two GPUs over RCCL: the SyncBatchNorm + DDP step with its collectives issued through torch.distributed, through the
    library's own two communicators (sp_comm_*: fp32 buckets and backward sums, fp64 forward sums), and the latter captured."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=rank, world_size=2, device_id=dev)
    from simple_pose_amd.sharding import rank_indices
    x, t, w = _batch(4, 64, 64, 11)
    idx = rank_indices(4, rank, 2)
    xs, ts, ws = (torch.from_numpy(v[idx]).to(dev) for v in (x, t, w))
    res = {}
    for mode in ("torch", "native", "native_graph"):
        model, _ = _model(3 + rank)
        model = model.to(dev)
        tr = PoseTrainer(model, in_h=64, in_w=64, lr=1e-3, bucket_mb=8.0, sync_bn=True, native_comm=mode != "torch")
        assert (tr._comm is not None) == (mode != "torch")
        if mode == "native_graph":
            g = tr.capture(xs, ts, ws, warmup=2)
            losses = [g.step(xs, ts, ws).item() for _ in range(2)]
        else:
            losses = [tr.step(xs, ts, ws).item() for _ in range(4)][2:]
        torch.cuda.synchronize(dev)
        res[mode] = (losses, tr.flat.data.cpu().numpy().copy(), tr.collective_count,
                     tr.buffers["layer4.2.bn3.running_mean"].cpu().numpy().copy())
        tr.close()
    np.savez(os.path.join(out_dir, f"two_rank_{rank}.npz"),
             **{f"{m}_{k}": v for m, (l, p, c, rm) in res.items() for k, v in (("loss", np.array(l)), ("param", p), ("n", np.array(c)), ("rm", rm))})
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL between two ranks (the 1-GPU boxes of this pool run the gloo and 1-rank RCCL variants)")
def test_two_rank_rccl_native_comm_matches_torch_distributed(tmp_path):
    """ADVICE r3 (high / medium): the direct-RCCL path with a PEER.  Two ranks x two images over nccl: the step's 104 SyncBatchNorm messages
    (fp64 forward sums through sp_comm_allreduce_sum_f64, fp32 backward sums) and gradient buckets through our own two communicators,
    eagerly and captured, give the parameters / running statistics of the torch.distributed path bit for bit, the same on both ranks, and
    the whole-batch statistics of ONE rank x four images to rounding (ddp...:89-93)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_rccl_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"two_rank_{r}.npz") for r in (0, 1))
    for mode in ("torch", "native", "native_graph"):
        assert int(r0[f"{mode}_n"]) == 104
        np.testing.assert_array_equal(r0[f"{mode}_param"], r1[f"{mode}_param"])
        np.testing.assert_array_equal(r0[f"{mode}_rm"], r1[f"{mode}_rm"])
    for mode in ("native", "native_graph"):
        np.testing.assert_array_equal(r0[f"{mode}_param"], r0["torch_param"])
        np.testing.assert_array_equal(r0[f"{mode}_loss"], r0["torch_loss"])
    model, _ = _model(3)
    x, t, w = _batch(4, 64, 64, 11)
    tr = PoseTrainer(model, in_h=64, in_w=64, lr=1e-3)
    for _ in range(4):
        tr.step(*(torch.from_numpy(v).to(DEV) for v in (x, t, w)))
    assert _l2(r0["native_param"], tr.flat.data.cpu().numpy()) < 1e-3
    np.testing.assert_allclose(r0["native_rm"], tr.buffers["layer4.2.bn3.running_mean"].cpu().numpy(), rtol=1e-3, atol=1e-5)


def _flat_init(seed, head="dconv"):
    from simple_pose_amd.train import FlatParams
    m, _ = _model(seed, head)
    return FlatParams(m).data.cpu().numpy()


@pytest.mark.parametrize("bf16", [False, True])
def test_maxpool_index_pair_matches_gather_kernel_and_torch(bf16):
    """nn.MaxPool2d(3,2,1) forward-with-index + index backward == the compare-based backward kernel == torch autograd, with many
    exact ties (values quantised to 1/4) so that the first-maximum rule is exercised."""
    from simple_pose_amd import _lib
    P = _lib.ptr
    B, H, W, C = 3, 18, 14, 8
    g = torch.Generator().manual_seed(5)
    x = (torch.randint(-3, 6, (B, H, W, C), generator=g).float() / 4).clamp_min(0)          # post-ReLU like, many zeros and ties
    dy = torch.randn(B, H // 2, W // 2, C, generator=g)
    xt = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yt = torch.nn.functional.max_pool2d(xt, 3, 2, 1)
    yt.backward(dy.permute(0, 3, 1, 2))
    ref_dx = xt.grad.permute(0, 2, 3, 1).contiguous()
    xd = x.to(DEV).to(torch.bfloat16 if bf16 else torch.float32).contiguous()
    dyd = dy.to(DEV).contiguous()
    y = torch.empty((B, H // 2, W // 2, C), dtype=xd.dtype, device=DEV)
    idx = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=DEV)
    dx_new = torch.empty((B, H, W, C), dtype=torch.float32, device=DEV)
    dx_old = torch.empty_like(dx_new)
    lib, st = _lib.lib(), _lib.current_stream()
    _lib.check(lib.sp_maxpool3x3s2_idx_nhwc(P(xd), int(bf16), P(y), P(idx), B, H, W, C, st))
    _lib.check(lib.sp_maxpool3x3s2_bwd_idx_nhwc(P(idx), P(dyd), 0, P(dx_new), B, H, W, C, st))
    _lib.check(lib.sp_maxpool3x3s2_bwd_nhwc(P(xd), int(bf16), P(dyd), P(dx_old), B, H, W, C, st))
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu(), yt.detach().permute(0, 2, 3, 1))                   # quarter values are exact in bf16
    assert torch.equal(dx_new, dx_old)
    assert torch.equal(dx_new.cpu(), ref_dx)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_repack_walks_give_the_same_packed_weights(dtype, monkeypatch):
    """sp_permute4_batched visits a job in destination order, as a tap loop per (i0, i3) pair, or as a tiled transpose (the `walk` field of
    a job): every packed forward / dgrad copy of the whole net must come out bit for bit the same whichever walk wrote it.  (Walk 3, the
    LDS-tiled multi-tap transpose of round 5, is off by default - it moves fewer bytes but measured slower in the step - and switched on here.)"""
    monkeypatch.setattr(PoseTrainer, "repack_tiled", True)
    model, _ = _model(5)
    tr = PoseTrainer(model, dtype=dtype)
    names = [(n, "fwd", L.w_fwd) for n, L in tr.layers.items()] + [(n, f"dgrad{i}", w) for n, L in tr.layers.items() if L.need_dgrad
                                                                       for i, w in enumerate(L.w_dgrad)]
    with_walks = [w.clone() for _, _, w in names]
    tab = tr._pack_table.cpu().numpy().view(np.uint8).reshape(-1, 104).copy()
    walks = tab[:, 92:96].view(np.int32).reshape(-1)
    assert set(walks.tolist()) == {0, 1, 2, 3}              # all four walks occur in ResNet-50 (3: the LDS-tiled multi-tap transpose, round 5)
    tab[:, 92:96] = 0                                        # destination order everywhere
    tr._pack_table = torch.from_numpy(tab.reshape(-1)).to(DEV)
    for _, _, w in names:
        w.fill_(7.0)
    tr.repack()
    torch.cuda.synchronize()
    for (n, kind, w), ref in zip(names, with_walks):
        assert torch.equal(w, ref), (n, kind)


def test_solver_counterpart_trains_validates_and_checkpoints(tmp_path):
    """DDPProcessor (ddp...:20-212) on synthetic data, single process: two epochs with a MultiStepLR drop, loss goes down, val()
    runs the eval-mode forward with the UPDATED weights, and the checkpoint has the reference's {"ema": state_dict, "epoch"} form."""
    import yaml
    from simple_pose_amd.processors.ddp_pose_resnet_solver import DDPProcessor
    cfg = {"model_name": "t", "gpus": "0",
           "data": {"synthetic": 8, "batch_size": 4, "num_workers": 0, "debug": False},
           "model": {"type": "pose_resnet_duc", "name": "resnet50", "num_joints": 17, "pretrained": False},
           "optim": {"lr": 1e-3, "amp": False, "sync_bn": True, "milestones": [1], "epochs": 2, "gamma": 0.1},
           "val": {"interval": 1, "weight_path": str(tmp_path / "w")}}
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    proc = DDPProcessor(str(path))
    v0 = proc.val(-1)
    proc.run()
    assert [h["lr"] for h in proc.history] == [1e-3, 1e-4]
    assert proc.history[1]["loss"] < proc.history[0]["loss"]
    v1 = proc.val(1)
    assert v1["loss"] != v0["loss"] and v1["results"] == 4 * len(proc.vloader)       # eval program was rebuilt from the new weights
    ck = torch.load(tmp_path / "w" / "t_last.pth")
    assert set(ck) == {"ema", "epoch"} and ck["epoch"] == 1
    assert set(ck["ema"]) == {k for k, _, _ in nets_oracle.state_dict_shapes_resnet50("duc")}
    assert float(ck["ema"]["bn1.num_batches_tracked"]) == 4.0                            # 2 epochs x 2 iterations


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_streamed_schedule_equals_plain_schedule(dtype):
    """step() runs weight gradients on a second stream and the optimizer (Adam + re-packing, bucket by bucket) on a third while
    backward continues; the plain schedule does everything in order on one stream.  Same kernels, same operands: parameters,
    optimizer state, BN buffers and losses must be bit-identical after several steps (a missing dependency would show here)."""
    x, t, w = _batch(4, 128, 96, 3)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    out = []
    for streamed in (True, False):
        model, _ = _model(3)
        tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3, dtype=dtype, overlap_wgrad=streamed, bucket_mb=8.0)
        tr.fuse_optimizer = streamed
        losses = [tr.step(xs, ts, ws).item() for _ in range(4)]
        torch.cuda.synchronize()
        out.append((losses, tr.flat.data.clone(), tr.exp_avg_sq.clone(), model.bn1.running_var.clone(),
                    [l.w_fwd.clone() for l in tr.layers.values()]))
    (l0, p0, v0, r0, w0), (l1, p1, v1, r1, w1) = out
    assert l0 == l1 and l0[-1] < l0[0]
    assert torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(r0, r1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("sync_bn_path", [False, True], ids=["local_bn", "sync_bn_path"])
def test_captured_step_equals_eager_steps_bitwise(dtype, sync_bn_path):
    """PoseTrainer.capture(): the whole step (three streams, optimizer inside backward, Adam's scalars from device memory) recorded into
    one hipGraph and replayed with fresh batches gives the parameters, optimizer state, BN buffers, packed weights and losses of the
    same steps run eagerly, bit for bit - a dependency that only stream order provided, or a pointer that moved between capture and
    replay, would show here.  `sync_bn_path`: with the SyncBatchNorm message path switched on (a delay kernel on the message stream
    stands in for the all-reduce on one rank), i.e. with a fourth stream inside the capture."""
    batches = []
    for seed in (3, 4, 5):
        x, t, w = _batch(4, 128, 96, seed)
        batches.append(tuple(torch.from_numpy(v).to(DEV) for v in (x, t, w)))
    out = []
    for graphed in (False, True):
        model, _ = _model(3)
        tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3, dtype=dtype, bucket_mb=8.0, sync_bn_latency_us=2.0 if sync_bn_path else 0.0)
        losses = []
        if graphed:
            g = tr.capture(*batches[0], warmup=2)                      # two eager steps on batch 0, then the capture
            for i, b in enumerate(batches[1:] + batches[:2]):
                # (the third of them eagerly, between replays: the trainer stays usable while a captured step exists)
                losses.append((tr.step(*b) if i == 2 else g.step(*b)).item())
            assert tr.step_count == 2 + 4
        else:
            for b in [batches[0]] * 2 + batches[1:] + batches[:2]:
                losses.append(tr.step(*b).item())
            losses = losses[2:]
        torch.cuda.synchronize()
        out.append((losses, tr.flat.data.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), model.bn1.running_var.clone(),
                    model.bn1.num_batches_tracked.clone(), [l.w_fwd.clone() for l in tr.layers.values()]))
    (l0, p0, m0, v0, r0, n0, w0), (l1, p1, m1, v1, r1, n1, w1) = out
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(r0, r1) and torch.equal(n0, n1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1))


@pytest.mark.parametrize("dtype,ltol,tol", [("fp32", 1e-5, 2e-3), ("bf16", 1e-3, 0.3)])
def test_trainer_autotune_only_moves_speed(measured, dtype, ltol, tol):
    """PoseTrainer.autotune pins another tile per forward / dgrad launch: conv results are tile-independent, BN partial sums regroup
    (fp32 rounding of <= 64-term sums: batch statistics move by ~1e-7).  Which tiles win is a timing outcome, so the comparison is a
    range, not a number: measured over repeated runs the gradient moved by 0 ... 4.3e-4 (fp32) and 4e-4 ... 0.11 (bf16: a 1e-7 change of
    a mean flips bf16 roundings downstream, and this net amplifies - the same regime as the two-rank test's 2e-3 and the bf16-vs-AMP
    test) relative L2, the loss by <= 8e-8 / 2e-4.  The bars sit at that regime's edge: they catch a broken tile, not a rounding."""
    x, t, w = _batch(8, 128, 96, 5)
    xs, ts, ws = (torch.from_numpy(v).to(DEV) for v in (x, t, w))
    res = []
    for tune in (False, True):
        model, _ = _model(3)
        tr = PoseTrainer(model, in_h=128, in_w=96, lr=1e-3, dtype=dtype)
        if tune:
            table = tr.autotune(8, reps=2, rounds=1)
            assert tr.tuned_for_batch == 8 and len(table) > 100
        loss = tr.forward_backward(xs, ts, ws).item()
        res.append((loss, tr.flat.grad.clone()))
    (l0, g0), (l1, g1) = res
    rel = float((g0 - g1).norm() / g0.norm())
    measured("loss_rel_diff", abs(l0 - l1) / abs(l0), ltol)
    measured("grad_rel_l2_diff", rel, tol)
    assert abs(l0 - l1) <= ltol * abs(l0), (l0, l1)
    assert rel <= tol, rel


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 5e-3)])
def test_full_size_step_properties(dtype, tol):
    """BASELINE config 4 size (32 images of 256x192 per GPU), where the oracle is too slow to be the checker: size-independent
    properties instead.  (1) the step is bit-reproducible; (2) train-mode BN makes the loss and the BN statistics a function of
    the batch as a SET: permuting the 32 samples changes only summation orders; (3) Adam's first step moves every parameter by
    at most lr."""
    B = 32
    x, t, w = _batch(B, 256, 192, 21)
    perm = np.random.default_rng(0).permutation(B)
    runs = {}
    for tag, idx in (("a", np.arange(B)), ("a2", np.arange(B)), ("perm", perm)):
        model, _ = _model(5)
        tr = PoseTrainer(model, lr=1e-3, dtype=dtype)
        p0 = tr.flat.data.clone()
        xs, ts, ws = (torch.from_numpy(v[idx]).to(DEV) for v in (x, t, w))
        loss = tr.step(xs, ts, ws).item()
        torch.cuda.synchronize()
        runs[tag] = (loss, tr.flat.data.clone(), model.layer3[2].bn2.running_var.clone(), (tr.flat.data - p0).abs().max().item())
    assert runs["a"][0] == runs["a2"][0] and torch.equal(runs["a"][1], runs["a2"][1])            # (1)
    assert abs(runs["perm"][0] - runs["a"][0]) <= tol * abs(runs["a"][0])                         # (2)
    rel = ((runs["perm"][2] - runs["a"][2]).abs().max() / runs["a"][2].abs().max()).item()
    assert rel <= max(tol, 1e-4), rel
    assert 0 < runs["a"][3] <= 1e-3 * (1 + 2e-4)           # (3): lr * |g| / (|g| + eps) <= lr, plus the rounding of p itself
    assert np.isfinite(runs["a"][0]) and runs["a"][0] > 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_shortcut_branch_stream_changes_no_bits(dtype, monkeypatch):
    """The projection shortcuts run on a branch stream beside the main chain (forward and backward).  Same kernels, same accumulation
    order: three steps give the same parameters bit for bit as the serial schedule - through PoseTrainer.step (step arena) and through
    the autograd surface (caching allocator: the tensors that cross streams must be kept alive / recorded on the right stream)."""
    x, t, w = _batch(4, 64, 64, 11)
    xd, td, wd = (torch.from_numpy(a).to(DEV) for a in (x, t, w))

    def steps(branch):
        monkeypatch.setenv("SP_BRANCH", "1" if branch else "0")
        m, _ = _model(11)
        tr = PoseTrainer(m, in_h=64, in_w=64, lr=1e-3, dtype=dtype)
        assert tr.overlap_shortcut == branch
        for _ in range(3):
            tr.step(xd, td, wd)
        torch.cuda.synchronize()
        return tr.flat.data.clone()

    def autograd(branch):
        monkeypatch.setenv("SP_BRANCH", "1" if branch else "0")
        m, _ = _model(11)
        m.compute_dtype = dtype
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        for _ in range(3):
            opt.zero_grad()
            p = m(xd)
            (0.5 * torch.nn.functional.mse_loss(p * wd[..., None, None], td * wd[..., None, None])).backward()
            opt.step()
        torch.cuda.synchronize()
        return torch.cat([q.detach().reshape(-1) for q in m.parameters()]).clone()

    for fn in (steps, autograd):
        ref = fn(False)
        for _ in range(2):
            assert torch.equal(fn(True), ref), fn.__name__


@pytest.mark.parametrize("grad_dtype,head", [("bf16", "dconv"), ("fp32", "dconv"), ("bf16", "duc")])
def test_batchnorm_relu_inside_the_consumer_conv_changes_no_bits(grad_dtype, head, monkeypatch):
    """Round 5: bn2 + ReLU of a Bottleneck is formed in conv3's staging pass (sp_conv2d_fwd_bn_stats_abn; `PoseTrainer.apply_in_consumer`),
    which also writes the activation and its ReLU bit mask for the backward pass - 16 launches less on the forward chain.  Same element map,
    same roundings: three bf16 steps give the same parameters, moments and BatchNorm buffers bit for bit as the stand-alone passes."""
    x, t, w = _batch(4, 128, 96, 17)
    xd, td, wd = (torch.from_numpy(a).to(DEV) for a in (x, t, w))

    def steps(on):
        monkeypatch.setattr(PoseTrainer, "apply_in_consumer", on)
        m, _ = _model(17, head)
        tr = PoseTrainer(m, in_h=128, in_w=96, lr=1e-3, dtype="bf16", grad_dtype=grad_dtype)
        losses = [tr.step(xd, td, wd).item() for _ in range(3)]
        torch.cuda.synchronize()
        bufs = torch.cat([b.detach().reshape(-1).double() for b in m.buffers()])
        return losses, tr.flat.data.clone(), tr.exp_avg.clone(), bufs

    ref = steps(False)
    got = steps(True)
    assert got[0] == ref[0]
    for a, b in zip(got[1:], ref[1:]):
        assert torch.equal(a, b)


def test_abn_conv_launch_equals_the_pass_plus_the_plain_launch():
    """sp_conv2d_fwd_bn_stats_abn alone: on z of 64 / 256 / 512 channels (1, 4, 8 K tiles), M not a multiple of the tile, with and without the
    mask: y, mask, the conv output and both partial-sum arrays equal sp_bn_apply_nhwc(relu) + sp_conv2d_fwd_bn_stats bit for bit, on every tile."""
    from simple_pose_amd import _lib
    from simple_pose_amd.train import ConvT
    lib, st = _lib.lib(), _lib.current_stream()
    P = _lib.ptr
    g = torch.Generator().manual_seed(5)

    class _Tr:                       # the little of a trainer a ConvT needs
        bf16, g16, kernel_events = True, True, None
        grad_dtype = torch.bfloat16
    for Cin, Cout, B, H, W in ((64, 256, 3, 9, 7), (256, 1024, 2, 5, 6), (512, 2048, 3, 4, 3), (128, 512, 5, 8, 6)):
        wt = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(DEV)
        flat_stub = type("F", (), {})()
        tr = _Tr()
        tr.flat = flat_stub
        layer = ConvT(tr, "c", "conv", wt, H, W)
        # pack the forward copy by hand (PackJob 0 = the forward pack)
        j = layer.pack_jobs[0]
        import ctypes
        d4 = (ctypes.c_int32 * 4)(*j.dims); s4 = (ctypes.c_int64 * 4)(*j.strides); v4 = (ctypes.c_int32 * 4)(*j.valid)
        _lib.check(lib.sp_permute4_f32(P(wt.reshape(-1)), P(j.dst), 1, d4, s4, v4, j.base, j.dst_off, st), "pack")
        rows = B * H * W
        z = (torch.randn(B, H, W, Cin, generator=g) * 1.2 + 0.1).bfloat16().to(DEV)
        mean, invstd = torch.randn(Cin, generator=g).mul(0.2).to(DEV), (0.5 + torch.rand(Cin, generator=g)).to(DEV)
        gamma, beta = (0.75 + 0.5 * torch.rand(Cin, generator=g)).to(DEV), (0.2 * torch.randn(Cin, generator=g)).to(DEV)
        for with_mask in (True, False):
            y_ref = torch.empty_like(z)
            m_ref = torch.empty(rows * Cin // 8, dtype=torch.uint8, device=DEV) if with_mask else None
            _lib.check(lib.sp_bn_apply_nhwc(P(z), 1, P(mean), P(invstd), P(gamma), P(beta), None, P(y_ref), rows, Cin, 1, P(m_ref), st), "apply")
            for tile in ((128, 128), (64, 128), (128, 64), (64, 64), (256, 64), (128, 32)):
                if layer.d_fwd.n_pad % tile[1]:
                    continue
                layer.d_fwd.tile_m, layer.d_fwd.tile_n = tile
                layer._rows_cache.clear()
                o_ref, p_ref, r_ref = layer.forward_bn_stats(y_ref, B)
                y = torch.full_like(z, float("nan"))
                m = torch.full((rows * Cin // 8,), 0xA5, dtype=torch.uint8, device=DEV) if with_mask else None
                o, p, r = layer.forward_bn_stats_abn(z, B, mean, invstd, gamma, beta, y, m)
                torch.cuda.synchronize()
                assert r == r_ref
                assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), (Cin, tile)
                if with_mask:
                    assert torch.equal(m, m_ref), (Cin, tile)
                assert torch.equal(o.view(torch.int16), o_ref.view(torch.int16)), (Cin, tile)
                assert torch.equal(p, p_ref), (Cin, tile)


def test_lazy_residual_gradient_changes_no_bits():
    """bf16 gradients: bn3's backward pass of an identity Bottleneck does not write the residual share g = dy * mask; conv1's dgrad forms it
    from (dy, ReLU bit mask) in its epilogue (sp_conv2d_dgrad_bn_bwd_stats_macc).  g is dy or zero, so three steps end on the same bits."""
    x, t, w = _batch(4, 64, 64, 13)
    xd, td, wd = (torch.from_numpy(a).to(DEV) for a in (x, t, w))
    outs = []
    for lazy in (False, True, True):
        m, _ = _model(13)
        tr = PoseTrainer(m, in_h=64, in_w=64, lr=1e-3, dtype="bf16")
        assert tr.g16
        tr.lazy_residual_grad = lazy
        losses = [float(tr.step(xd, td, wd)) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((losses, tr.flat.data.clone()))
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[1][1], outs[2][1])


# ---------------------------------------------------------------------------------------------- the reference's own loop, through autograd
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_reference_training_loop_through_autograd(dtype):
    """`model.train(); optimizer.zero_grad(); predicts = model(x); loss = 0.5 * MSE(predicts * m, targets * m); loss.backward();
    optimizer.step()` - the body of DDPProcessor.train (ddp...:114-119) verbatim, with torch.optim.Adam - runs on the HIP tape through
    one autograd node.  Gradients equal the fused trainer's (same kernels; only d loss / d heat comes from torch instead of
    sp_masked_mse), three steps track PoseTrainer.step (the library's Adam kernel against torch's), BatchNorm running statistics and
    num_batches_tracked advance as in torch, and `.grad` accumulates like autograd's when it is not reset."""
    B, H, W = 4, 64, 64
    x, t, w = _batch(B, H, W, 11)
    xd, td, wd = (torch.from_numpy(a).to(DEV) for a in (x, t, w))
    ma, _ = _model(11)
    mb, _ = _model(11)
    if dtype == "bf16":
        ma.compute_dtype = "bf16"
    opt = torch.optim.Adam(ma.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    trb = PoseTrainer(mb, in_h=H, in_w=W, lr=1e-3, dtype=dtype)
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    for step in range(3):
        opt.zero_grad()
        predicts = ma(xd)
        assert predicts.requires_grad and predicts.shape == (B, 17, H // 4, W // 4)
        if step == 0:
            predicts.retain_grad()
        loss = 0.5 * crit(predicts * wd[..., None, None], td * wd[..., None, None])
        loss.backward()
        if step == 0:
            # (1) same tape, same d loss / d heat -> the same bits as the trainer's own forward_tape / backward
            mc, _ = _model(11)
            trc = PoseTrainer(mc, in_h=H, in_w=W, lr=1e-3, dtype=dtype)
            heat_c, bwd_c = trc.forward_tape(xd)
            assert torch.equal(heat_c, predicts.detach())
            bwd_c(predicts.grad)
            torch.cuda.synchronize()
            for k in pa:
                assert torch.equal(pa[k].grad, trc.flat.view(k, grad=True).view(pa[k].shape)), k
            # (2) against the fused step, whose d loss / d heat comes from sp_masked_mse instead of torch's MSELoss backward (1 ulp
            # apart): measured 3.0e-6 worst per-tensor deviation (a BatchNorm bias gradient: a sum with cancellation)
            lb = trb.forward_backward(xd, td, wd)
            torch.cuda.synchronize()
            assert abs(loss.item() - lb.item()) <= 1e-6 * abs(lb.item())
            worst = max((float((pa[k].grad - trb.flat.view(k, grad=True).view(pa[k].shape)).abs().max() /
                               (trb.flat.view(k, grad=True).abs().max() + 1e-30)), k) for k in pa)
            print(f"autograd vs fused trainer, {dtype}: worst per-tensor gradient deviation {worst[0]:.2e} ({worst[1]})")
            assert worst[0] < (1e-5 if dtype == "fp32" else 2e-3), worst
            trb.optimizer_step()                              # (completes mb's first step: forward_backward + Adam + repack)
        else:
            trb.step(xd, td, wd)
        opt.step()
        if step == 0:                                         # one Adam step: torch.optim.Adam here, sp_adam_step there
            torch.cuda.synchronize()
            d1 = max(float((pa[k] - pb[k]).abs().max()) for k in pa)
            f1 = np.mean([float(((pa[k] - pb[k]).abs() <= 1e-6).float().mean()) for k in pa])
            print(f"after 1 step, {dtype}: max parameter deviation {d1:.2e}, {100 * f1:.4f} % within 1e-6")
            assert f1 > (0.999 if dtype == "fp32" else 0.98) and d1 <= 2.1e-3   # (a sign flip of a ~0 gradient moves a weight by 2 lr)
    torch.cuda.synchronize()
    # Three steps of this deep BatchNorm net at 4 images are chaotic: the 3e-6 gradient difference of step 1 grows (measured fp32:
    # 95.2 % of the weights within 2e-5 after 3 steps, max 2.9e-3 = 3 lr) - the two runs track each other, they are not equal.
    dev = max(float((pa[k] - pb[k]).abs().max()) for k in pa)
    frac = np.mean([float(((pa[k] - pb[k]).abs() <= 2e-5).float().mean()) for k in pa])
    print(f"3 reference-loop steps vs 3 PoseTrainer steps, {dtype}: max parameter deviation {dev:.2e}, {100 * frac:.3f} % within 2e-5")
    frac_lr = np.mean([float(((pa[k] - pb[k]).abs() <= 1e-3).float().mean()) for k in pa])
    print(f"    ... {100 * frac_lr:.3f} % within one learning rate (1e-3)")
    # bf16: a 1e-7 weight difference can flip the bf16 rounding of an operand, so the runs part faster (measured 29 % within 2e-5)
    assert (frac > 0.90 if dtype == "fp32" else frac_lr > 0.90) and dev <= 6.1e-3
    ba = dict(ma.named_buffers())
    for k in ba:
        if k.endswith("num_batches_tracked"):
            assert int(ba[k]) == 3, k
    # eval mode after training: the inference program is rebuilt from the updated parameters and running statistics - the same heat
    # maps, bit for bit, as a fresh model that loads this state_dict (no stale packed weight anywhere)
    ma.eval()
    fresh = pose_resnet_dconv.resnet50(pretrained=False, num_classes=17)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in ma.state_dict().items()}, strict=True)
    fresh = fresh.to(DEV).eval()
    fresh.compute_dtype = ma.compute_dtype
    with torch.no_grad():
        assert torch.equal(ma(xd), fresh(xd))
    # ... and in train mode too: the tape's packed copies were refreshed after the torch optimizer's in-place updates
    ma.train(); fresh.train()
    with torch.no_grad():
        assert torch.equal(ma(xd), fresh(xd))
    # accumulation semantics: a second backward without zero_grad adds to .grad (the kernels overwrite their buffer, the surface
    # switches buffers when live gradients alias it)
    opt.zero_grad()
    l1 = 0.5 * crit(ma(xd) * wd[..., None, None], td * wd[..., None, None]); l1.backward()
    g1 = {k: v.grad.clone() for k, v in pa.items()}
    l2 = 0.5 * crit(ma(xd) * wd[..., None, None], td * wd[..., None, None]); l2.backward()
    k = "layer3.2.conv2.weight"
    # (the second forward saw BN running statistics one step further, but batch statistics - and so the gradients - are the same)
    assert torch.allclose(pa[k].grad, 2 * g1[k], rtol=1e-5, atol=1e-12)
    opt.zero_grad(set_to_none=False)
    l3 = 0.5 * crit(ma(xd) * wd[..., None, None], td * wd[..., None, None]); l3.backward()
    assert torch.allclose(pa[k].grad, g1[k], rtol=1e-5, atol=1e-12)


def test_train_mode_forward_of_the_selayer_net_through_autograd():
    """`resnet50(reduction=True)` in train() mode through the autograd surface (no trainer object in sight): loss.backward() fills every
    parameter's .grad, the SELayer's FC weights and biases included, finite and non-zero."""
    from simple_pose_amd.nets import pose_resnet_dconv as prd
    torch.manual_seed(0)
    m = prd.resnet50(pretrained=False, num_classes=17, reduction=True).to(DEV).train()
    with torch.no_grad():
        for p in m.parameters():                                  # the reference's init (std 1e-3) gives gradients too small to look at
            if p.dim() == 4:
                p.normal_(0, (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
    out = m(torch.randn(2, 3, 64, 64, device=DEV))
    assert out.shape == (2, 17, 16, 16) and out.requires_grad
    out.square().mean().backward()
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    for k in ("layer1.0.se.fc.0.weight", "layer1.0.se.fc.0.bias", "layer3.0.se.fc.2.weight", "layer4.0.se.fc.2.bias"):
        assert float(grads[k].abs().max()) > 0, k
