"""Parameter containers + HIP dispatch shared by the DConv and DUC ResNet pose nets.

The module tree exists only to own parameters/buffers under the reference's exact names so that
`state_dict()` / `load_state_dict()` / `.to()` / `.parameters()` behave like the reference's
(`nets/pose_resnet_dconv.py:136-268`, `nets/pose_resnet_duc.py:136-251`).  No torch op computes anything in
`forward`: the tensors are packed once into the HIP library's layout (simple_pose_amd.engine) and re-packed whenever
a parameter changes (load_state_dict, optimizer step, .to()).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import engine
from .._lib import HipLibraryError, require_cuda_f32


class _Bottleneck(nn.Module):
    """Parameter holder for one bottleneck block: conv1/bn1 (1x1), conv2/bn2 (3x3, carries the stride), conv3/bn3 (1x1 x4),
    optional `downsample` = Sequential(conv1x1, bn).  Keys as in nets/pose_resnet_dconv.py:83-110."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, with_downsample: bool, with_se: bool = False, width: int = 0, groups: int = 1):
        super().__init__()
        width = width or planes          # wide_resnet*_2 / resnext*: width = `int(planes * (base_width / 64.)) * groups` (pose_resnet_dconv.py:97)
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)      # (:101)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None
        self.stride = stride
        if with_se:                      # registered after `downsample`, as in the reference (pose_resnet_dconv.py:105-110)
            from .commons import SELayer
            self.se = SELayer(planes * 4)


class _BasicBlock(nn.Module):
    """Parameter holder for one BasicBlock (resnet18 / resnet34; pose_resnet_dconv.py:38-80): conv1/bn1 (3x3, carries the stride),
    conv2/bn2 (3x3), optional `downsample` = Sequential(conv1x1, bn), optional SELayer(planes)."""
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int, with_downsample: bool, with_se: bool = False, width: int = 0, groups: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        else:
            self.downsample = None
        self.stride = stride
        if with_se:
            from .commons import SELayer
            self.se = SELayer(planes)


class PoseResNetBase(nn.Module):
    """ResNet trunk (conv1, bn1, layer1..4 of Bottlenecks - resnet50 / 101 / 152, wide_resnet*_2 - or BasicBlocks - resnet18 / 34) + a
    head defined by the subclass (`_build_head`, `HEAD`)."""
    HEAD = ""
    BLOCKS = (3, 4, 6, 3)
    BLOCK = "bottleneck"       # or "basic"
    WIDTH_PER_GROUP = 64       # 128: wide_resnet50_2 / wide_resnet101_2 (pose_resnet_dconv.py:370-403); 4 / 8: resnext50_32x4d / resnext101_32x8d
    GROUPS = 1                 # 32: the resnext factories (pose_resnet_dconv.py:342-368)

    def __init__(self, num_classes: int = 17, reduction: bool = False, blocks=None, block: str = "bottleneck", width_per_group: int = 64,
                 groups: int = 1):
        super().__init__()
        if block not in ("bottleneck", "basic"):
            raise ValueError(block)
        if block == "basic" and (width_per_group != 64 or groups != 1):
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")      # (the reference's own check, :46-47)
        self.BLOCK, self.WIDTH_PER_GROUP, self.GROUPS = block, width_per_group, groups
        if blocks is not None:           # resnet101 / resnet152: same bottleneck trunk, other depths (pose_resnet_dconv.py:318-339)
            self.BLOCKS = tuple(blocks)
        self.reduction = reduction     # SELayer on the first block of every layer (pose_resnet_dconv.py:215-218)
        self.num_classes = num_classes
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        Block = _Bottleneck if block == "bottleneck" else _BasicBlock
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), self.BLOCKS), start=1):
            blocks = []
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 1) else 1
                # `_make_layer` (pose_resnet_dconv.py:205-221): a projection shortcut where the shape changes (always for a stage's first
                # Bottleneck; NOT for layer1.0 of the BasicBlock nets), and the SELayer only on blocks that have one
                down = bi == 0 and (stride != 1 or inplanes != planes * Block.expansion)
                blocks.append(Block(inplanes, planes, stride, with_downsample=down, with_se=(reduction and down),
                                    width=planes * width_per_group // 64 * groups, groups=groups))
                inplanes = planes * Block.expansion
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self._build_head(inplanes, num_classes)
        self._init_like_reference()
        self.compute_dtype = "fp32"   # "bf16": bf16 activations/weights with fp32 accumulation (master weights stay fp32)
        self.autotune = True  # time the legal tile shapes per layer on first use (batch >= 16); speed only
        self._program: Optional[engine.Program] = None
        self._program_key = None

    # -- reference init (pose_resnet_dconv.py:180-189): conv / deconv N(0, 0.001), conv bias 0, BN 1/0 --
    def _init_like_reference(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _build_head(self, inplanes: int, num_classes: int):
        raise NotImplementedError

    # -- HIP dispatch ------------------------------------------------------------------------------------
    fuse_bottlenecks = True    # bf16: layer1.1 / layer1.2 as one launch each (sp_bottleneck_c64); same bits, +0.6 ... 0.8 % end to end
    fuse_stem = True           # conv1 + bn1 + relu + maxpool as one launch on the fp32 NCHW image (sp_stem7_pool); same bits

    def _tensors_key(self, x):
        sd = self.state_dict(keep_vars=True)
        return (tuple(x.shape[2:]), str(x.device), self.compute_dtype, self.fuse_bottlenecks, self.fuse_stem) + tuple((v.data_ptr(), v._version) for v in sd.values())

    def hip_program(self, x: torch.Tensor) -> engine.Program:
        key = self._tensors_key(x)
        if self._program is None or key != self._program_key:
            sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            for k, v in sd.items():
                if v.device != x.device:
                    raise HipLibraryError(f"parameter {k} is on {v.device} but the input is on {x.device}; call .to(device)")
            self._program = engine.resnet_program(sd, self.HEAD, in_h=x.shape[2], in_w=x.shape[3], blocks=self.BLOCKS,
                                                   dtype=self.compute_dtype, fuse_bottlenecks=self.fuse_bottlenecks, fuse_stem=self.fuse_stem)
            self._program_key = key
        return self._program

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 NCHW [B,3,H,W] (H,W multiples of 32) on the GPU -> heat maps [B,num_classes,H/4,W/4]."""
        x = require_cuda_f32(x, "input")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H,W multiples of 32, got {tuple(x.shape)}")
        if self.training:
            return self._forward_train(x)
        prog = self.hip_program(x)
        if self.autotune and x.shape[0] >= 16 and x.shape[0] >= 4 * prog.tuned_for_batch:
            prog.autotune(x)  # once per (weights, input shape): pins the fastest tile per layer; results unchanged
        return prog.run(x)

    # -- train mode: the reference loop `predicts = model(x); loss = ...; loss.backward(); optimizer.step()` (ddp...:114-119) ------
    def _forward_train(self, x: torch.Tensor) -> torch.Tensor:
        return train_forward(self, x)

    def forward_crops(self, crops: torch.Tensor) -> torch.Tensor:
        return forward_uint8_crops(self, crops)


def train_forward(self, x: torch.Tensor) -> torch.Tensor:
    """Train-mode forward as one autograd node: the HIP train-mode forward (batch-statistics BatchNorm, running statistics
    updated) records its tape; `loss.backward()` runs the HIP backward (dgrad / wgrad / BN backward) and hands every parameter
    gradient to autograd, so `.grad`, gradient hooks (DistributedDataParallel) and any torch optimizer work as in the reference.
    `simple_pose_amd.train.PoseTrainer.step` stays the faster fused path (loss, Adam and repack as kernels of the same tape)."""
    from ..train import PoseTrainer
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not getattr(self, "_warned_local_bn", False):
        # the reference converts to SyncBatchNorm under DDP (ddp...:89-90); this surface computes LOCAL gradients and leaves the
        # exchange to the caller's DistributedDataParallel wrapper, so its BatchNorm statistics are per rank
        import warnings
        warnings.warn("simple_pose_amd: model(x) in train() mode under an initialised process group uses per-rank BatchNorm statistics "
                      "(sync_bn: False behaviour); PoseTrainer(model, process_group=..., sync_bn=True).step() is the SyncBatchNorm path")
        self._warned_local_bn = True
    key = (x.shape[2], x.shape[3], str(x.device), self.compute_dtype)
    tr = getattr(self, "_trainer", None)
    if tr is None or self._trainer_key != key or not tr.still_owns_parameters():
        tr = PoseTrainer(self, in_h=x.shape[2], in_w=x.shape[3], dtype="bf16" if self.compute_dtype == "bf16" else "fp32",
                         collectives=False)
        self._trainer, self._trainer_key = tr, key
    if not torch.is_grad_enabled():
        return tr.forward_tape(x)[0]
    return _TrainForward.apply(x, tr, *tr.sd.values())


class _TrainForward(torch.autograd.Function):
    """heat = f(x; parameters) with the HIP train step's tape as the backward (simple_pose_amd.train.PoseTrainer.forward_tape)."""

    @staticmethod
    def forward(ctx, x, trainer, *params):
        heat, backward = trainer.forward_tape(x)
        ctx.trainer, ctx.run_backward, ctx.params = trainer, backward, params
        return heat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dheat):
        if ctx.run_backward is None:
            raise RuntimeError("the HIP tape of this forward has already been consumed (backward twice)")
        run, ctx.run_backward = ctx.run_backward, None
        grads = ctx.trainer.autograd_backward(run, dheat.contiguous(), ctx.params)
        return (None, None) + grads          # no gradient w.r.t. the input image (the stem's dgrad is not computed, as in the solver)


def forward_uint8_crops(model, crops: torch.Tensor) -> torch.Tensor:
    """uint8 BGR crops [B,H,W,3] on the GPU (what `cv.warpAffine` / `datasets.naive_data.crop_boxes` produce) -> heat maps.  The
    collate normalisation of datasets/coco.py:136 (`x/255 - mean`, BGR -> RGB) happens inside the first launch, which writes the
    network's NHWC input directly; bit-identical to `model(normalize_crops(crops))`."""
    if not (isinstance(crops, torch.Tensor) and crops.is_cuda and crops.dtype == torch.uint8 and crops.dim() == 4 and crops.shape[-1] == 3):
        raise HipLibraryError("forward_crops: expected a CUDA uint8 tensor [B,H,W,3]")
    B, H, W, _ = crops.shape
    if H % 32 or W % 32:
        raise ValueError(f"expected H,W multiples of 32, got {H}x{W}")
    if model.training:
        raise NotImplementedError("forward_crops is the eval-mode path; training goes through simple_pose_amd.train.PoseTrainer")
    prog = model.hip_program(torch.empty((0, 3, H, W), device=crops.device))
    crops = crops.contiguous()
    if model.autotune and B >= 16 and B >= 4 * prog.tuned_for_batch:
        prog.autotune(crops)
    return prog.run(crops)


def load_pretrained_like_reference(model: nn.Module, arch: str):
    """The reference downloads ImageNet weights with strict=False (pose_resnet_dconv.py:271-279).  There is no
    network in the target environment, so `pretrained=True` needs a local file."""
    raise RuntimeError(
        f"pretrained=True would download {arch} ImageNet weights; no network here. Load a local checkpoint with "
        "model.load_state_dict(torch.load(path), strict=False) instead.")
