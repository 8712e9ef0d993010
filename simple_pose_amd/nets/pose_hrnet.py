"""HRNet (W32 / W48) pose network, MI355X-native: drop-in for the reference's `nets/pose_hrnet.py`.

    model = get_pose_net("simple_pose_amd/nets/hrnet_w32.yaml", pretrained=None, joint_num=17)   # pose_hrnet.py:489-496

Same yaml schema, same 1,754 state_dict keys (W32).  The forward (`PoseHighResolutionNet.forward`, pose_hrnet.py:419-454;
`HighResolutionModule.forward`, :241-259) is lowered by `engine.hrnet_program` onto the fp32 implicit-GEMM conv kernel;
the multi-resolution fuse sums use the conv epilogue's residual input (down paths), `sp_upsample_add_nhwc` (up paths)
and keep the reference's left-to-right summation order.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import yaml

from .. import engine
from .._lib import HipLibraryError, require_cuda_f32
from ._param_tree import ParamTree

Shape = Tuple[str, Tuple[int, ...], str]
F32, I64 = "torch.float32", "torch.int64"


def _conv(out: List[Shape], key: str, o: int, i: int, k: int, bias: bool = False):
    out.append((key + ".weight", (o, i, k, k), F32))
    if bias:
        out.append((key + ".bias", (o,), F32))


def _bn(out: List[Shape], key: str, c: int):
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        out.append((f"{key}.{leaf}", (c,), F32))
    out.append((f"{key}.num_batches_tracked", (), I64))


def stage_channels(extra: dict) -> List[List[int]]:
    return [list(extra[f"STAGE{s}"]["NUM_CHANNELS"]) for s in (2, 3, 4)]


def hrnet_state_dict_shapes(cfg: dict, joint_num: int = 17) -> List[Shape]:
    """(key, shape, dtype) in the reference's registration order (pose_hrnet.py:262-325)."""
    extra = cfg["MODEL"]["EXTRA"]
    out: List[Shape] = []
    _conv(out, "conv1", 64, 3, 3); _bn(out, "bn1", 64)
    _conv(out, "conv2", 64, 64, 3); _bn(out, "bn2", 64)
    inpl = 64
    for b in range(4):                                   # layer1: 4 Bottlenecks, planes 64 (:283, :368-385)
        p = f"layer1.{b}"
        _conv(out, p + ".conv1", 64, inpl, 1); _bn(out, p + ".bn1", 64)
        _conv(out, p + ".conv2", 64, 64, 3); _bn(out, p + ".bn2", 64)
        _conv(out, p + ".conv3", 256, 64, 1); _bn(out, p + ".bn3", 256)
        if b == 0:
            _conv(out, p + ".downsample.0", 256, inpl, 1); _bn(out, p + ".downsample.1", 256)
        inpl = 256
    pre = [256]
    for si, s in enumerate((2, 3, 4)):
        sc = extra[f"STAGE{s}"]
        if sc["BLOCK"] != "BASIC":
            raise NotImplementedError("only BASIC blocks in stages (as in hrnet_w32/w48.yaml)")
        cur = list(sc["NUM_CHANNELS"])
        t = f"transition{si + 1}"                         # :327-366
        for i, c in enumerate(cur):
            if i < len(pre):
                if c != pre[i]:
                    _conv(out, f"{t}.{i}.0", c, pre[i], 3); _bn(out, f"{t}.{i}.1", c)
            else:
                for j in range(i + 1 - len(pre)):
                    oc = c if j == i - len(pre) else pre[-1]
                    _conv(out, f"{t}.{i}.{j}.0", oc, pre[-1], 3); _bn(out, f"{t}.{i}.{j}.1", oc)
        nb = sc["NUM_BRANCHES"]
        for m in range(sc["NUM_MODULES"]):                # :387-417, HighResolutionModule :95-236
            multi = not (s == 4 and m == sc["NUM_MODULES"] - 1)
            base = f"stage{s}.{m}"
            for b in range(nb):
                for k in range(sc["NUM_BLOCKS"][b]):
                    p = f"{base}.branches.{b}.{k}"
                    _conv(out, p + ".conv1", cur[b], cur[b], 3); _bn(out, p + ".bn1", cur[b])
                    _conv(out, p + ".conv2", cur[b], cur[b], 3); _bn(out, p + ".bn2", cur[b])
            for i in range(nb if multi else 1):
                for j in range(nb):
                    f = f"{base}.fuse_layers.{i}.{j}"
                    if j > i:
                        _conv(out, f + ".0", cur[i], cur[j], 1); _bn(out, f + ".1", cur[i])
                    elif j < i:
                        for k in range(i - j):
                            oc = cur[i] if k == i - j - 1 else cur[j]
                            _conv(out, f"{f}.{k}.0", oc, cur[j], 3); _bn(out, f"{f}.{k}.1", oc)
        pre = cur
    kf = extra["FINAL_CONV_KERNEL"]
    _conv(out, "final_layer", joint_num, pre[0], kf, bias=True)
    return out


class PoseHighResolutionNet(ParamTree):
    def __init__(self, cfg: dict, joint_num: int = 17):
        super().__init__(hrnet_state_dict_shapes(cfg, joint_num))
        self.cfg = cfg
        self.joint_num = joint_num
        self.compute_dtype = "fp32"   # "bf16": bf16 activations/weights with fp32 accumulation (master weights stay fp32)
        self.autotune = True
        self._program: Optional[engine.Program] = None
        self._program_key = None
        # reference default init when `pretrained` is None: torch defaults (init_weights is not called, :494-495);
        # we reproduce the *shapes*; values come from load_state_dict.  Give BN affine the torch default (1, 0).
        for k, v in self.named_parameters():
            if k.endswith(".weight") and v.dim() == 1:
                torch.nn.init.ones_(v)
            elif v.dim() == 4:
                torch.nn.init.normal_(v, std=0.001)

    HEAD = "hrnet"         # which network walk simple_pose_amd.train.PoseTrainer takes
    fuse_stem = True       # bf16: conv1 + bn1 + relu + conv2 + bn2 + relu as one launch on the fp32 NCHW image (sp_hrnet_stem); same bits
    fuse_terms = True      # the identity / upsampled terms of a fuse output in one launch (sp_upsample_add_n_nhwc); fp32: same bits as the chain
    fuse_transition = True # bf16: transition1's two 3x3 convs on layer1's output as one launch (sp_hrnet_transition1); agrees with the two launches to fp32 summation order
    fuse_tail = True       # bf16: layer1.0's conv3 + projection shortcut as one launch (sp_dual_pw_bf16); same bits
    fuse_bottlenecks = True  # bf16: layer1.1-1.3 as one launch each (sp_bottleneck_c64, eight-wave kernel); same bits, +2.3 % (round 6)
    fuse_blocks = True     # bf16: the 32 BasicBlocks of the 32-channel branch as one launch each (sp_basic_block_c32, eight-wave strip kernel); same bits, +6.5 % (round 6)

    fuse_blocks64 = False  # True: the 64-channel branch's BasicBlocks too (sp_basic_block_c64); same bits, pays at small batch only (profiles/r06_bb64_ab.txt)

    def hip_program(self, x: torch.Tensor) -> engine.Program:
        sd = self.state_dict(keep_vars=True)
        key = (tuple(x.shape[2:]), str(x.device), self.compute_dtype, self.fuse_blocks, self.fuse_blocks64, self.fuse_stem, self.fuse_terms, self.fuse_transition, self.fuse_tail, self.fuse_bottlenecks) + tuple((v.data_ptr(), v._version) for v in sd.values())
        if self._program is None or key != self._program_key:
            for k, v in sd.items():
                if v.device != x.device:
                    raise HipLibraryError(f"parameter {k} is on {v.device} but the input is on {x.device}; call .to(device)")
            self._program = engine.hrnet_program({k: v.detach() for k, v in sd.items()}, self.cfg, x.shape[2], x.shape[3],
                                                  dtype=self.compute_dtype, fuse_blocks=self.fuse_blocks, fuse_blocks64=self.fuse_blocks64, fuse_stem=self.fuse_stem, fuse_terms=self.fuse_terms, fuse_transition=self.fuse_transition, fuse_tail=self.fuse_tail, fuse_bottlenecks=self.fuse_bottlenecks)
            self._program_key = key
        return self._program

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = require_cuda_f32(x, "input")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H,W multiples of 32, got {tuple(x.shape)}")
        if self.training:                       # one autograd node over the HIP tape, as the ResNets (PoseTrainer lowers pose_hrnet.py:419-454)
            from ._resnet_base import train_forward
            return train_forward(self, x)
        prog = self.hip_program(x)
        if self.autotune and x.shape[0] >= 16 and x.shape[0] >= 4 * prog.tuned_for_batch:
            prog.autotune(x)
        return prog.run(x)

    def forward_crops(self, crops: torch.Tensor) -> torch.Tensor:
        from ._resnet_base import forward_uint8_crops
        return forward_uint8_crops(self, crops)


def load_cfg(cfg_path: str) -> dict:
    with open(cfg_path, "r") as fh:
        return yaml.safe_load(fh)


def get_pose_net(cfg_path: str, pretrained: Optional[str] = None, joint_num: int = 17) -> PoseHighResolutionNet:
    """Same call shape as the reference (pose_hrnet.py:489-496).  `pretrained`: path of a local checkpoint whose keys
    are loaded non-strictly, like the reference's init_weights (:456-486); None = no weights loaded."""
    model = PoseHighResolutionNet(load_cfg(cfg_path), joint_num=joint_num)
    if pretrained:
        if not os.path.isfile(pretrained):
            raise ValueError(f"{pretrained} is not exist!")
        model.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
    return model
