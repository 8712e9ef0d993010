"""ResNet-50 + DUC head (PixelShuffle -> DUC(512,1024) -> DUC(256,512) -> conv3x3+bias), MI355X-native.

Drop-in for the reference's `nets/pose_resnet_duc.py` (`_make_duc_layer` :227-232, `final_layer` :172-177): same
factory, same 332 state_dict keys (`duc_layers.{1,2}.{conv.weight,bn.*}`, `final_layer.weight [J,128,3,3]`).
"""
from __future__ import annotations

from torch import nn

from ._resnet_base import PoseResNetBase, load_pretrained_like_reference
from .commons import DUC

__all__ = ["ResNet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152", "wide_resnet50_2", "wide_resnet101_2",
           "resnext50_32x4d", "resnext101_32x8d"]


class ResNet(PoseResNetBase):
    HEAD = "duc"

    def _build_head(self, inplanes: int, num_classes: int):
        self.duc_layers = nn.Sequential(nn.PixelShuffle(2), DUC(inplanes // 4, 1024), DUC(256, 512))
        self.final_layer = nn.Conv2d(128, num_classes, kernel_size=3, padding=1)


def _resnet(arch: str, blocks, pretrained: bool, kwargs, block: str = "bottleneck", width_per_group: int = 64, groups: int = 1) -> ResNet:
    model = ResNet(num_classes=kwargs.pop("num_classes", 1000), reduction=kwargs.pop("reduction", False), blocks=blocks, block=block,
                   width_per_group=width_per_group, groups=groups)
    if kwargs:
        raise TypeError(f"unsupported arguments for the HIP {arch}: {sorted(kwargs)}")
    if pretrained:
        load_pretrained_like_reference(model, arch)
    return model


def resnet50(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Same call shape as the reference factory: resnet50(pretrained=..., num_classes=J, reduction=bool)."""
    return _resnet("resnet50", (3, 4, 6, 3), pretrained, kwargs)


def resnet101(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottleneck depths [3, 4, 23, 3] (reference factory of the same name, pose_resnet_duc.py)."""
    return _resnet("resnet101", (3, 4, 23, 3), pretrained, kwargs)


def resnet152(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottleneck depths [3, 8, 36, 3] (reference factory of the same name, pose_resnet_duc.py)."""
    return _resnet("resnet152", (3, 8, 36, 3), pretrained, kwargs)


def resnet18(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """BasicBlock depths [2, 2, 2, 2] (reference factory of the same name, pose_resnet_duc.py:265-386); the head starts from 512 channels."""
    return _resnet("resnet18", (2, 2, 2, 2), pretrained, kwargs, block="basic")


def resnet34(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """BasicBlock depths [3, 4, 6, 3] (reference factory of the same name)."""
    return _resnet("resnet34", (3, 4, 6, 3), pretrained, kwargs, block="basic")


def wide_resnet50_2(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottlenecks [3, 4, 6, 3] with twice the inner width (`width_per_group = 128`; reference factory of the same name)."""
    return _resnet("wide_resnet50_2", (3, 4, 6, 3), pretrained, kwargs, width_per_group=128)


def wide_resnet101_2(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottlenecks [3, 4, 23, 3] with twice the inner width (reference factory of the same name)."""
    return _resnet("wide_resnet101_2", (3, 4, 23, 3), pretrained, kwargs, width_per_group=128)


def resnext50_32x4d(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottlenecks [3, 4, 6, 3] whose 3x3 conv has 32 groups of 4 x (planes / 64) channels (reference factory of the same name, pose_resnet_duc.py;
    `groups = 32`, `width_per_group = 4`).  Eval-mode forward on the HIP kernels (grouped implicit GEMM: sp_conv_desc.c_in_group); training is
    refused by PoseTrainer with a message."""
    return _resnet("resnext50_32x4d", (3, 4, 6, 3), pretrained, kwargs, width_per_group=4, groups=32)


def resnext101_32x8d(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottlenecks [3, 4, 23, 3], 32 groups of 8 x (planes / 64) channels (reference factory of the same name)."""
    return _resnet("resnext101_32x8d", (3, 4, 23, 3), pretrained, kwargs, width_per_group=8, groups=32)
