"""ResNet-50 + 3 x [ConvTranspose2d(4,2,1) -> BN -> ReLU] + 1x1 head, MI355X-native.

Drop-in for the reference's `nets/pose_resnet_dconv.py`: `resnet50(pretrained, num_classes, reduction)` (:306-315)
returns a module with the reference's state_dict keys/shapes (338 entries; `deconv_layers.{0,3,6}.weight` are
[Cin,Cout,4,4], `deconv_layers.{1,4,7}.*` BN, `final_layer.{weight,bias}`), whose `forward` runs entirely in
libsimple_pose_hip.so.
"""
from __future__ import annotations

from torch import nn

from ._resnet_base import PoseResNetBase, load_pretrained_like_reference

__all__ = ["ResNet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152", "wide_resnet50_2", "wide_resnet101_2",
           "resnext50_32x4d", "resnext101_32x8d"]


class ResNet(PoseResNetBase):
    HEAD = "dconv"

    def _build_head(self, inplanes: int, num_classes: int):
        layers = []
        for _ in range(3):  # pose_resnet_dconv.py:230-249
            layers += [nn.ConvTranspose2d(inplanes, 256, 4, stride=2, padding=1, output_padding=0, bias=False),
                       nn.BatchNorm2d(256), nn.ReLU(inplace=True)]
            inplanes = 256
        self.deconv_layers = nn.Sequential(*layers)
        self.final_layer = nn.Conv2d(256, num_classes, 1)  # :173-178


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
    """Bottleneck depths [3, 4, 23, 3] (reference factory of the same name, pose_resnet_dconv.py:306-339)."""
    return _resnet("resnet101", (3, 4, 23, 3), pretrained, kwargs)


def resnet152(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottleneck depths [3, 8, 36, 3] (reference factory of the same name, pose_resnet_dconv.py:306-339)."""
    return _resnet("resnet152", (3, 8, 36, 3), pretrained, kwargs)


def resnet18(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """BasicBlock depths [2, 2, 2, 2] (reference factory of the same name, pose_resnet_dconv.py:282-403); the head starts from 512 channels."""
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
    """Bottlenecks [3, 4, 6, 3] whose 3x3 conv has 32 groups of 4 x (planes / 64) channels (reference factory of the same name, pose_resnet_dconv.py:342-368;
    `groups = 32`, `width_per_group = 4`).  Eval-mode forward on the HIP kernels (grouped implicit GEMM: sp_conv_desc.c_in_group); training is
    refused by PoseTrainer with a message."""
    return _resnet("resnext50_32x4d", (3, 4, 6, 3), pretrained, kwargs, width_per_group=4, groups=32)


def resnext101_32x8d(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    """Bottlenecks [3, 4, 23, 3], 32 groups of 8 x (planes / 64) channels (reference factory of the same name)."""
    return _resnet("resnext101_32x8d", (3, 4, 23, 3), pretrained, kwargs, width_per_group=8, groups=32)
