"""ResNet-50 + DUC head (PixelShuffle -> DUC(512,1024) -> DUC(256,512) -> conv3x3+bias), MI355X-native.

Drop-in for the reference's `nets/pose_resnet_duc.py` (`_make_duc_layer` :227-232, `final_layer` :172-177): same
factory, same 332 state_dict keys (`duc_layers.{1,2}.{conv.weight,bn.*}`, `final_layer.weight [J,128,3,3]`).
"""
from __future__ import annotations

from torch import nn

from ._resnet_base import PoseResNetBase, load_pretrained_like_reference
from .commons import DUC

__all__ = ["ResNet", "resnet50"]


class ResNet(PoseResNetBase):
    HEAD = "duc"

    def _build_head(self, inplanes: int, num_classes: int):
        self.duc_layers = nn.Sequential(nn.PixelShuffle(2), DUC(inplanes // 4, 1024), DUC(256, 512))
        self.final_layer = nn.Conv2d(128, num_classes, kernel_size=3, padding=1)


def resnet50(pretrained: bool = False, progress: bool = True, **kwargs) -> ResNet:
    model = ResNet(num_classes=kwargs.pop("num_classes", 1000), reduction=kwargs.pop("reduction", False))
    if kwargs:
        raise TypeError(f"unsupported arguments for the HIP ResNet-50: {sorted(kwargs)}")
    if pretrained:
        load_pretrained_like_reference(model, "resnet50")
    return model
