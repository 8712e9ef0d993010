"""Parameter holder for the reference's DUC block (`nets/commons.py:21-43`): conv3x3 (no bias) -> BN -> ReLU ->
PixelShuffle(r).  Compute happens in the HIP conv kernel with the shuffle fused into its store."""
from torch import nn


class DUC(nn.Module):
    def __init__(self, inplanes: int, planes: int, upscale_factor: int = 2):
        super().__init__()
        if upscale_factor != 2:
            raise NotImplementedError("only PixelShuffle(2) is lowered")
        self.conv = nn.Conv2d(inplanes, planes, kernel_size=3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)


class SELayer(nn.Module):
    """Parameter holder for the reference's SELayer (`nets/commons.py:4-18`, reduction=1: C -> C -> C):
    keys fc.0.{weight [C,C,1,1], bias}, fc.2.{weight, bias}."""

    def __init__(self, channel: int, reduction: int = 1):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1, 1, 0), nn.ReLU(inplace=True),
                                nn.Conv2d(channel // reduction, channel, 1, 1, 0), nn.Sigmoid())
