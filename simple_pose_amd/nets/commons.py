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
