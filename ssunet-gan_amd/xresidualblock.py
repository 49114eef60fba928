"""xResidualBlock, MI355X-native (mirrors the reference's scripts/xresidualblock.py:5-34; unwired in
the reference -- SURVEY.md 8a row A11 -- and built as a per-op block with its own golden vectors).
Constructors, module names and parameter order are the reference's."""
import torch.nn as nn

from . import ops
from ._lib import ACT_RELU


class Gaussian(nn.Module):
    def forward(self, input):
        return ops.gaussian(input)


class Modulecell(nn.Module):
    """conv kxk (+bias) -> x1;  gate = Gaussian(BN(dwconv9x9(ReLU(BN(x1)))));  out = x1 * gate."""

    def __init__(self, in_channels=1, out_channels=64, kernel_size=3, skernel_size=9):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=((kernel_size - 1) // 2), bias=True))
        self.module = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(),
            nn.Conv2d(out_channels, out_channels, kernel_size=skernel_size, stride=1, padding=((skernel_size - 1) // 2),
                      groups=out_channels),
            nn.BatchNorm2d(out_channels),
            Gaussian())

    def forward(self, x):
        f = self.features[0]
        x1 = ops.conv2d(x, f.weight, f.bias, 1, f.padding[0])
        g = ops.batch_norm_act(x1, self.module[0], act=ACT_RELU)
        dw = self.module[2]
        g = ops.dwconv2d(g, dw.weight, dw.bias, 1, dw.padding[0])
        g = ops.gaussian(ops.batch_norm_act(g, self.module[3]))
        return ops.mul(x1, g)


class xResidualBlock(nn.Module):
    def __init__(self, in_channels=64, planes=64, kernel_size=3, s=1):
        super().__init__()
        self.md = Modulecell(in_channels, planes, kernel_size)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size, stride=s, padding=1)
        self.bn1 = nn.BatchNorm2d(planes)

    def forward(self, x):
        y = self.md(x)
        y = ops.conv2d(y, self.conv2.weight, self.conv2.bias, self.conv2.stride[0], self.conv2.padding[0])
        return ops.batch_norm_act(y, self.bn1, res=ops.as_nhwc(x))          # bn1(conv2(y)) + x
