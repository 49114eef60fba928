"""CPU oracle for the reference's named-but-unwired blocks -- TEST
INFRASTRUCTURE, see oracle/__init__.py.  Plain torch fp32 ops; each item cites the reference
file:line (relative to /root/reference/scripts) it restates.  Pinned by tests/golden/unwired.npz
(generated from the imported reference modules by oracle/gen_golden.py --only unwired)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- A9: sync-BN arithmetic
def sync_bn_forward(x_parts, weight, bias, eps=1e-5):
    """batchnorm.py:50-80,115-127 for W replicas holding `x_parts`: per-replica sum / sum of squares,
    summed over replicas; mean = sum/n; biased var = (ssum - sum*mean)/n; inv_std = clamp(var, eps)^-1/2;
    y = (x-mean)*(inv_std*w)+b.  Returns (outputs per replica, mean, unbiased var)."""
    c = x_parts[0].shape[1]
    n = sum(p.numel() // c for p in x_parts)
    s = sum(p.transpose(0, 1).reshape(c, -1).sum(1) for p in x_parts)
    ss = sum((p.transpose(0, 1).reshape(c, -1) ** 2).sum(1) for p in x_parts)
    mean = s / n
    sumvar = ss - s * mean
    inv_std = (sumvar / n).clamp(eps) ** -0.5
    outs = [(p - mean.view(1, c, 1, 1)) * (inv_std * weight).view(1, c, 1, 1) + bias.view(1, c, 1, 1) for p in x_parts]
    return outs, mean, sumvar / (n - 1)


# ----------------------------------------------------------------------------- A13: up_conv
class UpConvCPU(nn.Module):
    """archs.py:848-860: nearest x2 -> conv3x3 (+bias) -> BN -> ReLU."""

    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(ch_in, ch_out, 3, 1, 1, bias=True),
                                nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.up(x)


# ----------------------------------------------------------------------------- A11: xResidualBlock
class ModulecellCPU(nn.Module):
    """xresidualblock.py:9-24."""

    def __init__(self, cin, cout, k=3, sk=9):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(cin, cout, k, padding=(k - 1) // 2, bias=True))
        self.module = nn.Sequential(nn.BatchNorm2d(cout), nn.ReLU(),
                                    nn.Conv2d(cout, cout, sk, 1, (sk - 1) // 2, groups=cout), nn.BatchNorm2d(cout))

    def forward(self, x):
        x1 = self.features(x)
        g = self.module(x1)
        return x1 * torch.exp(-g * g)                                  # Gaussian, :5-7


class XResidualBlockCPU(nn.Module):
    """xresidualblock.py:26-34: bn1(conv2(Modulecell(x))) + x."""

    def __init__(self, cin=64, planes=64, k=3, s=1):
        super().__init__()
        self.md = ModulecellCPU(cin, planes, k)
        self.conv2 = nn.Conv2d(planes, planes, k, stride=s, padding=1)
        self.bn1 = nn.BatchNorm2d(planes)

    def forward(self, x):
        return self.bn1(self.conv2(self.md(x))) + x


# ----------------------------------------------------------------------------- A12: spectral norm
def spectral_norm_step(w, u, v, n_iter=1, eps=1e-12):
    """spectral_norm.py:38-88: returns (w/sigma, u', v', sigma) after n_iter power iterations."""
    wm = w.reshape(w.shape[0], -1)
    for _ in range(n_iter):
        v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
        u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma, u, v, sigma


# ----------------------------------------------------------------------------- A10: MBConv / EfficientNet features
def same_pad(size, k, s):
    out = math.ceil(size / s)
    p = max((out - 1) * s + (k - 1) + 1 - size, 0)
    return p // 2, p - p // 2


class SameConvCPU(nn.Conv2d):
    """efficientnet_pytorch/utils.py:123-146 (static TF-"same" padding from image_size)."""

    def __init__(self, cin, cout, k, image_size, **kw):
        super().__init__(cin, cout, k, **kw)
        s = self.stride[0]
        pt, pb = same_pad(image_size, k, s)
        self.pads = (pt, pb, pt, pb)                                   # square images/kernels: (l, r, t, b)

    def forward(self, x):
        return F.conv2d(F.pad(x, [self.pads[0], self.pads[1], self.pads[2], self.pads[3]]), self.weight, self.bias,
                        self.stride, 0, 1, self.groups)


def swish(x):
    return x * torch.sigmoid(x)                                        # utils.py:37-56


class MBConvCPU(nn.Module):
    """efficientnet_pytorch/model.py:18-99 (drop-connect off: it is RNG-dependent)."""

    def __init__(self, k, s, inp, out, expand, se_ratio, image_size, bn_mom=0.01, bn_eps=1e-3, stride_literal=None):
        super().__init__()
        stride_literal = s if stride_literal is None else stride_literal
        oup = inp * expand
        # model.py:93-94 tests `stride == 1` literally; a decoded BlockArgs stride is the list [1], which is != 1
        self.expand, self.skip = expand, (stride_literal == 1 and inp == out)
        if expand != 1:
            self._expand_conv = SameConvCPU(inp, oup, 1, image_size, bias=False)
            self._bn0 = nn.BatchNorm2d(oup, momentum=bn_mom, eps=bn_eps)
        self._depthwise_conv = SameConvCPU(oup, oup, k, image_size, groups=oup, stride=s, bias=False)
        self._bn1 = nn.BatchNorm2d(oup, momentum=bn_mom, eps=bn_eps)
        nsq = max(1, int(inp * se_ratio))
        self._se_reduce = SameConvCPU(oup, nsq, 1, image_size)
        self._se_expand = SameConvCPU(nsq, oup, 1, image_size)
        self._project_conv = SameConvCPU(oup, out, 1, image_size, bias=False)
        self._bn2 = nn.BatchNorm2d(out, momentum=bn_mom, eps=bn_eps)

    def forward(self, x):
        inp = x
        if self.expand != 1:
            x = swish(self._bn0(self._expand_conv(x)))
        x = swish(self._bn1(self._depthwise_conv(x)))
        sq = self._se_expand(swish(self._se_reduce(F.adaptive_avg_pool2d(x, 1))))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        return x + inp if self.skip else x
