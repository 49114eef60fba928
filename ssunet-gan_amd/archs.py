"""Segmentation generators, MI355X-native (mirrors the reference's scripts/archs.py).

`archs.__dict__[name](num_classes, input_channels, deep_supervision)` is the constructor
surface models_seg_gan.Generator uses (models_seg_gan.py:212-214).  Module/parameter names
and creation order follow the reference so seeds and checkpoints interchange; every forward
runs on the hand-written HIP kernels behind include/ssunet_hip.h."""
import torch
import torch.nn as nn
from torch.nn import init

from . import ops
from .blocks import basic_block
from .normalization import SPADE
from ._lib import ACT_NONE, ACT_RELU

# archs.py:8 -- the reference's export list.  UNet_R_SS_v2 is the arch config_v1.json wires.
__all__ = ['UNet_R_SS_v2']


def _sync_group(bn):
    return getattr(bn, '_ssg_sync_group', None)


class BasicBlock(nn.Module):
    """archs.py:205-241."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False))

    def forward(self, x, x2=None):
        """`x2` (optional) is a second tensor concatenated after `x` along channels: the
        decoder's torch.cat([enc, up], 1) (archs.py:651-667) without materialising it."""
        sc = self.shortcut[0] if len(self.shortcut) else None
        if self.training:
            return basic_block(x, x2, self.conv1, self.bn1, self.conv2, self.bn2, sc, group=_sync_group(self.bn1))
        s = self.conv1.stride[0]
        y = ops.conv2d(x, self.conv1.weight, None, s, 1, x2=x2)
        y = ops.batch_norm_act(y, self.bn1, act=ACT_RELU)
        y = ops.conv2d(y, self.conv2.weight, None, 1, 1)
        r = ops.conv2d(x, sc.weight, None, s, 0, x2=x2) if sc is not None else x
        return ops.batch_norm_act(y, self.bn2, res=r, act=ACT_RELU)


class conv_block(nn.Module):
    """archs.py:831-846: (conv3x3 + bias -> BN -> ReLU) x 2."""

    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True),
            nn.Conv2d(ch_out, ch_out, kernel_size=3, stride=1, padding=1, bias=True), nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True))

    def forward(self, x, x2=None):
        c = self.conv
        y = ops.conv2d(x, c[0].weight, c[0].bias, 1, 1, x2=x2)
        y = ops.batch_norm_act(y, c[1], act=ACT_RELU, group=_sync_group(c[1]))
        y = ops.conv2d(y, c[3].weight, c[3].bias, 1, 1)
        return ops.batch_norm_act(y, c[4], act=ACT_RELU, group=_sync_group(c[4]))


class up_conv(nn.Module):
    """archs.py:848-860: nearest x2 -> conv3x3 + bias -> BN -> ReLU."""

    def __init__(self, ch_in, ch_out):
        super().__init__()
        self.up = nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(ch_in, ch_out, kernel_size=3, stride=1, padding=1, bias=True),
                                nn.BatchNorm2d(ch_out), nn.ReLU(inplace=True))

    def forward(self, x):
        y = ops.upsample2x_nearest(x)
        y = ops.conv2d(y, self.up[1].weight, self.up[1].bias, 1, 1)
        return ops.batch_norm_act(y, self.up[2], act=ACT_RELU, group=_sync_group(self.up[2]))


class UNet_R_SS_v2(nn.Module):
    """archs.py:559-671: six-level residual U-Net with self-conditioned SPADE after every block,
    max-unpooling with the encoder's indices for the three deepest decoder stages and bilinear
    (align_corners) upsampling for the last two."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb_filter = [64, 128, 256, 384, 512, 768]
        spade_mid = num_classes
        # kept for state/attribute parity (archs.py:571-573); the forward uses the fused HIP ops
        self.pool = nn.MaxPool2d(2, 2, return_indices=True)
        self.unpool = nn.MaxUnpool2d(2, stride=2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        context = 'spadebatch3x3'
        ss_scale = 16
        self.conv0_0 = BasicBlock(input_channels, nb_filter[0])
        self.SPADE0_0 = SPADE(context, nb_filter[0], spade_mid, nb_filter[0] / ss_scale)
        self.conv1_0 = BasicBlock(nb_filter[0], nb_filter[1])
        self.SPADE1_0 = SPADE(context, nb_filter[1], spade_mid, nb_filter[1] / ss_scale)
        self.conv2_0 = BasicBlock(nb_filter[1], nb_filter[2])
        self.SPADE2_0 = SPADE(context, nb_filter[2], spade_mid, nb_filter[2] / ss_scale)
        self.conv3_0 = BasicBlock(nb_filter[2], nb_filter[3])
        self.SPADE3_0 = SPADE(context, nb_filter[3], spade_mid, nb_filter[3] / ss_scale)
        self.conv4_0 = BasicBlock(nb_filter[3], nb_filter[4])
        self.SPADE4_0 = SPADE(context, nb_filter[4], spade_mid, nb_filter[4] / ss_scale)
        self.conv5_0 = BasicBlock(nb_filter[4], nb_filter[5])
        self.SPADE5_0 = SPADE(context, nb_filter[5], spade_mid, nb_filter[5] / ss_scale)
        self.conv_head5_0 = nn.Conv2d(nb_filter[5], nb_filter[4], kernel_size=1, stride=1, bias=False)
        self.conv4_1 = BasicBlock(nb_filter[4] + nb_filter[4], nb_filter[4])
        self.SPADE4_1 = SPADE(context, nb_filter[4], spade_mid, nb_filter[4] / ss_scale)
        self.conv_head4_1 = nn.Conv2d(nb_filter[4], nb_filter[3], kernel_size=1, stride=1, bias=False)
        self.conv3_1 = BasicBlock(nb_filter[3] + nb_filter[3], nb_filter[3])
        self.SPADE3_1 = SPADE(context, nb_filter[3], spade_mid, nb_filter[3] / ss_scale)
        self.conv_head3_1 = nn.Conv2d(nb_filter[3], nb_filter[2], kernel_size=1, stride=1, bias=False)
        self.conv2_1 = BasicBlock(nb_filter[2] + nb_filter[2], nb_filter[2])
        self.SPADE2_1 = SPADE(context, nb_filter[2], spade_mid, nb_filter[2] / ss_scale)
        self.conv1_1 = BasicBlock(nb_filter[1] + nb_filter[2], nb_filter[1])
        self.SPADE1_1 = SPADE(context, nb_filter[1], spade_mid, nb_filter[1] / ss_scale)
        self.conv0_1 = BasicBlock(nb_filter[0] + nb_filter[1], nb_filter[0])
        self.SPADE0_1 = SPADE(context, nb_filter[0], spade_mid, nb_filter[0] / ss_scale)
        self.final = nn.Conv2d(nb_filter[0], num_classes, kernel_size=1)
        self.init_weights()

    def init_weights(self):
        init.kaiming_uniform_(self.final.weight, mode='fan_in')
        self.final.bias.data.fill_(0)

    def forward(self, input):
        x = ops.as_nhwc(input)
        enc_0 = self.SPADE0_0(self.conv0_0(x))
        p0, _ = ops.max_pool2x2(enc_0)
        enc_1 = self.SPADE1_0(self.conv1_0(p0))
        p1, _ = ops.max_pool2x2(enc_1)
        enc_2 = self.SPADE2_0(self.conv2_0(p1))
        p2, i2 = ops.max_pool2x2(enc_2)
        enc_3 = self.SPADE3_0(self.conv3_0(p2))
        p3, i3 = ops.max_pool2x2(enc_3)
        enc_4 = self.SPADE4_0(self.conv4_0(p3))
        p4, i4 = ops.max_pool2x2(enc_4)
        enc_5 = self.SPADE5_0(self.conv5_0(p4))
        enc_5 = ops.conv2d(enc_5, self.conv_head5_0.weight)
        dec_4 = self.SPADE4_1(self.conv4_1(enc_4, ops.max_unpool2x2(enc_5, i4)))
        dec_4 = ops.conv2d(dec_4, self.conv_head4_1.weight)
        dec_3 = self.SPADE3_1(self.conv3_1(enc_3, ops.max_unpool2x2(dec_4, i3)))
        dec_3 = ops.conv2d(dec_3, self.conv_head3_1.weight)
        dec_2 = self.SPADE2_1(self.conv2_1(enc_2, ops.max_unpool2x2(dec_3, i2)))
        dec_1 = self.SPADE1_1(self.conv1_1(enc_1, ops.upsample2x_bilinear(dec_2)))
        dec_0 = self.SPADE0_1(self.conv0_1(enc_0, ops.upsample2x_bilinear(dec_1)))
        return ops.conv2d(dec_0, self.final.weight, self.final.bias)
