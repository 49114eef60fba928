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
__all__ = ['UNet', 'NestedUNet', 'SSUNet', 'UNet_ori', 'UNet_B_SS', 'AttUNet', 'UNet_R_SS', 'UNet_R_SS_v2']


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
        # eval mode: the running-stat batch norms are folded into the conv weights, so the block is three
        # MFMA launches with bias / residual / ReLU epilogues and no batch-norm pass (sliding-window
        # inference, aerial_image_segmentation_api.py:376-390)
        s = self.conv1.stride[0]
        w1, b1, w2, b2 = self._folded()
        y = ops.conv2d(x, w1, b1, s, 1, act=ACT_RELU, x2=x2)
        r = ops.conv2d(x, sc.weight, None, s, 0, x2=x2) if sc is not None else x
        return ops.conv2d(y, w2, b2, 1, 1, act=ACT_RELU, res=r)

    def _folded(self):
        """(w1', b1', w2', b2') with w' = w * gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps);
        cached until a parameter or running statistic changes."""
        srcs = (self.conv1.weight, self.conv2.weight, self.bn1.weight, self.bn1.bias, self.bn1.running_mean, self.bn1.running_var,
                self.bn2.weight, self.bn2.bias, self.bn2.running_mean, self.bn2.running_var)
        stamp = tuple((t.data_ptr(), t._version) for t in srcs) + (ops._WEIGHT_EPOCH[0], ops._STATS_EPOCH[0])
        hit = getattr(self, '_fold_cache', None)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            out = []
            for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2)):
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                out += [(conv.weight * scale.view(-1, 1, 1, 1)).contiguous(), (bn.bias - bn.running_mean * scale).contiguous()]
        self._fold_cache = (stamp, tuple(out))
        return self._fold_cache[1]


def _cat_split(xs):
    """(x, x2) for a conv that consumes torch.cat(xs, 1): the last tensor rides the second input pointer,
    the others are materialised only if there are more than two."""
    xs = list(xs)
    if len(xs) == 1:
        return xs[0], None
    return ops.concat_channels(*xs[:-1]), xs[-1]


def _pool(x):
    return ops.max_pool2x2(x)[0]


class VGGBlock(nn.Module):
    """archs.py:92-111: (conv3x3 + bias -> BN -> ReLU) x 2."""

    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_channels, middle_channels, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(middle_channels)
        self.conv2 = nn.Conv2d(middle_channels, out_channels, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(out_channels)

    def forward(self, x, x2=None):
        y = ops.conv2d(x, self.conv1.weight, self.conv1.bias, 1, 1, x2=x2)
        y = ops.batch_norm_act(y, self.bn1, act=ACT_RELU, group=_sync_group(self.bn1))
        y = ops.conv2d(y, self.conv2.weight, self.conv2.bias, 1, 1)
        return ops.batch_norm_act(y, self.bn2, act=ACT_RELU, group=_sync_group(self.bn2))


class Bottleneck(nn.Module):
    """archs.py:244-269: 1x1 -> 3x3 -> 1x1 with BN + ReLU, BN on the 1x1 shortcut."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, self.expansion * planes, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(self.expansion * planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False),
                                          nn.BatchNorm2d(self.expansion * planes))

    def forward(self, x, x2=None):
        s = self.conv2.stride[0]
        y = ops.batch_norm_act(ops.conv2d(x, self.conv1.weight, x2=x2), self.bn1, act=ACT_RELU, group=_sync_group(self.bn1))
        y = ops.batch_norm_act(ops.conv2d(y, self.conv2.weight, None, s, 1), self.bn2, act=ACT_RELU, group=_sync_group(self.bn2))
        y = ops.conv2d(y, self.conv3.weight)
        if len(self.shortcut):
            r = ops.batch_norm_act(ops.conv2d(x, self.shortcut[0].weight, None, s, 0, x2=x2), self.shortcut[1],
                                   group=_sync_group(self.shortcut[1]))
        else:
            r = x
        return ops.batch_norm_act(y, self.bn3, res=r, act=ACT_RELU, group=_sync_group(self.bn3))


class SubPixelConvolutionalBlock(nn.Module):
    """archs.py:145-175.  UNet_R_SS constructs one (`sp_up1_3`, archs.py:515) and never calls it: it exists
    here so seeds and state_dicts line up; it has no HIP forward."""

    def __init__(self, kernel_size=3, n_channels=64, scaling_factor=2):
        super().__init__()
        self.conv = nn.Conv2d(in_channels=n_channels, out_channels=n_channels * (scaling_factor ** 2), kernel_size=kernel_size,
                              padding=kernel_size // 2)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor=scaling_factor)
        self.prelu = nn.PReLU()

    def forward(self, input):
        raise NotImplementedError('SubPixelConvolutionalBlock is never called on the hot path (archs.py:515 is dead code)')


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


class Attention_block(nn.Module):
    """archs.py:115-144: additive attention gate.  psi = sigmoid(BN1(conv1x1(relu(BN(W_g g) + BN(W_x x))))), out = x * psi.
    The ReLU rides the second batch norm's residual epilogue; sigmoid and the per-pixel broadcast are one kernel."""

    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv2d(F_g, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(F_l, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)

    def forward(self, g, x):
        g1 = ops.batch_norm_act(ops.conv2d(g, self.W_g[0].weight, self.W_g[0].bias), self.W_g[1], group=_sync_group(self.W_g[1]))
        a = ops.batch_norm_act(ops.conv2d(x, self.W_x[0].weight, self.W_x[0].bias), self.W_x[1], res=g1, act=ACT_RELU,
                               group=_sync_group(self.W_x[1]))
        p = ops.conv2d(a, self.psi[0].weight, self.psi[0].bias)
        p = ops.batch_norm_act(p, self.psi[1], group=_sync_group(self.psi[1]))
        return ops.pixel_gate(x, p)


class AttUNet(nn.Module):
    """archs.py:271-345: UNet_ori with an attention gate on every skip connection."""

    def __init__(self, output_ch, img_ch=3, deep_supervision=False, **kwargs):
        super().__init__()
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Conv1 = conv_block(ch_in=img_ch, ch_out=64)
        self.Conv2 = conv_block(ch_in=64, ch_out=128)
        self.Conv3 = conv_block(ch_in=128, ch_out=256)
        self.Conv4 = conv_block(ch_in=256, ch_out=512)
        self.Conv5 = conv_block(ch_in=512, ch_out=1024)
        self.Up5 = up_conv(ch_in=1024, ch_out=512)
        self.Att5 = Attention_block(F_g=512, F_l=512, F_int=256)
        self.Up_conv5 = conv_block(ch_in=1024, ch_out=512)
        self.Up4 = up_conv(ch_in=512, ch_out=256)
        self.Att4 = Attention_block(F_g=256, F_l=256, F_int=128)
        self.Up_conv4 = conv_block(ch_in=512, ch_out=256)
        self.Up3 = up_conv(ch_in=256, ch_out=128)
        self.Att3 = Attention_block(F_g=128, F_l=128, F_int=64)
        self.Up_conv3 = conv_block(ch_in=256, ch_out=128)
        self.Up2 = up_conv(ch_in=128, ch_out=64)
        self.Att2 = Attention_block(F_g=64, F_l=64, F_int=32)
        self.Up_conv2 = conv_block(ch_in=128, ch_out=64)
        self.Conv_1x1 = nn.Conv2d(64, output_ch, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        x1 = self.Conv1(ops.as_nhwc(x))
        x2 = self.Conv2(_pool(x1))
        x3 = self.Conv3(_pool(x2))
        x4 = self.Conv4(_pool(x3))
        x5 = self.Conv5(_pool(x4))
        d5 = self.Up5(x5)
        d5 = self.Up_conv5(self.Att5(d5, x4), d5)          # cat((gated skip, d5)) is the conv's second input pointer
        d4 = self.Up4(d5)
        d4 = self.Up_conv4(self.Att4(d4, x3), d4)
        d3 = self.Up3(d4)
        d3 = self.Up_conv3(self.Att3(d3, x2), d3)
        d2 = self.Up2(d3)
        d2 = self.Up_conv2(self.Att2(d2, x1), d2)
        return ops.conv2d(d2, self.Conv_1x1.weight, self.Conv_1x1.bias)


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
        # every encoder output feeds the pool AND the decoder's concat: one node sums the two gradients in the pool backward
        def enc(block, t):                  # profiling label only: bench.py's `encoder_3x3` sub-metric (54.7 GMAC/img)
            with ops.role('encoder_3x3'):
                return block(t)
        p0, _, enc_0 = ops.max_pool2x2_skip(self.SPADE0_0(enc(self.conv0_0, x)))
        p1, _, enc_1 = ops.max_pool2x2_skip(self.SPADE1_0(enc(self.conv1_0, p0)))
        p2, i2, enc_2 = ops.max_pool2x2_skip(self.SPADE2_0(enc(self.conv2_0, p1)))
        p3, i3, enc_3 = ops.max_pool2x2_skip(self.SPADE3_0(enc(self.conv3_0, p2)))
        p4, i4, enc_4 = ops.max_pool2x2_skip(self.SPADE4_0(enc(self.conv4_0, p3)))
        enc_5 = self.SPADE5_0(enc(self.conv5_0, p4))
        enc_5 = ops.conv2d(enc_5, self.conv_head5_0.weight)
        dec_4 = self.SPADE4_1(self.conv4_1(enc_4, ops.max_unpool2x2(enc_5, i4)))
        dec_4 = ops.conv2d(dec_4, self.conv_head4_1.weight)
        dec_3 = self.SPADE3_1(self.conv3_1(enc_3, ops.max_unpool2x2(dec_4, i3)))
        dec_3 = ops.conv2d(dec_3, self.conv_head3_1.weight)
        dec_2 = self.SPADE2_1(self.conv2_1(enc_2, ops.max_unpool2x2(dec_3, i2)))
        dec_1 = self.SPADE1_1(self.conv1_1(enc_1, ops.upsample2x_bilinear(dec_2)))
        dec_0 = self.SPADE0_1(self.conv0_1(enc_0, ops.upsample2x_bilinear(dec_1)))
        return ops.conv2d(dec_0, self.final.weight, self.final.bias)


class _PlainUNetBase(nn.Module):
    """Shared forward of the 5-level encoder/decoder archs whose stages are `block(x[, x2])`, with an
    optional SPADE after every stage: UNet (archs.py:791-829), SSUNet (:673-743), UNet_B_SS (:347-407)."""

    def _stage(self, name, x, x2=None):
        y = getattr(self, 'conv' + name)(x, x2)
        sp = getattr(self, 'SPADE' + name, None)
        return sp(y, y) if sp is not None else y

    def forward(self, input):
        x = ops.as_nhwc(input)
        up = ops.upsample2x_bilinear
        x0_0 = self._stage('0_0', x)
        x1_0 = self._stage('1_0', _pool(x0_0))
        x2_0 = self._stage('2_0', _pool(x1_0))
        x3_0 = self._stage('3_0', _pool(x2_0))
        x4_0 = self._stage('4_0', _pool(x3_0))
        x3_1 = self._stage('3_1', x3_0, up(x4_0))
        x2_2 = self._stage('2_2', x2_0, up(x3_1))
        x1_3 = self._stage('1_3', x1_0, up(x2_2))
        x0_4 = self._stage('0_4', x0_0, up(x1_3))
        return ops.conv2d(x0_4, self.final.weight, self.final.bias)


class UNet(_PlainUNetBase):
    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [64, 128, 256, 512, 1024]
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv0_0 = VGGBlock(input_channels, nb[0], nb[0])
        self.conv1_0 = VGGBlock(nb[0], nb[1], nb[1])
        self.conv2_0 = VGGBlock(nb[1], nb[2], nb[2])
        self.conv3_0 = VGGBlock(nb[2], nb[3], nb[3])
        self.conv4_0 = VGGBlock(nb[3], nb[4], nb[4])
        self.conv3_1 = VGGBlock(nb[3] + nb[4], nb[3], nb[3])
        self.conv2_2 = VGGBlock(nb[2] + nb[3], nb[2], nb[2])
        self.conv1_3 = VGGBlock(nb[1] + nb[2], nb[1], nb[1])
        self.conv0_4 = VGGBlock(nb[0] + nb[1], nb[0], nb[0])
        self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)


class SSUNet(_PlainUNetBase):
    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [32, 64, 128, 256, 512]
        sm, ctx, sc = num_classes, 'spadebatch3x3', 4
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        chain = [('0_0', input_channels, 0), ('1_0', nb[0], 1), ('2_0', nb[1], 2), ('3_0', nb[2], 3), ('4_0', nb[3], 4),
                 ('3_1', nb[3] + nb[4], 3), ('2_2', nb[2] + nb[3], 2), ('1_3', nb[1] + nb[2], 1), ('0_4', nb[0] + nb[1], 0)]
        for name, cin, lvl in chain:                      # creation order of archs.py:690-715: conv, SPADE per stage
            setattr(self, 'conv' + name, VGGBlock(cin, nb[lvl], nb[lvl]))
            setattr(self, 'SPADE' + name, SPADE(ctx, nb[lvl], sm, nb[lvl] / sc))
        self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)


class UNet_B_SS(_PlainUNetBase):
    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [64, 128, 256, 512, 1024]
        sm, ctx, sc = num_classes, 'spadebatch3x3', 16
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        for name, lvl in (('0_0', 0), ('1_0', 1), ('2_0', 2), ('3_0', 3), ('4_0', 4), ('3_1', 3), ('2_2', 2), ('1_3', 1), ('0_4', 0)):
            setattr(self, 'SPADE' + name, SPADE(ctx, nb[lvl], sm, nb[lvl] / sc))       # archs.py:361-371: all SPADEs first
        self.conv0_0 = Bottleneck(input_channels, nb[0])
        self.conv1_0 = Bottleneck(nb[0], nb[1])
        self.conv2_0 = Bottleneck(nb[1], nb[2])
        self.conv3_0 = Bottleneck(nb[2], nb[3])
        self.conv4_0 = Bottleneck(nb[3], nb[4])
        self.conv3_1 = Bottleneck(nb[3] + nb[4], nb[3])
        self.conv2_2 = Bottleneck(nb[2] + nb[3], nb[2])
        self.conv1_3 = Bottleneck(nb[1] + nb[2], nb[1])
        self.conv0_4 = Bottleneck(nb[0] + nb[1], nb[0])
        self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)


class UNet_R_SS(nn.Module):
    """archs.py:469-556: the six-level residual/SPADE U-Net with bilinear upsampling on every level."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [64, 128, 256, 384, 512, 768]
        self.six_step = True
        sm, ctx, sc = num_classes, 'spadebatch3x3', 16
        self.pool = nn.MaxPool2d(2, 2, return_indices=False)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)

        def stage(name, cin, lvl):
            setattr(self, 'conv' + name, BasicBlock(cin, nb[lvl]))
            setattr(self, 'SPADE' + name, SPADE(ctx, nb[lvl], sm, nb[lvl] / sc))

        stage('0_0', input_channels, 0); stage('1_0', nb[0], 1); stage('2_0', nb[1], 2); stage('3_0', nb[2], 3)
        stage('4_0', nb[3], 4); stage('5_0', nb[4], 5); stage('4_1', nb[4] + nb[5], 4); stage('3_1', nb[3] + nb[4], 3)
        stage('2_2', nb[2] + nb[3], 2); stage('1_3', nb[1] + nb[2], 1)
        self.sp_up1_3 = SubPixelConvolutionalBlock(3, nb[1], 2)            # archs.py:515: constructed, never called
        stage('0_4', nb[0] + nb[1], 0)
        self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)
        self.init_weights()

    def init_weights(self):
        init.kaiming_uniform_(self.final.weight, mode='fan_in')
        self.final.bias.data.fill_(0)

    def _stage(self, name, x, x2=None):
        y = getattr(self, 'conv' + name)(x, x2)
        return getattr(self, 'SPADE' + name)(y, y)

    def forward(self, input):
        x = ops.as_nhwc(input)
        up = ops.upsample2x_bilinear
        x0_0 = self._stage('0_0', x)
        x1_0 = self._stage('1_0', _pool(x0_0))
        x2_0 = self._stage('2_0', _pool(x1_0))
        x3_0 = self._stage('3_0', _pool(x2_0))
        x4_0 = self._stage('4_0', _pool(x3_0))
        x5_0 = self._stage('5_0', _pool(x4_0))
        x4_1 = self._stage('4_1', x4_0, up(x5_0))
        x3_1 = self._stage('3_1', x3_0, up(x4_1))
        x2_2 = self._stage('2_2', x2_0, up(x3_1))
        x1_3 = self._stage('1_3', x1_0, up(x2_2))
        x0_4 = self._stage('0_4', x0_0, up(x1_3))
        return ops.conv2d(x0_4, self.final.weight, self.final.bias)


class NestedUNet(nn.Module):
    """archs.py:863-933 (UNet++), optional deep supervision (list of four outputs, train.py:84-95)."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [64, 128, 256, 512, 1024]
        self.deep_supervision = deep_supervision
        self.pool = nn.MaxPool2d(2, 2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv0_0 = VGGBlock(input_channels, nb[0], nb[0])
        self.conv1_0 = VGGBlock(nb[0], nb[1], nb[1])
        self.conv2_0 = VGGBlock(nb[1], nb[2], nb[2])
        self.conv3_0 = VGGBlock(nb[2], nb[3], nb[3])
        self.conv4_0 = VGGBlock(nb[3], nb[4], nb[4])
        self.conv0_1 = VGGBlock(nb[0] + nb[1], nb[0], nb[0])
        self.conv1_1 = VGGBlock(nb[1] + nb[2], nb[1], nb[1])
        self.conv2_1 = VGGBlock(nb[2] + nb[3], nb[2], nb[2])
        self.conv3_1 = VGGBlock(nb[3] + nb[4], nb[3], nb[3])
        self.conv0_2 = VGGBlock(nb[0] * 2 + nb[1], nb[0], nb[0])
        self.conv1_2 = VGGBlock(nb[1] * 2 + nb[2], nb[1], nb[1])
        self.conv2_2 = VGGBlock(nb[2] * 2 + nb[3], nb[2], nb[2])
        self.conv0_3 = VGGBlock(nb[0] * 3 + nb[1], nb[0], nb[0])
        self.conv1_3 = VGGBlock(nb[1] * 3 + nb[2], nb[1], nb[1])
        self.conv0_4 = VGGBlock(nb[0] * 4 + nb[1], nb[0], nb[0])
        if self.deep_supervision:
            self.final1 = nn.Conv2d(nb[0], num_classes, kernel_size=1)
            self.final2 = nn.Conv2d(nb[0], num_classes, kernel_size=1)
            self.final3 = nn.Conv2d(nb[0], num_classes, kernel_size=1)
            self.final4 = nn.Conv2d(nb[0], num_classes, kernel_size=1)
        else:
            self.final = nn.Conv2d(nb[0], num_classes, kernel_size=1)

    def forward(self, input):
        x = ops.as_nhwc(input)
        up = ops.upsample2x_bilinear
        c = lambda blk, *xs: blk(*_cat_split(xs))
        x0_0 = self.conv0_0(x)
        x1_0 = self.conv1_0(_pool(x0_0))
        x0_1 = c(self.conv0_1, x0_0, up(x1_0))
        x2_0 = self.conv2_0(_pool(x1_0))
        x1_1 = c(self.conv1_1, x1_0, up(x2_0))
        x0_2 = c(self.conv0_2, x0_0, x0_1, up(x1_1))
        x3_0 = self.conv3_0(_pool(x2_0))
        x2_1 = c(self.conv2_1, x2_0, up(x3_0))
        x1_2 = c(self.conv1_2, x1_0, x1_1, up(x2_1))
        x0_3 = c(self.conv0_3, x0_0, x0_1, x0_2, up(x1_2))
        x4_0 = self.conv4_0(_pool(x3_0))
        x3_1 = c(self.conv3_1, x3_0, up(x4_0))
        x2_2 = c(self.conv2_2, x2_0, x2_1, up(x3_1))
        x1_3 = c(self.conv1_3, x1_0, x1_1, x1_2, up(x2_2))
        x0_4 = c(self.conv0_4, x0_0, x0_1, x0_2, x0_3, up(x1_3))
        if self.deep_supervision:
            return [ops.conv2d(t, f.weight, f.bias) for t, f in ((x0_1, self.final1), (x0_2, self.final2), (x0_3, self.final3), (x0_4, self.final4))]
        return ops.conv2d(x0_4, self.final.weight, self.final.bias)


class UNet_ori(nn.Module):
    """archs.py:935-996: conv_block encoder, nearest-up_conv decoder."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kwargs):
        super().__init__()
        nb = [64, 128, 256, 512, 1024]
        self.Maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.Conv1 = conv_block(ch_in=input_channels, ch_out=nb[0])
        self.Conv2 = conv_block(ch_in=nb[0], ch_out=nb[1])
        self.Conv3 = conv_block(ch_in=nb[1], ch_out=nb[2])
        self.Conv4 = conv_block(ch_in=nb[2], ch_out=nb[3])
        self.Conv5 = conv_block(ch_in=nb[3], ch_out=nb[4])
        self.Up5 = up_conv(ch_in=nb[4], ch_out=nb[3]); self.Up_conv5 = conv_block(ch_in=nb[4], ch_out=nb[3])
        self.Up4 = up_conv(ch_in=nb[3], ch_out=nb[2]); self.Up_conv4 = conv_block(ch_in=nb[3], ch_out=nb[2])
        self.Up3 = up_conv(ch_in=nb[2], ch_out=nb[1]); self.Up_conv3 = conv_block(ch_in=nb[2], ch_out=nb[1])
        self.Up2 = up_conv(ch_in=nb[1], ch_out=nb[0]); self.Up_conv2 = conv_block(ch_in=nb[1], ch_out=nb[0])
        self.Conv_1x1 = nn.Conv2d(nb[0], num_classes, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        x1 = self.Conv1(ops.as_nhwc(x))
        x2 = self.Conv2(_pool(x1))
        x3 = self.Conv3(_pool(x2))
        x4 = self.Conv4(_pool(x3))
        x5 = self.Conv5(_pool(x4))
        d5 = self.Up_conv5(x4, self.Up5(x5))
        d4 = self.Up_conv4(x3, self.Up4(d5))
        d3 = self.Up_conv3(x2, self.Up3(d4))
        d2 = self.Up_conv2(x1, self.Up2(d3))
        return ops.conv2d(d2, self.Conv_1x1.weight, self.Conv_1x1.bias)
