"""Helpers of the EfficientNet encoder (mirrors efficientnet_pytorch/utils.py of the reference: same
names and numbers, HIP forwards)."""
import collections
import math
import re
from functools import partial

import torch
import torch.nn as nn

from .. import ops

GlobalParams = collections.namedtuple('GlobalParams', [
    'batch_norm_momentum', 'batch_norm_epsilon', 'dropout_rate', 'num_classes', 'width_coefficient',
    'depth_coefficient', 'depth_divisor', 'min_depth', 'drop_connect_rate', 'image_size'])
BlockArgs = collections.namedtuple('BlockArgs', [
    'kernel_size', 'num_repeat', 'input_filters', 'output_filters', 'expand_ratio', 'id_skip', 'stride', 'se_ratio'])
GlobalParams.__new__.__defaults__ = (None,) * len(GlobalParams._fields)
BlockArgs.__new__.__defaults__ = (None,) * len(BlockArgs._fields)


class MemoryEfficientSwish(nn.Module):
    """utils.py:37-52: x*sigmoid(x) with backward grad*(s*(1+x*(1-s))); one HIP kernel each way."""

    def forward(self, x):
        return ops.swish(x)


Swish = MemoryEfficientSwish          # utils.py:54-56: same function, autograd-derived backward


def round_filters(filters, global_params):
    """utils.py:60-72."""
    mult = global_params.width_coefficient
    if not mult:
        return filters
    div = global_params.depth_divisor
    min_depth = global_params.min_depth or div
    filters *= mult
    new = max(min_depth, int(filters + div / 2) // div * div)
    if new < 0.9 * filters:
        new += div
    return int(new)


def round_repeats(repeats, global_params):
    """utils.py:75-80."""
    mult = global_params.depth_coefficient
    return repeats if not mult else int(math.ceil(mult * repeats))


def drop_connect(inputs, p, training):
    """utils.py:83-92: per-sample Bernoulli(keep) mask scaled by 1/keep (RNG-dependent in training)."""
    if not training:
        return inputs
    n, c = inputs.shape[0], inputs.shape[1]
    keep = 1 - p
    mask = torch.floor(keep + torch.rand([n, 1, 1, 1], dtype=inputs.dtype, device=inputs.device)) / keep
    return ops.channel_scale(inputs, mask.expand(n, c, 1, 1).contiguous(memory_format=torch.channels_last))


def same_padding(size, k, s, d=1):
    """TensorFlow "SAME" padding for one axis of a `size`-long input: (before, after)."""
    out = math.ceil(size / s)
    pad = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return pad // 2, pad - pad // 2


class Conv2dStaticSamePadding(nn.Conv2d):
    """utils.py:123-146: padding fixed at construction from `image_size`.  Dense convs run on the MFMA
    implicit GEMM, depthwise ones (groups == channels) on the depthwise kernel; the zero padding is
    folded into the kernels' tap offsets instead of a ZeroPad2d copy."""

    def __init__(self, in_channels, out_channels, kernel_size, image_size=None, **kwargs):
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.stride = self.stride if len(self.stride) == 2 else [self.stride[0]] * 2
        assert image_size is not None
        ih, iw = image_size if type(image_size) == list else [image_size, image_size]
        kh, kw = self.weight.size()[-2:]
        pt, pb = same_padding(ih, kh, self.stride[0], self.dilation[0])
        pl, pr = same_padding(iw, kw, self.stride[1], self.dilation[1])
        self.static_pad = (pt, pb, pl, pr)

    def forward(self, x):
        if self.dilation[0] != 1 or self.stride[0] != self.stride[1]:
            raise NotImplementedError('dilated / anisotropic-stride conv has no HIP path')
        if self.groups == 1:
            return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.static_pad)
        if self.groups == self.in_channels == self.out_channels:
            return ops.dwconv2d(x, self.weight, self.bias, self.stride[0], self.static_pad)
        raise NotImplementedError('grouped conv with 1 < groups < channels has no HIP path')


def get_same_padding_conv2d(image_size=None):
    if image_size is None:
        raise NotImplementedError('dynamic same-padding conv: pass image_size (from_name does)')
    return partial(Conv2dStaticSamePadding, image_size=image_size)


def efficientnet_params(model_name):
    """utils.py:162-177: (width, depth, resolution, dropout)."""
    return {
        'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b1': (1.0, 1.1, 240, 0.2),
        'efficientnet-b2': (1.1, 1.2, 260, 0.3), 'efficientnet-b3': (1.2, 1.4, 300, 0.3),
        'efficientnet-b4': (1.4, 1.8, 380, 0.4), 'efficientnet-b5': (1.6, 2.2, 456, 0.4),
        'efficientnet-b6': (1.8, 2.6, 528, 0.5), 'efficientnet-b7': (2.0, 3.1, 600, 0.5),
        'efficientnet-b8': (2.2, 3.6, 672, 0.5), 'efficientnet-l2': (4.3, 5.3, 800, 0.5),
    }[model_name]


class BlockDecoder(object):
    """utils.py:180-246: 'r1_k3_s11_e1_i32_o16_se0.25' -> BlockArgs."""

    @staticmethod
    def _decode_block_string(block_string):
        opts = {}
        for op in block_string.split('_'):
            m = re.split(r'(\d.*)', op)
            if len(m) >= 2:
                opts[m[0]] = m[1]
        assert len(opts['s']) == 1 or (len(opts['s']) == 2 and opts['s'][0] == opts['s'][1])
        return BlockArgs(kernel_size=int(opts['k']), num_repeat=int(opts['r']), input_filters=int(opts['i']),
                         output_filters=int(opts['o']), expand_ratio=int(opts['e']), id_skip=('noskip' not in block_string),
                         se_ratio=float(opts['se']) if 'se' in opts else None, stride=[int(opts['s'][0])])

    @staticmethod
    def decode(string_list):
        return [BlockDecoder._decode_block_string(s) for s in string_list]


def efficientnet(width_coefficient=None, depth_coefficient=None, dropout_rate=0.2, drop_connect_rate=0.2,
                 image_size=None, num_classes=1000):
    """utils.py:249-276."""
    blocks = BlockDecoder.decode([
        'r1_k3_s11_e1_i32_o16_se0.25', 'r2_k3_s22_e6_i16_o24_se0.25', 'r2_k5_s22_e6_i24_o40_se0.25',
        'r3_k3_s22_e6_i40_o80_se0.25', 'r3_k5_s11_e6_i80_o112_se0.25', 'r4_k5_s22_e6_i112_o192_se0.25',
        'r1_k3_s11_e6_i192_o320_se0.25'])
    gp = GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, dropout_rate=dropout_rate,
                      drop_connect_rate=drop_connect_rate, num_classes=num_classes, width_coefficient=width_coefficient,
                      depth_coefficient=depth_coefficient, depth_divisor=8, min_depth=None, image_size=image_size)
    return blocks, gp


def get_model_params(model_name, override_params):
    if not model_name.startswith('efficientnet'):
        raise NotImplementedError('model name is not pre-defined: %s' % model_name)
    w, d, s, p = efficientnet_params(model_name)
    blocks, gp = efficientnet(width_coefficient=w, depth_coefficient=d, dropout_rate=p, image_size=s)
    if override_params:
        gp = gp._replace(**override_params)
    return blocks, gp
