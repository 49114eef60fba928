"""MBConvBlock and the EfficientNet feature extractor on the HIP kernels (mirrors
efficientnet_pytorch/model.py:18-99,132-218 of the reference; parameter names and creation order kept)."""
import os

import torch
import torch.nn as nn

from .. import bf16, ops
from .._lib import ACT_SWISH
from .utils import (MemoryEfficientSwish, Swish, drop_connect, get_model_params, get_same_padding_conv2d, round_filters,
                    round_repeats)


# SSG_BF16_STEM=0: in bf16 mode the stem stays an fp32 conv + fp32 batch norm followed by a conversion (rounds 2-3)
BF16_STEM = os.environ.get('SSG_BF16_STEM', '1') != '0'


class MBConvBlock(nn.Module):
    """1x1 expand + BN + swish -> depthwise k x k (s) + BN + swish -> squeeze-excite -> 1x1 project + BN
    -> drop-connect + identity skip (model.py:18-99)."""

    def __init__(self, block_args, global_params):
        super().__init__()
        self._block_args = block_args
        self._bn_mom = 1 - global_params.batch_norm_momentum
        self._bn_eps = global_params.batch_norm_epsilon
        self.has_se = (block_args.se_ratio is not None) and (0 < block_args.se_ratio <= 1)
        self.id_skip = block_args.id_skip
        Conv2d = get_same_padding_conv2d(image_size=global_params.image_size)
        inp = block_args.input_filters
        oup = block_args.input_filters * block_args.expand_ratio
        if block_args.expand_ratio != 1:
            self._expand_conv = Conv2d(in_channels=inp, out_channels=oup, kernel_size=1, bias=False)
            self._bn0 = nn.BatchNorm2d(num_features=oup, momentum=self._bn_mom, eps=self._bn_eps)
        self._depthwise_conv = Conv2d(in_channels=oup, out_channels=oup, groups=oup, kernel_size=block_args.kernel_size,
                                      stride=block_args.stride, bias=False)
        self._bn1 = nn.BatchNorm2d(num_features=oup, momentum=self._bn_mom, eps=self._bn_eps)
        if self.has_se:
            nsq = max(1, int(block_args.input_filters * block_args.se_ratio))
            self._se_reduce = Conv2d(in_channels=oup, out_channels=nsq, kernel_size=1)
            self._se_expand = Conv2d(in_channels=nsq, out_channels=oup, kernel_size=1)
        self._project_conv = Conv2d(in_channels=oup, out_channels=block_args.output_filters, kernel_size=1, bias=False)
        self._bn2 = nn.BatchNorm2d(num_features=block_args.output_filters, momentum=self._bn_mom, eps=self._bn_eps)
        self._swish = MemoryEfficientSwish()

    def forward(self, inputs, drop_connect_rate=None):
        if torch.is_tensor(inputs) and inputs.dtype == torch.bfloat16:
            return self._forward_bf16(inputs, drop_connect_rate)
        x = ops.as_nhwc(inputs)
        inputs = x
        # swish(bn(.)) (model.py:75,80) is ONE pass: the batch-norm apply kernel evaluates the swish, and the backward
        # recomputes the pre-activation from the conv output and (scale, shift) instead of storing it
        if self._block_args.expand_ratio != 1:
            x = ops.batch_norm_act(self._expand_conv(x), self._bn0, act=ACT_SWISH)
        x = ops.batch_norm_act(self._depthwise_conv(x), self._bn1, act=ACT_SWISH)
        if self.has_se:
            sq = ops.global_avgpool(x)
            gate = ops.se_gate(sq, self._se_reduce, self._se_expand)          # model.py:84-86 in 2 + 3 kernels (csrc/se_gate.hip)
            if gate is None:
                gate = ops.sigmoid(self._se_expand(self._swish(self._se_reduce(sq))))
            x = ops.channel_scale(x, gate)
        x = self._project_conv(x)
        a = self._block_args
        # model.py:93-94 compares `stride == 1` literally: decoded BlockArgs carry stride as a LIST ([1]), so
        # the first block of a stage never takes the skip; repeats get stride=1 (int, model.py:162) and do.
        skip = self.id_skip and a.stride == 1 and a.input_filters == a.output_filters
        if skip and not (drop_connect_rate and self.training):
            return ops.batch_norm_act(x, self._bn2, res=inputs)            # bn2(...) + inputs in one pass
        x = ops.batch_norm_act(x, self._bn2)
        if skip:
            x = drop_connect(x, p=drop_connect_rate, training=self.training)
            x = ops.add(x, inputs)
        return x

    def _forward_bf16(self, inputs, drop_connect_rate=None):
        """The same block on a bf16 tensor (BASELINE config 4): pointwise convs on the bf16 MFMA GEMM, depthwise / batch norm
        / squeeze-excite pooling and gating on the bf16 instantiations; the two tiny SE convs see fp32 [N, C, 1, 1]."""
        x = inputs
        dw = self._depthwise_conv
        if self._block_args.expand_ratio != 1:
            x = bf16.batch_norm_act(bf16.conv1x1(x, self._expand_conv.weight), self._bn0, act=ACT_SWISH)
        x = bf16.batch_norm_act(bf16.dwconv2d(x, dw.weight, dw.stride[0], dw.static_pad), self._bn1, act=ACT_SWISH)
        if self.has_se:
            sq = bf16.global_avgpool(x)
            gate = ops.se_gate(sq, self._se_reduce, self._se_expand)
            if gate is None:
                gate = ops.sigmoid(self._se_expand(self._swish(self._se_reduce(sq))))
            x = bf16.channel_scale(x, gate)
        x = bf16.conv1x1(x, self._project_conv.weight)
        a = self._block_args
        skip = self.id_skip and a.stride == 1 and a.input_filters == a.output_filters
        if skip and not (drop_connect_rate and self.training):
            return bf16.batch_norm_act(x, self._bn2, res=inputs)
        x = bf16.batch_norm_act(x, self._bn2)
        if skip:
            n, c = x.shape[0], x.shape[1]
            keep = 1 - drop_connect_rate                                     # utils.py:83-92
            mask = torch.floor(keep + torch.rand([n, 1, 1, 1], dtype=torch.float32, device=x.device)) / keep
            x = bf16.channel_scale(x, mask.expand(n, c, 1, 1).contiguous(memory_format=torch.channels_last))
            x = bf16.add(x, inputs)
        return x

    def set_swish(self, memory_efficient=True):
        self._swish = MemoryEfficientSwish() if memory_efficient else Swish()


class EfficientNet(nn.Module):
    """model.py:132-260.  `extract_features` is the per-op scope; the classifier head (`_fc`) exists for
    state_dict parity but `forward` (classification) is out of scope and raises."""

    def __init__(self, blocks_args=None, global_params=None):
        super().__init__()
        assert isinstance(blocks_args, list) and len(blocks_args) > 0
        self._global_params = global_params
        self._blocks_args = blocks_args
        Conv2d = get_same_padding_conv2d(image_size=global_params.image_size)
        bn_mom = 1 - global_params.batch_norm_momentum
        bn_eps = global_params.batch_norm_epsilon
        out_channels = round_filters(32, global_params)
        self._conv_stem = Conv2d(3, out_channels, kernel_size=3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(num_features=out_channels, momentum=bn_mom, eps=bn_eps)
        self._blocks = nn.ModuleList([])
        for block_args in self._blocks_args:
            block_args = block_args._replace(input_filters=round_filters(block_args.input_filters, global_params),
                                             output_filters=round_filters(block_args.output_filters, global_params),
                                             num_repeat=round_repeats(block_args.num_repeat, global_params))
            self._blocks.append(MBConvBlock(block_args, global_params))
            if block_args.num_repeat > 1:
                block_args = block_args._replace(input_filters=block_args.output_filters, stride=1)
            for _ in range(block_args.num_repeat - 1):
                self._blocks.append(MBConvBlock(block_args, global_params))
        in_channels = block_args.output_filters
        out_channels = round_filters(1280, global_params)
        self._conv_head = Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
        self._bn1 = nn.BatchNorm2d(num_features=out_channels, momentum=bn_mom, eps=bn_eps)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(global_params.dropout_rate)
        self._fc = nn.Linear(out_channels, global_params.num_classes)
        self._swish = MemoryEfficientSwish()

    def set_swish(self, memory_efficient=True):
        self._swish = MemoryEfficientSwish() if memory_efficient else Swish()
        for b in self._blocks:
            b.set_swish(memory_efficient)

    def set_compute_dtype(self, dtype):
        """torch.float32 (default: the reference's arithmetic) or torch.bfloat16 (BASELINE config 4): activations between
        the stem and the head live in HBM as bf16, the 1x1 convolutions run on v_mfma_f32_32x32x16_bf16 with fp32
        accumulation; parameters, batch-norm statistics, SE gates and every gradient of a parameter stay fp32.  The stem
        (a 3-channel 3x3 stride-2 conv) stays on the fp32 path; extract_features still takes and returns fp32 tensors."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('compute dtype must be torch.float32 or torch.bfloat16')
        self._ssg_dtype = dtype
        return self

    def extract_features(self, inputs):
        lowp = getattr(self, '_ssg_dtype', torch.float32) == torch.bfloat16
        stem = self._conv_stem
        if lowp and BF16_STEM and stem.bias is None and stem.out_channels % 8 == 0:
            # the stem in the bf16 family too (csrc/conv_stem_bf16.hip): bf16-rounded image and weights, fp32 accumulation, bf16 output
            x = bf16.batch_norm_act(bf16.conv_thin(ops.as_nhwc(inputs), stem.weight, stem.stride[0], stem.static_pad), self._bn0, act=ACT_SWISH)
        else:
            x = ops.batch_norm_act(stem(ops.as_nhwc(inputs)), self._bn0, act=ACT_SWISH)
            if lowp:
                x = bf16.to_bf16(x)
        for idx, block in enumerate(self._blocks):
            rate = self._global_params.drop_connect_rate
            if rate:
                rate *= float(idx) / len(self._blocks)
            x = block(x, drop_connect_rate=rate)
        if lowp:
            x = bf16.batch_norm_act(bf16.conv1x1(x, self._conv_head.weight), self._bn1, act=ACT_SWISH)
            return bf16.to_f32(x)
        return ops.batch_norm_act(self._conv_head(x), self._bn1, act=ACT_SWISH)

    def forward(self, inputs):
        raise NotImplementedError('the ImageNet classifier head is out of scope (SURVEY.md 2, row 10); use extract_features')

    @classmethod
    def from_name(cls, model_name, override_params=None):
        cls._check_model_name_is_valid(model_name)
        blocks_args, global_params = get_model_params(model_name, override_params)
        return cls(blocks_args, global_params)

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise NotImplementedError('from_pretrained needs ../pretrained/normal/*.pth, which the reference does not ship')

    @classmethod
    def _check_model_name_is_valid(cls, model_name):
        valid = ['efficientnet-b' + str(i) for i in range(9)]
        if model_name not in valid:
            raise ValueError('model_name should be one of: ' + ', '.join(valid))
