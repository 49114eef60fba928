"""MBConvBlock and the EfficientNet feature extractor on the HIP kernels (mirrors
efficientnet_pytorch/model.py:18-99,132-218 of the reference; parameter names and creation order kept)."""
import torch
import torch.nn as nn

from .. import ops
from .utils import (MemoryEfficientSwish, Swish, drop_connect, get_model_params, get_same_padding_conv2d, round_filters,
                    round_repeats)


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
        x = ops.as_nhwc(inputs)
        inputs = x
        if self._block_args.expand_ratio != 1:
            x = self._swish(ops.batch_norm_act(self._expand_conv(x), self._bn0))
        x = self._swish(ops.batch_norm_act(self._depthwise_conv(x), self._bn1))
        if self.has_se:
            sq = ops.global_avgpool(x)
            sq = self._se_expand(self._swish(self._se_reduce(sq)))
            x = ops.channel_scale(x, ops.sigmoid(sq))
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

    def extract_features(self, inputs):
        x = self._swish(ops.batch_norm_act(self._conv_stem(ops.as_nhwc(inputs)), self._bn0))
        for idx, block in enumerate(self._blocks):
            rate = self._global_params.drop_connect_rate
            if rate:
                rate *= float(idx) / len(self._blocks)
            x = block(x, drop_connect_rate=rate)
        return self._swish(ops.batch_norm_act(self._conv_head(x), self._bn1))

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
