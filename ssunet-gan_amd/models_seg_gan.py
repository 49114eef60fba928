"""Generator / Discriminator, MI355X-native (mirrors the reference's scripts/models_seg_gan.py).

Constructor signatures, module names and state_dict keys are the reference's
(models_seg_gan.py:13-64,193-300); forwards run on the HIP kernels."""
import os

import torch
import torch.nn as nn

from . import archs, ops
from ._lib import ACT_LRELU, ACT_NONE
from .spectral_norm import SpectralNorm


def remove_prefix(state_dict, prefix):
    f = lambda x: x.split(prefix, 1)[-1] if x.startswith(prefix) else x
    return {f(key): value for key, value in state_dict.items()}


class ConvolutionalBlock(nn.Module):
    """models_seg_gan.py:13-64: conv (+bias) [+ BN] [+ activation]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, batch_norm=False, activation=None):
        super().__init__()
        if activation is not None:
            activation = activation.lower()
            assert activation in {'prelu', 'leakyrelu', 'tanh'}
        layers = [nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                            padding=kernel_size // 2)]
        if batch_norm is True:
            layers.append(nn.BatchNorm2d(num_features=out_channels))
        if activation == 'prelu':
            layers.append(nn.PReLU())
        elif activation == 'leakyrelu':
            layers.append(nn.LeakyReLU(0.2))
        elif activation == 'tanh':
            layers.append(nn.Tanh())
        self.conv_block = nn.Sequential(*layers)
        self._act = activation
        self._bn = batch_norm is True

    def forward(self, input):
        conv = self.conv_block[0]
        # The conv runs through ops.conv2d on conv.weight, not through nn.Conv2d.__call__: forward pre-hooks registered on the
        # conv module (spectral_norm re-parametrises `weight` in one: spectral_norm.py:99-101) are honoured here.
        sn = None
        for hook in conv._forward_pre_hooks.values():
            if isinstance(hook, SpectralNorm) and hook.name == 'weight' and hook.dim == 0 and conv.weight_orig.is_cuda:
                sn = hook            # handled inside the conv below: W / sigma is never materialised (ops.conv2d_sn)
            else:
                hook(conv, (input,))
        if self._act not in (None, 'leakyrelu'):
            raise NotImplementedError('ConvolutionalBlock activation %r has no HIP path (only the discriminator\'s '
                                      'LeakyReLU flavour is on the hot path)' % self._act)
        act = ACT_LRELU if self._act == 'leakyrelu' else ACT_NONE
        slope = self.conv_block[-1].negative_slope if act == ACT_LRELU else 0.0
        if sn is not None:
            def run(a, sl):
                return ops.conv2d_sn(input, conv.weight_orig, conv.weight_u, conv.weight_v, conv.bias, conv.stride[0], conv.padding[0],
                                     act=a, slope=sl, n_power_iterations=sn.n_power_iterations if conv.training else 0, eps=sn.eps)
        else:
            def run(a, sl, bn_stats=False):
                return ops.conv2d(input, conv.weight, conv.bias, conv.stride[0], conv.padding[0], act=a, slope=sl, bn_stats=bn_stats)
        if not self._bn:
            return run(act, slope)
        bn = self.conv_block[1]
        part = None
        if sn is None and bn.training:
            y, part = run(ACT_NONE, 0.0, bn_stats=True)       # (sum, sum of squares) of y from the conv epilogue
        else:
            y = run(ACT_NONE, 0.0)
        return ops.batch_norm_act(y, bn, act=act, slope=slope, group=getattr(bn, '_ssg_sync_group', None), stats_part=part)


class Generator(nn.Module):
    """models_seg_gan.py:193-243: thin wrapper around archs.__dict__[config['arch']]."""

    def __init__(self, config):
        super().__init__()
        if config['arch'] not in archs.__all__:
            raise NotImplementedError('arch %r is not built in ssunet-gan_amd (available: %s)' % (config['arch'], archs.__all__))
        self.net = archs.__dict__[config['arch']](config['num_classes'], config['input_channels'], config['deep_supervision'])

    def initialize_with_srresnet(self, model_folder, config):
        """models_seg_gan.py:216-227: load stage-1 weights, stripping 'module.'.  The reference
        loads with strict=False and is silent on mismatches; here a key mismatch raises."""
        model_dict = torch.load(os.path.join(model_folder, '%s/model.pth' % config['name']), map_location='cpu')
        if 'state_dict' in model_dict.keys():
            model_dict = remove_prefix(model_dict['state_dict'], 'module.')
        else:
            model_dict = remove_prefix(model_dict, 'module.')
        missing, unexpected = self.net.load_state_dict(model_dict, strict=False)
        if missing or unexpected:
            raise RuntimeError('checkpoint key mismatch: missing %s unexpected %s' % (missing[:5], unexpected[:5]))
        print("\nLoaded weights from pre-trained SS-UNet-R.\n")

    def forward(self, lr_imgs):
        return self.net(lr_imgs)


class Discriminator(nn.Module):
    """models_seg_gan.py:246-300."""

    def __init__(self, num_classes, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024):
        super().__init__()
        in_channels = num_classes
        conv_blocks = list()
        for i in range(n_blocks):
            out_channels = (n_channels if i == 0 else in_channels * 2) if i % 2 == 0 else in_channels
            conv_blocks.append(ConvolutionalBlock(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                                                  stride=1 if i % 2 == 0 else 2, batch_norm=i != 0, activation='LeakyReLu'))
            in_channels = out_channels
        self.conv_blocks = nn.Sequential(*conv_blocks)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((6, 6))
        self.fc1 = nn.Linear(out_channels * 6 * 6, fc_size)
        self.leaky_relu = nn.LeakyReLU(0.2)
        self.fc2 = nn.Linear(1024, 1)

    def forward(self, imgs):
        x = ops.as_nhwc(imgs)
        x = self.conv_blocks(x)
        flat = ops.adaptive_avgpool_flat(x, 6)
        h = ops.linear(flat, self.fc1.weight, self.fc1.bias, act=ACT_LRELU, slope=self.leaky_relu.negative_slope)
        return ops.linear(h, self.fc2.weight, self.fc2.bias)
