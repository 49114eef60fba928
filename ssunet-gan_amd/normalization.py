"""SPADE block, MI355X-native (mirrors the reference's scripts/normalization.py:67-122).

Same constructor, same parameter/buffer names and creation order (so `torch.manual_seed`
reproduces the reference's initial weights and state_dicts interchange); the forward pass
runs on the hand-written HIP kernels (blocks._SpadeFn)."""
import re

import torch.nn as nn

from .blocks import spade_self


class SPADE(nn.Module):
    def __init__(self, config_text, norm_nc, label_nc, nhidden=64):
        super().__init__()
        assert config_text.startswith('spade')
        parsed = re.search(r'spade(\D+)(\d)x\d', config_text)
        norm_type = str(parsed.group(1))
        ks = int(parsed.group(2))
        # normalization.py:76-84: the param-free norm is constructed (its buffers are part of the
        # state_dict) but the wired forward never applies it (normalization.py:110).
        if norm_type == 'instance':
            self.param_free_norm = nn.InstanceNorm2d(norm_nc, affine=False)
        elif norm_type in ('batch', 'syncbatch'):
            self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)
        else:
            raise ValueError('%s is not a recognized param-free norm type in SPADE' % norm_type)
        nhidden = int(max(nhidden, 4))
        pw = ks // 2
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, nhidden, kernel_size=ks, padding=pw), nn.ReLU())
        self.x2map = nn.Conv2d(norm_nc, label_nc, kernel_size=ks, padding=pw)
        self.mlp_gamma = nn.Conv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)
        self.mlp_beta = nn.Conv2d(nhidden, norm_nc, kernel_size=ks, padding=pw)

    def forward(self, x, segmap=None):
        if segmap is None or segmap is x:
            return spade_self(x, self.x2map, self.mlp_shared[0], self.mlp_gamma, self.mlp_beta)
        raise NotImplementedError('SPADE with segmap != x has no HIP path: every SPADE call on the hot path is '
                                  'self-conditioned (archs.py:626-669)')
