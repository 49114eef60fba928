"""Losses on the hot path, MI355X-native (mirrors the reference's scripts/losses.py:130-136,274-302)."""
import torch
import torch.nn as nn

from . import ops

__all__ = ['BCEDiceLoss', 'LovaszHingeLoss']          # as the reference's losses.__all__ (losses.py:14)


class StableBCELoss(nn.Module):
    """losses.py:130-136: mean(max(x,0) - x*t + log(1+exp(-|x|)))."""

    def forward(self, input, target):
        return ops.seg_loss(input, target)[2]


class BCEDiceLoss(nn.Module):
    """losses.py:274-302: 0.5*StableBCE + (1 - mean_n softDice_n), 2*dice if the BCE is inf/nan.
    One fused HIP pass also yields MSE, IoU and Dice (see ops.seg_loss); train() reuses it."""

    def forward(self, input, target):
        return ops.seg_loss(input, target)[0]


class LovaszHingeLoss(nn.Module):
    """losses.py:222-233.  Exported by the reference's losses.__all__ (so it is a legal config['loss'] name), but the only
    config selects BCEDiceLoss (config_v1.json:26) and the Lovasz family is off the hot path: there is
    no HIP kernel for it, and no CPU fallback by design -- constructing it is allowed, calling it raises."""

    def forward(self, input, target):
        raise NotImplementedError('LovaszHingeLoss has no HIP path in ssunet-gan_amd (off the hot path; use BCEDiceLoss)')
