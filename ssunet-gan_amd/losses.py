"""Losses on the hot path, MI355X-native (mirrors the reference's scripts/losses.py:130-136,274-302)."""
import torch
import torch.nn as nn

from . import ops

__all__ = ['BCEDiceLoss', 'StableBCELoss']


class StableBCELoss(nn.Module):
    """losses.py:130-136: mean(max(x,0) - x*t + log(1+exp(-|x|)))."""

    def forward(self, input, target):
        return ops.seg_loss(input, target)[2]


class BCEDiceLoss(nn.Module):
    """losses.py:274-302: 0.5*StableBCE + (1 - mean_n softDice_n), 2*dice if the BCE is inf/nan.
    One fused HIP pass also yields MSE, IoU and Dice (see ops.seg_loss); train() reuses it."""

    def forward(self, input, target):
        return ops.seg_loss(input, target)[0]
