"""Synchronized BatchNorm, MI355X-native (mirrors the surface of the reference's vendored
Synchronized-BatchNorm-PyTorch: scripts/batchnorm.py + comm.py + replicate.py; unwired there --
SURVEY.md 8a row A9).

The reference synchronises replicas of ONE process over Python threads and queue.Queue rendezvous
(comm.py:18-138), reducing [sum, ssum] to a master device and broadcasting [mean, inv_std] back
(batchnorm.py:92-113).  Here every GPU is its own process: the fp64 [sum, ssum] vector is all-reduced
over RCCL between the two kernel stages of the batch norm, every rank evaluates
`_compute_mean_std` (batchnorm.py:115-127: mean = sum/n, biased var, inv_std = clamp(var, eps)^-1/2,
running_var from the unbiased var) itself, and backward all-reduces [sum g, sum g*xhat] the same way.
"""
import torch.nn as nn

from . import dp, ops
from ._lib import ACT_NONE

__all__ = ['SynchronizedBatchNorm1d', 'SynchronizedBatchNorm2d', 'SynchronizedBatchNorm3d', 'convert_model',
           'patch_sync_batchnorm', 'DataParallelWithCallback', 'patch_replication_callback']


class _SynchronizedBatchNorm(nn.modules.batchnorm._BatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)

    def _group(self):
        g = getattr(self, '_ssg_sync_group', None)
        if g is None and dp.is_dist():
            import torch.distributed as dist
            g = dist.group.WORLD
        return g

    def forward(self, input):
        shape = input.shape
        x = input if input.dim() == 4 else input.reshape(shape[0], shape[1], -1, 1)
        group = self._group()
        # batchnorm.py:52-55: single device or eval -> F.batch_norm (torch formula); parallel+train -> sync formula
        y = ops.batch_norm_act(x, self, act=ACT_NONE, group=group if self.training else None,
                               var_mode=(1 if (group is not None and self.training) else 0))
        return y if input.dim() == 4 else y.reshape(shape)


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    pass


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    pass


def convert_model(module):
    """batchnorm.py:313-361: swap every torch BatchNorm for its synchronized twin (weights shared)."""
    mod = module
    for src, dst in ((nn.BatchNorm1d, SynchronizedBatchNorm1d), (nn.BatchNorm2d, SynchronizedBatchNorm2d),
                     (nn.BatchNorm3d, SynchronizedBatchNorm3d)):
        if isinstance(module, src) and not isinstance(module, _SynchronizedBatchNorm):
            mod = dst(module.num_features, module.eps, module.momentum, module.affine)
            mod.running_mean, mod.running_var = module.running_mean, module.running_var
            mod.num_batches_tracked = module.num_batches_tracked
            if module.affine:
                mod.weight, mod.bias = module.weight, module.bias
    for name, child in module.named_children():
        mod.add_module(name, convert_model(child))
    return mod


def patch_sync_batchnorm(module, group=None):
    """In-place alternative to convert_model: keep the modules, mark them synchronized."""
    return dp.convert_sync_batchnorm(module, group)


class DataParallelWithCallback(nn.Module):
    """replicate.py:50-67 wrapped nn.DataParallel so replicas could rendezvous.  With one process per
    GPU there is nothing to replicate: this keeps the name and simply forwards to the wrapped module."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def patch_replication_callback(data_parallel):
    return data_parallel
