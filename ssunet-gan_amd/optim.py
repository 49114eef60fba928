"""Fused clip + Adam step on a torch.optim.Adam instance (train_seg_gan.py:211-215,229-233).

The training loop receives stock torch.optim.Adam objects from its caller; this helper runs
the same update as `clip_gradient(opt, c); opt.step()` in ONE multi-tensor HIP launch while
keeping the optimizer's own state (`exp_avg`, `exp_avg_sq`, `step`) authoritative, so
optimizer.state_dict() stays interchangeable with the reference's."""
import math

import torch

from . import _lib, ops
from ._lib import call, ptr, stream_ptr

ADAM_CHUNK = 4096
_PLAN_CACHE = {}


def _supported(opt):
    if type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable'):
            return False
        if isinstance(g['lr'], torch.Tensor):
            return False
    return True


def _plan(params, states, device):
    key = tuple((p.data_ptr(), p.grad.data_ptr(), s['exp_avg'].data_ptr(), s['exp_avg_sq'].data_ptr()) for p, s in zip(params, states))
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        return hit
    ptrs, sizes, blk_t, blk_c = [], [], [], []
    for t, (p, s) in enumerate(zip(params, states)):
        ptrs += [p.data_ptr(), p.grad.data_ptr(), s['exp_avg'].data_ptr(), s['exp_avg_sq'].data_ptr()]
        n = p.numel()
        sizes.append(n)
        for c in range((n + ADAM_CHUNK - 1) // ADAM_CHUNK):
            blk_t.append(t); blk_c.append(c)
    plan = (torch.tensor(ptrs, dtype=torch.int64).to(device), torch.tensor(sizes, dtype=torch.int64).to(device),
            torch.tensor(blk_t, dtype=torch.int32).to(device), torch.tensor(blk_c, dtype=torch.int32).to(device), len(blk_t))
    if len(_PLAN_CACHE) > 64:
        _PLAN_CACHE.clear()
    _PLAN_CACHE[key] = plan
    return plan


def clip_adam_step(optimizer, grad_clip=None):
    """Equivalent of `clip_gradient(optimizer, grad_clip); optimizer.step()`."""
    if not _supported(optimizer):
        raise NotImplementedError('clip_adam_step supports plain torch.optim.Adam (no amsgrad/maximize/capturable)')
    for group in optimizer.param_groups:
        params = [p for p in group['params'] if p.grad is not None]
        if not params:
            continue
        states = []
        for p in params:
            _lib.require_gpu(p)
            if not (p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32):
                raise ValueError('clip_adam_step: parameters and grads must be contiguous fp32')
            st = optimizer.state[p]
            if len(st) == 0:                    # same lazy init as torch.optim.Adam
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            states.append(st)
        steps = {float(st['step']) for st in states}
        if len(steps) != 1:
            raise NotImplementedError('clip_adam_step: parameters with different step counts in one group')
        step = steps.pop() + 1.0
        for st in states:
            st['step'] += 1
        beta1, beta2 = group['betas']
        bc1 = 1.0 - beta1 ** step
        bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
        ptrs, sizes, blk_t, blk_c, nblk = _plan(params, states, params[0].device)
        call('ssg_clamp_adam_multi_f32', ptr(ptrs), ptr(sizes), ptr(blk_t), ptr(blk_c), nblk,
             float(grad_clip) if grad_clip else 0.0, float(group['lr']), float(beta1), float(beta2), float(group['eps']),
             float(group['weight_decay']), bc1, bc2_sqrt, stream_ptr())
    ops.bump_weight_epoch()


def clamp_parameters_(params, clip):
    """`for p in params: p.data.clamp_(-clip, clip)` (train.py:111-112) in one multi-tensor launch."""
    params = [p for p in params]
    if not params:
        return
    key = ('clampw',) + tuple(p.data_ptr() for p in params)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        ptrs, sizes, blk_t, blk_c = [], [], [], []
        for t, p in enumerate(params):
            _lib.require_gpu(p)
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError('clamp_parameters_: parameters must be contiguous fp32')
            ptrs += [p.data_ptr(), 0, 0, 0]
            sizes.append(p.numel())
            for c in range((p.numel() + ADAM_CHUNK - 1) // ADAM_CHUNK):
                blk_t.append(t); blk_c.append(c)
        dev = params[0].device
        plan = (torch.tensor(ptrs, dtype=torch.int64).to(dev), torch.tensor(sizes, dtype=torch.int64).to(dev),
                torch.tensor(blk_t, dtype=torch.int32).to(dev), torch.tensor(blk_c, dtype=torch.int32).to(dev), len(blk_t))
        _PLAN_CACHE[key] = plan
    ptrs, sizes, blk_t, blk_c, nblk = plan
    call('ssg_clamp_multi_f32', ptr(ptrs), ptr(sizes), ptr(blk_t), ptr(blk_c), nblk, 0, -float(clip), float(clip), stream_ptr())
    ops.bump_weight_epoch()              # packed-weight caches must see the clamped values
