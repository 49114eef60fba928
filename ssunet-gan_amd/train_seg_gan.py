"""Stage-2 GAN training step, MI355X-native (mirrors train()/validate() of the reference's
scripts/train_seg_gan.py:167-294 -- same signature, same sequence, same hyper-parameters).

What differs from the reference, by design:
  * BCEDice + MSE + IoU + Dice come from ONE fused HIP pass over (logits, target) instead of
    four torch/numpy passes; IoU/Dice stay on the device (the reference syncs twice per step);
  * `clip_gradient` + `optimizer.step()` is one multi-tensor HIP launch that updates the
    caller's torch.optim.Adam state in place;
  * with torch.distributed initialised (one process per GPU over RCCL), gradients are
    all-reduced per bucket on a side stream, overlapped with backward (dp.GradSync), which
    replaces nn.DataParallel (train_seg_gan.py:480-481).
"""
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from . import dp, ops
from .losses import BCEDiceLoss
from .optim import clip_adam_step
from .srgan_utils import clip_gradient
from .utils import AverageMeter

ALPA, BETA, GRAD_CLIP = 1e-4, 1e-3, 0.8            # train_seg_gan.py:172-174


def _to_float(v):
    return float(v.item()) if torch.is_tensor(v) else float(v)


def _adv_loss(criterion, logits, label):
    if isinstance(criterion, nn.BCEWithLogitsLoss) and criterion.weight is None and criterion.pos_weight is None \
            and criterion.reduction == 'mean':
        return ops.bce_with_logits_const(logits, label)
    return criterion(logits, torch.full_like(logits, label))


def _step_optimizer(optimizer, grad_clip):
    if type(optimizer) is torch.optim.Adam:
        clip_adam_step(optimizer, grad_clip)
    else:                                           # any other optimizer: HIP clamp, then its own step
        if grad_clip is not None:
            clip_gradient(optimizer, grad_clip)
        optimizer.step()
        ops.bump_weight_epoch()


# Default: every gradient the reference computes is computed.  SSG_ELIDE_DEAD_D_GRADS=1 is an opt-in
# measurement switch (see gan_step).
ELIDE_DEAD_D_GRADS = os.environ.get('SSG_ELIDE_DEAD_D_GRADS', '0') == '1'


class _stage(object):
    """Profiler range around one stage of the step: torch.cuda.nvtx is roctx on ROCm, so
    `rocprofv3 --marker-trace --kernel-trace` splits a step into G forward / losses / G backward / G optimizer / D forwards /
    D backward / D optimizer.  SSG_MARKERS=0 switches the ranges off (they cost ~1 us each)."""
    ON = os.environ.get('SSG_MARKERS', '1') != '0'

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _stage.ON:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *a):
        if _stage.ON:
            torch.cuda.nvtx.range_pop()


class _params_frozen(object):
    """Run a forward with the module's parameters not requiring grad (restored on exit)."""

    def __init__(self, module, on):
        self.params = [p for p in module.parameters() if p.requires_grad] if on else []

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p in self.params:
            p.requires_grad_(True)


def gan_step(input, target, generator, discriminator, criterion, adversarial_loss_criterion, content_loss_criterion,
             optimizer_g, optimizer_d, num_class, sync_g=None, sync_d=None):
    """One iteration of train_seg_gan.py:182-233.  Returns device scalars (loss, iou, dice, closs, adv_g, adv_d)."""
    with _stage('ssg.G_forward'):
        generator_output = generator(input)                                    # :188
        generator_output = ops.nan_to_zero_(generator_output)                  # :190
    fused = isinstance(criterion, BCEDiceLoss) and isinstance(content_loss_criterion, nn.MSELoss) \
        and content_loss_criterion.reduction == 'mean'
    torch.cuda.nvtx.range_push('ssg.losses_and_D_forward_1') if _stage.ON else None
    if fused:
        res, msums = ops.seg_loss(generator_output, target, metric_first_channel=1, with_sums=True)   # :191-198 in one pass
        loss, content_loss, iou, dice = res[0], res[1], res[4], res[5]
    else:
        loss = criterion(generator_output, target)
        content_loss = content_loss_criterion(generator_output, target)
        m, msums = ops.seg_loss(generator_output.detach(), target, metric_first_channel=1, with_sums=True)
        iou, dice = m[4], m[5]
    if dp.is_dist():                                # whole-batch ratios: reduce the sums, not the ratios
        iou, dice = dp.reduce_metric_sums(msums)

    # The discriminator's PARAMETER gradients of this backward are dead in the reference: optimizer_g only steps
    # the generator and optimizer_d.zero_grad() (:225) clears them before anything reads them.  They are computed
    # all the same (the step does the reference's work); SSG_ELIDE_DEAD_D_GRADS=1 leaves them out (identical
    # results, -2.6 % step time at bs16/512^2 -- DESIGN.md) and bench.py then says so in its JSON line.
    with _params_frozen(discriminator, ELIDE_DEAD_D_GRADS):
        seg_discriminated = discriminator(generator_output)                    # :202
    adversarial_loss = _adv_loss(adversarial_loss_criterion, seg_discriminated, 1.0)
    perceptual_loss = loss + ALPA * content_loss + BETA * adversarial_loss     # :205
    adv_g = adversarial_loss.detach()
    torch.cuda.nvtx.range_pop() if _stage.ON else None

    optimizer_g.zero_grad()
    with _stage('ssg.G_step_backward'):
        if sync_g is not None:
            sync_g.begin()
        perceptual_loss.backward()
        if sync_g is not None:
            sync_g.finish()
    with _stage('ssg.G_optimizer'):
        _step_optimizer(optimizer_g, GRAD_CLIP)                                # :211-215

    with _stage('ssg.D_forward_2_3'):
        hr_discriminated = discriminator(target)                               # :217
        sr_discriminated = discriminator(generator_output.detach())            # :218
        adversarial_loss = _adv_loss(adversarial_loss_criterion, sr_discriminated, 0.0) + \
            _adv_loss(adversarial_loss_criterion, hr_discriminated, 1.0)

    optimizer_d.zero_grad()                                                    # :225 drops D grads of the G step
    with _stage('ssg.D_step_backward'):
        if sync_d is not None:
            sync_d.begin()
        adversarial_loss.backward()
        if sync_d is not None:
            sync_d.finish()
    with _stage('ssg.D_optimizer'):
        _step_optimizer(optimizer_d, GRAD_CLIP)                                # :229-233
    return dp.reduce_mean(loss), iou.detach(), dice.detach(), content_loss.detach(), adv_g, adversarial_loss.detach()


def train(epoch, config, train_loader, generator, discriminator, criterion, adversarial_loss_criterion,
          content_loss_criterion, optimizer_g, optimizer_d):
    avg_meters = {'loss': AverageMeter(), 'iou': AverageMeter(), 'dice': AverageMeter()}
    generator.train()
    discriminator.train()
    lr_val = optimizer_g.param_groups[0]['lr']
    print('generator learning rate {:d}: {:f}'.format(epoch, lr_val))
    num_class = int(config['num_classes'])
    sync_g, sync_d = dp.grad_syncs(generator, discriminator)

    for ori_img, input, target, targets, _ in train_loader:
        input = input.cuda(non_blocking=True)
        target = target.cuda(non_blocking=True)
        loss, iou, dice, _, _, _ = gan_step(input, target, generator, discriminator, criterion,
                                             adversarial_loss_criterion, content_loss_criterion, optimizer_g, optimizer_d,
                                             num_class, sync_g, sync_d)
        n = input.size(0)
        avg_meters['loss'].update(loss, n)
        avg_meters['iou'].update(iou, n)
        avg_meters['dice'].update(dice, n)

    return OrderedDict([('loss', _to_float(avg_meters['loss'].avg)), ('iou', _to_float(avg_meters['iou'].avg)),
                        ('dice', _to_float(avg_meters['dice'].avg))])


def validate(config, val_loader, generator, criterion):
    """train_seg_gan.py:253-294."""
    avg_meters = {'loss': AverageMeter(), 'iou': AverageMeter(), 'dice': AverageMeter()}
    generator.eval()
    with torch.no_grad():
        for ori_img, input, target, targets, _ in val_loader:
            input = input.cuda(non_blocking=True)
            target = target.cuda(non_blocking=True)
            output = ops.nan_to_zero_(generator(input))
            res = ops.seg_loss(output, target, metric_first_channel=1)
            loss = res[0] if isinstance(criterion, BCEDiceLoss) else criterion(output, target)
            n = input.size(0)
            avg_meters['loss'].update(loss, n)
            avg_meters['iou'].update(res[4], n)
            avg_meters['dice'].update(res[5], n)
    return OrderedDict([('loss', _to_float(avg_meters['loss'].avg)), ('iou', _to_float(avg_meters['iou'].avg)),
                        ('dice', _to_float(avg_meters['dice'].avg))])
