"""Stage-1 supervised training step, MI355X-native (mirrors train()/validate() of the reference's
scripts/train.py:68-199 -- SURVEY.md 8f row N1).  Same signature and the same per-iteration order,
including its quirk: parameters are clamped to +-config['clip'] AFTER the forward pass and BEFORE
backward (train.py:111-115), so backward runs on the clamped weights."""
from collections import OrderedDict

import torch

from . import dp, ops
from .losses import BCEDiceLoss
from .optim import clamp_parameters_, clip_adam_step
from .utils import AverageMeter


def _to_float(v):
    return float(v.item()) if torch.is_tensor(v) else float(v)


def _loss_and_metrics(config, model, criterion, input, target, num_class):
    if config['deep_supervision'] and str(config['deep_supervision']) != 'False':
        outputs = model(input)                                               # train.py:84-95
        loss = 0
        for output in outputs:
            loss = loss + criterion(output, target)
        loss = loss / len(outputs)
        m = ops.seg_loss(outputs[-1].detach(), target, metric_first_channel=0)
        return loss, m[4], m[5]
    output = ops.nan_to_zero_(model(input))                                  # :98-100
    if isinstance(criterion, BCEDiceLoss):
        res = ops.seg_loss(output, target, metric_first_channel=1)          # loss + IoU/Dice on channels 1: in one pass
        return res[0], res[4], res[5]
    m = ops.seg_loss(output.detach(), target, metric_first_channel=1)
    return criterion(output, target), m[4], m[5]


def train(epoch, config, train_loader, model, criterion, optimizer, cnn_optimizer):
    avg_meters = {'loss': AverageMeter(), 'iou': AverageMeter(), 'dice': AverageMeter()}
    model.train()
    clip = float(config['clip'])
    print('learning rate {:d}: {:f}'.format(epoch, optimizer.param_groups[0]['lr']))
    num_class = int(config['num_classes'])
    sync = dp.grad_sync(model)
    params = [p for p in model.parameters()]
    for ori_img, input, target, targets, _ in train_loader:
        input = input.cuda(non_blocking=True)
        target = target.cuda(non_blocking=True)
        loss, iou, dice = _loss_and_metrics(config, model, criterion, input, target, num_class)
        clamp_parameters_(params, clip)                                      # :111-112
        optimizer.zero_grad()
        if sync is not None:
            sync.begin()
        loss.backward()
        if sync is not None:
            sync.finish()
        if type(optimizer) is torch.optim.Adam:
            clip_adam_step(optimizer, None)                                  # :115 (no gradient clipping in stage 1)
        else:
            optimizer.step()
            ops.bump_weight_epoch()
        if cnn_optimizer is not None and epoch > 1:                          # :117-119
            cnn_optimizer.step()
            ops.bump_weight_epoch()
        n = input.size(0)
        avg_meters['loss'].update(loss.detach(), n)
        avg_meters['iou'].update(iou.detach(), n)
        avg_meters['dice'].update(dice.detach(), n)
    return OrderedDict([('loss', _to_float(avg_meters['loss'].avg)), ('iou', _to_float(avg_meters['iou'].avg)),
                        ('dice', _to_float(avg_meters['dice'].avg))])


def validate(config, val_loader, model, criterion):
    avg_meters = {'loss': AverageMeter(), 'iou': AverageMeter(), 'dice': AverageMeter()}
    model.eval()
    num_class = int(config['num_classes'])
    with torch.no_grad():
        for ori_img, input, target, targets, _ in val_loader:
            input = input.cuda(non_blocking=True)
            target = target.cuda(non_blocking=True)
            loss, iou, dice = _loss_and_metrics(config, model, criterion, input, target, num_class)
            n = input.size(0)
            avg_meters['loss'].update(loss, n)
            avg_meters['iou'].update(iou, n)
            avg_meters['dice'].update(dice, n)
    return OrderedDict([('loss', _to_float(avg_meters['loss'].avg)), ('iou', _to_float(avg_meters['iou'].avg)),
                        ('dice', _to_float(avg_meters['dice'].avg))])
