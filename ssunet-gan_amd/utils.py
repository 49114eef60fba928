"""Host bookkeeping (mirrors the reference's scripts/utils.py:58-74)."""


class AverageMeter(object):
    """Computes and stores the average and current value.  Accepts python floats or 0-dim
    device tensors; tensor values are accumulated on the device (no host sync per update)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count


def str2bool(v):
    if v.lower() in ['true', '1']:
        return True
    if v.lower() in ['false', '0']:
        return False
    raise ValueError('boolean value expected')


def count_params(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
