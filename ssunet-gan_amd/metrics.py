"""Metrics, MI355X-native (mirrors the reference's scripts/metrics.py:6-35).  The reference
moves tensors to numpy (a host sync per call); here the sums are reduced on the device by the
fused loss kernel and only the final scalar is read back when the caller asks for a float."""
from . import ops


def _metrics(output, target):
    return ops.seg_loss(output.detach(), target, metric_first_channel=0)


def iou_score(output, target):
    """Hard IoU at 0.5 on sigmoid(output) over the whole tensor, smooth 1e-5 (metrics.py:6-22)."""
    return float(_metrics(output, target)[4].item())


def dice_coef(output, target):
    """Soft Dice over the whole flattened tensor, smooth 1e-5 (metrics.py:25-35)."""
    return float(_metrics(output, target)[5].item())
