"""clip_gradient, MI355X-native (mirrors the reference's scripts/srgan_utils.py:186-195)."""
from . import _lib
from ._lib import call, ptr, stream_ptr


def clip_gradient(optimizer, grad_clip):
    """Clamp every parameter gradient element-wise to [-grad_clip, grad_clip], in place."""
    for group in optimizer.param_groups:
        for param in group['params']:
            g = param.grad
            if g is None:
                continue
            _lib.require_gpu(g)
            if not g.is_contiguous():
                raise ValueError('clip_gradient: non-contiguous gradient')
            call('ssg_clamp_f32', ptr(g), g.numel(), -float(grad_clip), float(grad_clip), stream_ptr())
