"""Narrow-Cout 3x3 layers of the step (SPADE's x -> map convs): fp32 256 x 32 register kernel vs the narrow k32 tiles, 16 images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE

def t(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    return best

for (ci, co, hw) in [(512, 16, 128), (768, 24, 64), (256, 16, 256), (1024, 32, 64), (128, 16, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
    fl = 2 * 9 * ci * co * 16 * hw * hw
    res = []
    for split in (False, True):
        ops.MFMA_SPLIT = split
        ops.PROFILE = []
        ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
        lab = ops.PROFILE[0][0]; ops.PROFILE = None
        ms = t(lambda: ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0))
        res.append('%s %.3f ms %.1f TF' % (lab, ms, fl / ms / 1e9))
    print('%4d->%-3d @%-3d  %s | %s' % (ci, co, hw, res[0], res[1]), flush=True)
