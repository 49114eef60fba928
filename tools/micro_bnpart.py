"""Cost of the batch-norm statistics epilogue of the k32 conv tiles: the same launch with and without bnpart, 16 images."""
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

for (ci, co, hw) in [(64, 128, 512), (64, 64, 512), (64, 128, 256), (128, 128, 256), (256, 256, 128)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
    out = ops.new_nhwc(16, co, hw, hw, 'cuda')
    fl = 2 * 9 * ci * co * 16 * hw * hw
    a = t(lambda: ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, out=out))
    b = t(lambda: ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, out=out, want_bn=True))
    print('%4d->%-4d@%-3d plain %.3f ms %.1f TF | with statistics %.3f ms %.1f TF (%+.1f %%)' % (ci, co, hw, a, fl / a / 1e9, b, fl / b / 1e9, 100 * (b / a - 1)), flush=True)
