"""LDS-DMA weight-gradient kernel, fp32 MFMA vs split operands, on the shapes of the step that use it (stride-2 3x3 of D, 1x1
shortcuts and up-path convs, thin-Cin 3x3)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
dev = 'cuda'
torch.manual_seed(0)
for (ci, co, hw, k, s) in [(128, 128, 256, 3, 2), (256, 256, 128, 3, 2), (512, 512, 64, 3, 2), (64, 64, 512, 3, 2), (192, 64, 512, 1, 1), (384, 128, 256, 1, 1),
                            (8, 256, 256, 3, 1), (16, 512, 128, 3, 1)]:
    oh = hw // s
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); dy = ops.to_nhwc(torch.randn(16, co, oh, oh, device=dev))
    fl = 2 * k * k * ci * co * 16 * oh * oh
    res = []
    for rnd in range(3):
        for split in (False, True):
            ops.MFMA_SPLIT = split
            for _ in range(2):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, k, k), s, k // 2)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, k, k), s, k // 2)
            e1.record(); torch.cuda.synchronize()
            res.append((split, e0.elapsed_time(e1) / 6))
    t32 = min(t for s_, t in res if not s_); t3 = min(t for s_, t in res if s_)
    print('%4d->%-4d@%-3d k%d s%d fp32 %.3f ms %.1f TF | split %.3f ms %.1f TF-equivalent  (x%.2f)' % (ci, co, hw, k, s, t32, fl / t32 / 1e9, t3, fl / t3 / 1e9, t32 / t3), flush=True)
