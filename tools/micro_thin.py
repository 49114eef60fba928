"""Micro-benchmark of the thin 4x4x1-MFMA kernels at 16 x 512^2 (for rocprofv3 counter passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
torch.manual_seed(0)
for (ci, co) in [(3, 64), (64, 3)]:
    x = ops.to_nhwc(torch.randn(16, ci, 512, 512, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev)
    dy = ops.to_nhwc(torch.randn(16, co, 512, 512, device=dev))
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    torch.cuda.synchronize()
print('done')
