"""A few launches of the dense 3x3 kernels at bench shapes (for rocprofv3 --pmc SQ counter passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
torch.manual_seed(0)
for (ci, co, hw) in [(128, 128, 256), (64, 64, 512)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev))
    for _ in range(4):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    torch.cuda.synchronize()
print('done')
