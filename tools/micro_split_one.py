"""One shape of the split-operand kernels, a few launches (for rocprofv3 --pmc passes).
argv: ci co hw split(0/1) [op = conv | wgrad] [k32 = 1 | 0]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE, call
ci, co, hw, split = [int(v) for v in sys.argv[1:5]]
op = sys.argv[5] if len(sys.argv) > 5 else 'conv'
k32 = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ops.MFMA_SPLIT = bool(split)
call('ssg_conv_set_k32_mode', 1 if k32 else 0); call('ssg_wgrad_set_k32_mode', 1 if k32 else 0)
x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device='cuda'))
for _ in range(4):
    if op == 'conv':
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    else:
        y = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
torch.cuda.synchronize()
print('done')
