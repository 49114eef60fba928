"""One shape of the split-operand kernel, a few launches (for rocprofv3 --pmc passes).  argv: ci co hw split(0/1)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
ci, co, hw, split = [int(v) for v in sys.argv[1:5]]
ops.MFMA_SPLIT = bool(split)
x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
for _ in range(4):
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
torch.cuda.synchronize()
print('done')
