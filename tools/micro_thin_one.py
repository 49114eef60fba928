"""One thin-Cin conv shape, 10 launches (for rocprofv3 counter passes): python tools/micro_thin_one.py [cin cout hw]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
ci, co, hw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 64, 512)
dev = 'cuda'
torch.manual_seed(0)
x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
w = torch.randn(co, ci, 3, 3, device=dev)
for _ in range(10):
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
torch.cuda.synchronize()
print('done')
