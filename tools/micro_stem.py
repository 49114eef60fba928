"""The bf16 stem conv (csrc/conv_stem_bf16.hip) at B4 / 4 x 3 x 1024^2: forward and weight gradient, ms and bytes moved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssunet_gan_amd as S
from ssunet_gan_amd import ops, bf16

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

for (n, hw, co) in [(4, 1024, 48), (4, 1024, 32), (16, 512, 48)]:
    x = ops.as_nhwc(torch.randn(n, 3, hw, hw, device='cuda'))
    w = torch.randn(co, 3, 3, 3, device='cuda', requires_grad=True)
    y = bf16.conv_thin(x, w, 2, (0, 1, 0, 1))
    dy = torch.randn_like(y)
    fwd = t(lambda: bf16._ConvThin.forward(type('C', (), {'save_for_backward': lambda *a: None})(), x, w, 2, (0, 1, 0, 1)))
    def bwd():
        yy = bf16.conv_thin(x, w, 2, (0, 1, 0, 1)); yy.backward(dy)
    both = t(bwd)
    out_mb = y.numel() * 2 / 1e6
    print('n%d %d^2 -> %d ch: forward %.3f ms (output %.0f MB), forward + weight gradient %.3f ms' % (n, hw, co, fwd, out_mb, both), flush=True)
