"""Input gradient of a 3x3 stride-2 conv: four per-class launches (conv_igemm_dma_x3) vs the merged parity launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
dev = 'cuda'
torch.manual_seed(0)
for (c, hw) in [(64, 512), (128, 256), (256, 128), (512, 64)]:
    w = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    dy = ops.to_nhwc(torch.randn(16, c, hw // 2, hw // 2, device=dev))
    fl = 2 * 9 * c * c * 16 * (hw // 2) ** 2
    res = {}
    for rnd in range(3):
        for merge in (False, True):
            ops.PARITY_MERGE = merge
            for _ in range(2):
                g = ops._conv_dgrad_impl(dy, w, 2, 1, hw, hw, 0, c)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                g = ops._conv_dgrad_impl(dy, w, 2, 1, hw, hw, 0, c)
            e1.record(); torch.cuda.synchronize()
            res[merge] = min(res.get(merge, 1e9), e0.elapsed_time(e1) / 6)
    print('%4d ch -> %dx%d: 4 launches %.3f ms %.1f TF | merged %.3f ms %.1f TF  (x%.2f)' % (c, hw, hw, res[False], fl / res[False] / 1e9, res[True], fl / res[True] / 1e9, res[False] / res[True]), flush=True)
