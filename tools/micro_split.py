"""fp32-MFMA halo kernel vs the bf16x3 split-operand kernel: accuracy against an fp64 reference and time, same process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
torch.manual_seed(0)
# accuracy on a small problem with an fp64 CPU reference
for (ci, co, hw, nb) in [(128, 128, 64, 2), (64, 64, 64, 2), (192, 64, 64, 2)]:
    xc = torch.randn(nb, ci, hw, hw); wc = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
    ref = torch.nn.functional.conv2d(xc.double(), wc.double(), None, 1, 1)
    x = ops.to_nhwc(xc.to(dev)); w = wc.to(dev)
    outs = {}
    for split in (False, True):
        ops.MFMA_SPLIT = split
        outs[split] = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0).cpu().double()
    scale = ref.abs().max().item()
    e32 = (outs[False] - ref).abs(); e3 = (outs[True] - ref).abs()
    print('%d->%d@%d: fp32 MFMA max err %.3e (rms %.3e) | split max err %.3e (rms %.3e) | relative to max|y| = %.2f' % (
        ci, co, hw, e32.max().item(), e32.pow(2).mean().sqrt().item(), e3.max().item(), e3.pow(2).mean().sqrt().item(), scale), flush=True)
for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (384, 384, 64), (64, 128, 256), (512, 512, 32), (192, 64, 512), (384, 128, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    fl = 2 * 9 * ci * co * 16 * hw * hw
    res = []
    for rnd in range(3):
        for split in (False, True):
            ops.MFMA_SPLIT = split
            for _ in range(2):
                y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
            e1.record(); torch.cuda.synchronize()
            res.append((split, e0.elapsed_time(e1) / 6))
    t32 = min(t for s_, t in res if not s_); t3 = min(t for s_, t in res if s_)
    print('%4d->%-4d@%-3d fp32 %.3f ms %.1f TF | split %.3f ms %.1f TF-equivalent  (x%.2f)' % (ci, co, hw, t32, fl / t32 / 1e9, t3, fl / t3 / 1e9, t32 / t3), flush=True)

print('weight gradient:')
for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (384, 384, 64), (64, 128, 256), (192, 64, 512)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev))
    fl = 2 * 9 * ci * co * 16 * hw * hw
    res = []
    for rnd in range(3):
        for split in (False, True):
            ops.MFMA_SPLIT = split
            for _ in range(2):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
            e1.record(); torch.cuda.synchronize()
            res.append((split, e0.elapsed_time(e1) / 6))
    t32 = min(t for s_, t in res if not s_); t3 = min(t for s_, t in res if s_)
    print('%4d->%-4d@%-3d fp32 %.3f ms %.1f TF | split %.3f ms %.1f TF-equivalent  (x%.2f)' % (ci, co, hw, t32, fl / t32 / 1e9, t3, fl / t3 / 1e9, t32 / t3), flush=True)
