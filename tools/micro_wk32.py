"""conv_wgrad_k32.hip against fp64 and against wgrad_halo_x3: accuracy on small problems (ragged widths, concat inputs, short
slabs), then time per bench shape, off / on back to back."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import call
dev = 'cuda'
torch.manual_seed(0)
ops.MFMA_SPLIT = True


def run(x1, x2, dy, shape, mode):
    call('ssg_wgrad_set_k32_mode', mode)
    ops.PROFILE = []
    try:
        dw = ops._conv_wgrad_impl(x1, x2, dy, shape, 1, 1)
        torch.cuda.synchronize()
        return dw, [r[0] for r in ops.PROFILE]
    finally:
        ops.PROFILE = None


bad = 0
for (c1, c2, co, h, w, nb) in [(64, 0, 64, 32, 32, 2), (128, 0, 128, 48, 64, 2), (128, 64, 64, 40, 50, 2), (64, 0, 192, 19, 33, 3), (256, 0, 128, 17, 70, 1),
                               (64, 64, 64, 64, 96, 4)]:
    ci = c1 + c2
    xc = torch.randn(nb, ci, h, w) * 1.5 + 0.3; dyc = torch.randn(nb, co, h, w)
    xr = xc.double().requires_grad_(False); wr = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, None, 1, 1).backward(dyc.double())
    ref = wr.grad
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    dy = ops.to_nhwc(dyc.to(dev))
    o0, l0 = run(x1, x2, dy, (co, ci, 3, 3), 0)
    o1, l1 = run(x1, x2, dy, (co, ci, 3, 3), 1)
    ops.MFMA_SPLIT = False
    o32, _ = run(x1, x2, dy, (co, ci, 3, 3), 0)
    ops.MFMA_SPLIT = True
    e = lambda t: ((t.cpu().double() - ref).abs().max().item(), (t.cpu().double() - ref).pow(2).mean().sqrt().item())
    e0, e1, e32 = e(o0), e(o1), e(o32)
    ok = any('k32' in l for l in l1) and e1[0] <= 2 * max(e0[0], e32[0]) and e1[1] <= 2 * max(e0[1], e32[1])
    print('%3d+%-3d->%-3d %dx%d n%d: fp32-MFMA max %.2e rms %.2e | x3 %.2e %.2e | k32 %.2e %.2e (max|dw| %.1f)  %s %s' % (
        c1, c2, co, h, w, nb, e32[0], e32[1], e0[0], e0[1], e1[0], e1[1], ref.abs().max().item(), [l for l in l1 if 'wgrad' in l and 'reduce' not in l], 'ok' if ok else 'BAD'), flush=True)
    bad += 0 if ok else 1
print('accuracy: %d bad' % bad, flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'acc':
    sys.exit(1 if bad else 0)

for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (384, 384, 64), (64, 128, 256), (192, 64, 512), (512, 512, 32), (384, 128, 256), (128, 64, 512)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev))
    fl = 2 * 9 * ci * co * 16 * hw * hw
    res = {0: [], 1: []}
    lab = {}
    for rnd in range(3):
        for mode in (0, 1):
            _, lab[mode] = run(x, None, dy, (co, ci, 3, 3), mode)
            for _ in range(2):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 6)
    t0 = min(res[0]); t1 = min(res[1])
    print('%4d->%-4d@%-3d x3 %.3f ms %.1f TF | k32 %.3f ms %.1f TF  (x%.2f)  %s' % (ci, co, hw, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1,
                                                                                   [l for l in lab[1] if 'wgrad' in l]), flush=True)
call('ssg_wgrad_set_k32_mode', 1)
