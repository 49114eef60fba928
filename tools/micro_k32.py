"""The 32-channel-chunk split kernel (conv_igemm_halo_k32.hip) against fp64 and against the 16-channel x3 kernel: accuracy on
small problems (forced with ssg_conv_set_k32_mode(2)), bias / residual / activation / batch-norm partial rows, concat inputs,
ragged grids; then time per bench shape, k32 off / on back to back."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
from ssunet_gan_amd import ops, _lib
from ssunet_gan_amd._lib import ACT_NONE, ACT_RELU, ACT_LRELU, call
dev = 'cuda'
torch.manual_seed(0)
ops.MFMA_SPLIT = True


def labels_of(fn):
    ops.PROFILE = []
    try:
        r = fn()
        torch.cuda.synchronize()
        return r, [rec[0] for rec in ops.PROFILE]
    finally:
        ops.PROFILE = None


bad = 0
for (c1, c2, co, h, w, nb, bias, res, act) in [
        (128, 0, 128, 64, 64, 2, False, False, ACT_NONE), (64, 0, 64, 64, 64, 2, True, False, ACT_RELU), (128, 64, 64, 48, 80, 2, False, True, ACT_LRELU),
        (64, 0, 128, 37, 45, 3, True, True, ACT_NONE), (256, 0, 256, 32, 32, 2, False, False, ACT_NONE), (32, 32, 128, 19, 33, 1, False, False, ACT_NONE),
        (64, 0, 192, 24, 40, 2, False, False, ACT_NONE)]:
    ci = c1 + c2
    xc = torch.randn(nb, ci, h, w) * 1.5 + 0.3; wc = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
    bc = torch.randn(co) if bias else None
    rc = torch.randn(nb, co, h, w) if res else None
    ref = F.conv2d(xc.double(), wc.double(), bc.double() if bias else None, 1, 1)
    pre = ref.clone()
    if res:
        ref = ref + rc.double()
    if act == ACT_RELU:
        ref = F.relu(ref)
    elif act == ACT_LRELU:
        ref = F.leaky_relu(ref, 0.2)
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    wd = wc.to(dev); bd = bc.to(dev) if bias else None; rd = ops.to_nhwc(rc.to(dev)) if res else None
    outs = {}
    for mode in (0, 2):
        call('ssg_conv_set_k32_mode', mode)
        want_bn = (not res) and act == ACT_NONE
        r, labels = labels_of(lambda: ops._conv_fwd_impl(x1, x2, wd, bd, 1, 1, act, 0.2, res=rd, want_bn=want_bn))
        y, part = r if want_bn else (r, None)
        outs[mode] = (y.cpu().double(), part, labels)
    ops.MFMA_SPLIT = False
    y32 = ops._conv_fwd_impl(x1, x2, wd, bd, 1, 1, act, 0.2, res=rd).cpu().double()
    ops.MFMA_SPLIT = True
    rms = lambda t: (t - ref).pow(2).mean().sqrt().item()
    print('   rms: fp32-MFMA %.3e  x3 %.3e  k32 %.3e | max: fp32-MFMA %.3e' % (rms(y32), rms(outs[0][0]), rms(outs[2][0]), (y32 - ref).abs().max().item()))
    scale = ref.abs().max().item()
    e0 = (outs[0][0] - ref).abs().max().item(); e2 = (outs[2][0] - ref).abs().max().item()
    ok = e2 <= max(2 * e0, 2e-6 * scale) and any('k32' in l for l in outs[2][2])
    msg = ''
    if outs[2][1] is not None:
        msg = ''
        for mode in (0, 2):
            if outs[mode][1] is None:
                continue
            part = outs[mode][1].cpu()
            s1 = part[:, 0, :].sum(0); s2 = part[:, 1, :].sum(0)
            pre = outs[mode][0]                           # the statistics are of the fp32 outputs the kernel produced
            cnt = pre.numel() / pre.shape[1]
            r1 = pre.sum((0, 2, 3)); r2 = (pre * pre).sum((0, 2, 3))
            mean_e = ((s1 - r1) / cnt).abs().max().item()                  # error of the mean, absolute
            var_ref = r2 / cnt - (r1 / cnt) ** 2; var_k = s2 / cnt - (s1 / cnt) ** 2
            var_e = ((var_k - var_ref).abs() / var_ref).max().item()        # relative error of the variance
            msg += ' [%s stats: mean err %.2e, var rel err %.2e, rows %d]' % ('k32' if mode else 'x3', mean_e, var_e, part.shape[0])
            if mode == 2:
                ok = ok and mean_e < 1e-6 and var_e < 2e-6
    print('%3d+%-3d->%-3d %dx%d n%d bias%d res%d act%d: x3 max err %.3e | k32 %.3e (of max|y| %.2f)%s  %s  %s' % (
        c1, c2, co, h, w, nb, bias, res, act, e0, e2, scale, msg, [l for l in outs[2][2] if 'conv' in l], 'ok' if ok else 'BAD'), flush=True)
    bad += 0 if ok else 1

# input gradient (stride 1) through the same kernel
for (ci, co, h, w, nb) in [(128, 128, 48, 64, 2), (64, 128, 40, 40, 2)]:
    wc = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5); dyc = torch.randn(nb, co, h, w)
    ref = F.conv_transpose2d(dyc.double(), wc.double(), None, 1, 1)
    dy = ops.to_nhwc(dyc.to(dev)); wd = wc.to(dev)
    outs = {}
    for mode in (0, 2):
        call('ssg_conv_set_k32_mode', mode)
        r, labels = labels_of(lambda: ops._conv_dgrad_impl(dy, wd, 1, 1, h, w, 0, ci))
        outs[mode] = (r.cpu().double(), labels)
    e0 = (outs[0][0] - ref).abs().max().item(); e2 = (outs[2][0] - ref).abs().max().item()
    ok = e2 <= max(2 * e0, 2e-6 * ref.abs().max().item()) and any('k32' in l for l in outs[2][1])
    print('dgrad %d<-%d %dx%d: x3 %.3e | k32 %.3e  %s %s' % (ci, co, h, w, e0, e2, outs[2][1], 'ok' if ok else 'BAD'), flush=True)
    bad += 0 if ok else 1
print('accuracy: %d bad' % bad, flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'acc':
    sys.exit(1 if bad else 0)

call('ssg_conv_set_k32_mode', 1)
for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (384, 384, 64), (64, 128, 256), (512, 512, 32), (192, 64, 512), (384, 128, 256), (128, 64, 512), (256, 128, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    fl = 2 * 9 * ci * co * 16 * hw * hw
    res = {0: [], 1: []}
    lab = {}
    for rnd in range(3):
        for mode in (0, 1):
            call('ssg_conv_set_k32_mode', mode)
            _, lab[mode] = labels_of(lambda: ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0))
            for _ in range(2):
                y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 8)
    t0 = min(res[0]); t1 = min(res[1])
    print('%4d->%-4d@%-3d x3 %.3f ms %.1f TF | k32 %.3f ms %.1f TF  (x%.2f)  %s' % (ci, co, hw, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1,
                                                                                   [l for l in lab[1] if 'conv' in l]), flush=True)
call('ssg_conv_set_k32_mode', 1)
