"""A/B of the thin-Cin kernel's prefetch depth (SSG_THIN4_PF = rows in flight) at 16 x 512^2: one subprocess per value."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
torch.manual_seed(0)
for (ci, co, hw) in [(3, 64, 512), (4, 64, 512), (8, 128, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev)
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('PF=%%s  %%d->%%d @%%d^2: %%.3f ms  %%.2f TB/s (output bytes)' %% (os.environ.get('SSG_THIN4_PF'), ci, co, hw, ms, 16 * hw * hw * co * 4 / ms / 1e9))
    if ci == 4:
        res = ops.new_nhwc(16, co, hw, hw, dev); res.normal_()
        for _ in range(3):
            y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, res=res)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, res=res)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('        with residual: %%.3f ms  %%.2f TB/s (residual read + output written)' %% (ms, 2 * 16 * hw * hw * co * 4 / ms / 1e9))
''' % ROOT
for pf in ('1', '2', '4'):
    r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, SSG_THIN4_PF=pf), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
