"""thin-Cin 3x3 convs (4-channel input): the 32x32x2-MFMA kernel (default) against the 4x4x1 kernel (SSG_THIN32=0), same process
is not possible (the switch is read once), so the script re-runs itself.  Usage: python tools/micro_thin32.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE, ACT_RELU
dev = 'cuda'
torch.manual_seed(0)
out = []
for (ci, co, hw, res) in [(4, 64, 512, False), (3, 64, 512, False), (4, 128, 512, False), (3, 64, 512, True), (3, 128, 256, False)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev)
    r = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev)) if res else None
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, res=r)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0, res=r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append('%%d->%%d@%%d%%s %%.3f ms %%.2f TB/s' %% (ci, co, hw, '+res' if res else '', ms, 16 * hw * hw * 4 * (co * (2 if res else 1) + 4) / ms / 1e9))
print('SSG_THIN32=%%s  %%s' %% (os.environ.get('SSG_THIN32', '1'), ' | '.join(out)))
''' % ROOT
for v, extra in (('1', {}), ('0', {}), ('1', {'SSG_THIN32_NT': '0'})):
    r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, SSG_THIN32=v, **extra), capture_output=True, text=True)
    print(str(extra), r.stdout.strip() or r.stderr[-600:], flush=True)
