"""Weight gradient of 4-channel-input 3x3 convs: wgrad32_cin (32x32x2 MFMA, default) vs wgrad4<thin_cin> (SSG_WGRAD32=0)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import ssunet_gan_amd as S
from ssunet_gan_amd import ops
dev = 'cuda'
torch.manual_seed(0)
out = []
for (ci, co, hw) in [(3, 64, 512), (4, 128, 512), (3, 128, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev))
    for _ in range(3):
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append('%%d->%%d@%%d %%.3f ms %%.2f TB/s' %% (ci, co, hw, ms, 16 * hw * hw * 4 * (co + 4) / ms / 1e9))
print('SSG_WGRAD32=%%s  %%s' %% (os.environ.get('SSG_WGRAD32', '1'), ' | '.join(out)))
''' % ROOT
for v in ('1', '0', '1'):
    r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, SSG_WGRAD32=v), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-600:], flush=True)
