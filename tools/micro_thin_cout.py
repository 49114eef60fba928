"""thin4_cout (wide input, <= 8 output channels) against its ablation builds libssunet_exp{3,4}.so (-DSSG_T4_EXP=3: no MFMAs,
4: every row load hits the same cache-resident KiB); built like tools/micro_thin_exp.py describes."""
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
out = []
for (ci, co, hw) in [(64, 3, 512), (128, 4, 512), (256, 8, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev)
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append('%%d->%%d@%%d %%.3f ms %%.2f TB/s' %% (ci, co, hw, ms, 16 * hw * hw * 4 * (ci + 4) / ms / 1e9))
print('%%-20s %%s' %% (os.path.basename(os.environ.get('SSG_LIB_PATH', 'shipped')), ' | '.join(out)))
''' % ROOT
for lib in [None] + ['libssunet_exp%s.so' % a for a in sys.argv[1:]]:
    env = dict(os.environ)
    if lib:
        env['SSG_LIB_PATH'] = os.path.join(ROOT, 'ssunet-gan_amd', lib)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-600:], flush=True)
