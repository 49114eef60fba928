"""Ablations of conv_igemm_halo_kernel against the shipped kernel: diagnostic builds libssunet_hexp{n}.so, built by hand:
  cd ssunet-gan_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSG_HALO_EXP=n -c conv_igemm_halo.hip -o /tmp/h.o
  && hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v conv_igemm_halo.o) /tmp/h.o -o ../libssunet_hexpn.so
(n: see SSG_HALO_EXP in conv_igemm_halo.hip; the ablated kernels compute wrong results, only their time is read).
Usage: python tools/micro_halo_exp.py [lib suffixes ...]   e.g.  1 2 3 4"""
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
for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (384, 384, 64)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append('%%d->%%d@%%d %%.3f ms %%.1f TF' %% (ci, co, hw, ms, 2 * 9 * ci * co * 16 * hw * hw / ms / 1e9))
print('%%-22s %%s' %% (os.path.basename(os.environ.get('SSG_LIB_PATH', 'shipped')), ' | '.join(out)))
''' % ROOT
libs = [None] + ['libssunet_hexp%s.so' % a for a in sys.argv[1:]] + [None]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env['SSG_LIB_PATH'] = os.path.join(ROOT, 'ssunet-gan_amd', lib)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-600:], flush=True)
