"""Ablations of thin4_cin against the shipped kernel: diagnostic builds libssunet_exp{1,2}.so (1 = no MFMAs, 2 = no output stores),
built by hand:  cd ssunet-gan_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSG_T4_EXP=n -c conv_thin4.hip -o /tmp/t4.o
                && hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v conv_thin4.o) /tmp/t4.o -o ../libssunet_expn.so
Round-2 result on 16 x 512^2, 4 -> 64: shipped 0.474 ms, no MFMAs 0.367, no stores 0.300, non-temporal stores 0.402 (now the default
for outputs >= 256 MB): neither the MFMA chain nor the store stream is the limiter on its own; the waves sit in s_waitcnt 55 %."""
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
x = ops.to_nhwc(torch.randn(16, 4, 512, 512, device=dev)); w = torch.randn(64, 4, 3, 3, device=dev)
for _ in range(3):
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
e1.record(); torch.cuda.synchronize()
print('%%s: %%.3f ms' %% (os.path.basename(os.environ.get('SSG_LIB_PATH', 'shipped')), e0.elapsed_time(e1) / 20))
''' % ROOT
for lib in (None, 'libssunet_exp1.so', 'libssunet_exp2.so'):
    env = dict(os.environ)
    if lib:
        env['SSG_LIB_PATH'] = os.path.join(ROOT, 'ssunet-gan_amd', lib)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-400:])
