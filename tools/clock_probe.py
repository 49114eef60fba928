"""In-kernel shader clock of conv_igemm_halo_kernel's main loop (cdna guide rule 28 / MI355X_MICROARCH 'DVFS give-back' item 6):
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped around the K loop of every workgroup after >= 2 s of
back-to-back launches on random data; median over workgroups.

Needs the diagnostic build of the library (no stamp executes in the product build):
  cd ssunet-gan_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSG_CLOCK_PROBE -c conv_igemm_halo.hip -o /tmp/halo_probe.o
  hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v conv_igemm_halo.o) /tmp/halo_probe.o -o ../libssunet_probe.so
  python tools/clock_probe.py

`python tools/clock_probe.py x3` stamps the pre-split bf16x3 kernel instead (conv_igemm_halo_x3.hip built with the same define
and linked into the same probe library): does the chip hold its clock when the loop runs on the bf16 pipe?
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SSG_LIB_PATH', os.path.join(ROOT, 'ssunet-gan_amd', 'libssunet_probe.so'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ssunet_gan_amd as S  # noqa: E402
from ssunet_gan_amd import ops, _lib  # noqa: E402
from ssunet_gan_amd._lib import ACT_NONE  # noqa: E402

dev = 'cuda'
lib = _lib.load()
probe = torch.zeros(2 * 65536, dtype=torch.int64, device=dev)
X3 = len(sys.argv) > 1 and sys.argv[1] == 'x3'
ops.MFMA_SPLIT = X3
rc = getattr(lib, 'ssg_debug_set_probe_buffer_x3' if X3 else 'ssg_debug_set_probe_buffer')(C.c_void_p(probe.data_ptr()))
assert rc == 0, rc
torch.manual_seed(0)
for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.5:                 # >= 2 s of back-to-back launches: the clock has settled
        for _ in range(20):
            y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
        torch.cuda.synchronize(); n += 20
    dt = (time.perf_counter() - t0) / n
    p = probe.cpu().view(-1, 2)
    p = p[p[:, 1] > 0]
    clk = (p[:, 0].double() / p[:, 1].double() * 100e6).median().item() / 1e9
    flops = 2.0 * 16 * hw * hw * ci * co * 9
    print('cin%d cout%d %dx%d: %.1f TFLOP/s (wall, incl. launch), in-kernel clock %.2f GHz over %d workgroups, '
          '%s MFMA peak at that clock %.1f TFLOP/s' % (ci, co, hw, hw, flops / dt / 1e12, clk, p.shape[0], 'bf16' if X3 else 'fp32',
                                                      256 * 4 * (1024 if X3 else 64) * clk / 1e3), flush=True)
    probe.zero_()
