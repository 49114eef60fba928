"""Where a tile of conv_halo_k32_kernel spends its cycles: s_memtime stamps at kernel start / main-loop start / main-loop end / after
the output stores, median over workgroups.  Needs the diagnostic build (no stamp executes in the product build):
  cd ssunet-gan_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSSG_K32_PROBE -c conv_igemm_halo_k32.hip -o /tmp/k32_probe.o
  hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v conv_igemm_halo_k32.o) /tmp/k32_probe.o -o ../libssunet_probe.so"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('SSG_LIB_PATH', os.path.join(ROOT, 'ssunet-gan_amd', 'libssunet_probe.so'))
sys.path.insert(0, ROOT)
import torch
import ssunet_gan_amd as S
from ssunet_gan_amd import ops, _lib
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
lib = _lib.load()
SLOTS = 32
probe = torch.zeros(SLOTS * 65536, dtype=torch.int64, device=dev)
assert lib.ssg_debug_set_probe_buffer_k32(C.c_void_p(probe.data_ptr())) == 0
ops.MFMA_SPLIT = True
torch.manual_seed(0)
for (ci, co, hw) in [(64, 64, 512), (192, 64, 512), (128, 128, 256), (512, 512, 32)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev)); w = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.5:                 # >= 2 s of back-to-back launches: the clock has settled
        for _ in range(20):
            y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
        torch.cuda.synchronize()
    probe.zero_()
    y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    p = probe.cpu().view(-1, SLOTS)
    p = p[p[:, 3] > 0].double()
    d = lambda a, b: (p[:, b] - p[:, a]).median().item()
    clk = ((p[:, 3] - p[:, 0]) / (p[:, 5] - p[:, 4]).clamp(min=1) * 100e6).median().item() / 1e9
    print('cin%d cout%d %dx%d: %d workgroups; per tile (median cycles): prologue %.0f, main loop %.0f (%.0f per step; MFMA issue alone: 3072), epilogue + store drain %.0f, '
          'total %.0f; in-kernel clock %.2f GHz (s_memtime / s_memrealtime)' % (ci, co, hw, hw, p.shape[0], d(0, 1), d(1, 2), d(1, 2) / (ci // 32 * 9), d(2, 3), d(0, 3), clk), flush=True)
    print('    prologue split: entry -> loads issued %.0f, -> pixels landed %.0f, -> split + written %.0f' % (d(7, 0), d(0, 6), d(6, 1)), flush=True)
    print('    epilogue split: loop end -> non-finite check done %.0f, -> stores issued %.0f, -> stores acknowledged %.0f' % (d(2, 24), d(24, 25), d(25, 3)), flush=True)
    nst = ci // 32 * 9
    nw = 8 if (p[:, 8 + 2 * 7] + p[:, 9 + 2 * 7]).median().item() > 0 else 4
    own = [p[:, 8 + 2 * w].median().item() / nst for w in range(nw)]
    bar = [p[:, 9 + 2 * w].median().item() / nst for w in range(nw)]
    print('    per K-step and wave (median over workgroups), cycles on its own vmcnt / LDS waits at the top of the step: ' + ' '.join('%.0f' % v for v in own), flush=True)
    print('    ... and inside the s_barrier (waiting for the slowest wave):                                             ' + ' '.join('%.0f' % v for v in bar), flush=True)
