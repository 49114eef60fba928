"""Bilinear x2 up-sampling forward / backward at the bench's decoder shapes: time, algorithmic bytes (1/4 read + 1 write per output
element forward, the mirror backward) and the rate they give.  `python tools/micro_spatial.py libA.so libB.so` runs each library
in a child process on the same box (A/B)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    import ssunet_gan_amd as S
    from ssunet_gan_amd import ops
    def t(fn):
        best = 1e9
        for _ in range(3):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8)
        return best
    for (c, hw) in [(384, 128), (96, 256), (128, 256), (256, 128), (64, 256), (512, 32)]:
        x = ops.to_nhwc(torch.randn(16, c, hw, hw, device='cuda'))
        dy = ops.to_nhwc(torch.randn(16, c, 2 * hw, 2 * hw, device='cuda'))
        by = 16 * hw * hw * c * 20.0
        a = t(lambda: ops._BilinearUp.forward(None, x))
        class Ctx: pass
        b = t(lambda: ops._BilinearUp.backward(None, dy))
        print('  c%-4d %3d->%-3d fwd %.3f ms %.2f TB/s | bwd %.3f ms %.2f TB/s' % (c, hw, 2 * hw, a, by / a / 1e9, b, by / b / 1e9), flush=True)
    sys.exit(0)
for rnd in range(2):
    for name in sys.argv[1:]:
        print(name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, SSG_LIB_PATH=os.path.join(ROOT, 'ssunet-gan_amd', name)))
