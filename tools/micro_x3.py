"""The round-3 split kernels that still carry stride-2 / 1x1 layers: forward, merged-parity input gradient and weight gradient of D's stride-2 convs,
a 1x1 shortcut; TFLOP/s of each.  `python tools/micro_x3.py libA.so libB.so` runs each library in a child process (same box)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    import ssunet_gan_amd as S
    from ssunet_gan_amd import ops
    from ssunet_gan_amd._lib import ACT_NONE
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
    for (ci, co, hw, k, s) in [(128, 128, 256, 3, 2), (256, 256, 128, 3, 2), (64, 64, 512, 3, 2), (192, 64, 512, 1, 1), (512, 256, 128, 1, 1)]:
        oh = hw // s
        x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, k, k, device='cuda') / (k * ci ** 0.5)
        dy = ops.to_nhwc(torch.randn(16, co, oh, oh, device='cuda'))
        fl = 2 * k * k * ci * co * 16 * oh * oh
        a = t(lambda: ops._conv_fwd_impl(x, None, w, None, s, k // 2, ACT_NONE, 0.0))
        b = t(lambda: ops._conv_dgrad_impl(dy, w, s, k // 2, hw, hw, 0, ci))
        c = t(lambda: ops._conv_wgrad_impl(x, None, dy, (co, ci, k, k), s, k // 2))
        print('  %4d->%-4d@%-3d k%d s%d fwd %.1f | dgrad %.1f | wgrad %.1f TF' % (ci, co, hw, k, s, fl / a / 1e9, fl / b / 1e9, fl / c / 1e9), flush=True)
    sys.exit(0)
for rnd in range(2):
    for name in sys.argv[1:]:
        print(name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, SSG_LIB_PATH=os.path.join(ROOT, 'ssunet-gan_amd', name)))
