import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch
    import ssunet_gan_amd as S
    from ssunet_gan_amd import ops
    from ssunet_gan_amd._lib import ACT_NONE
    ops.MFMA_SPLIT = True
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
    for (ci, co, hw) in [(128, 128, 256), (256, 256, 128), (64, 64, 512), (192, 64, 512), (384, 384, 64)]:
        x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device='cuda')); w = torch.randn(co, ci, 3, 3, device='cuda') / (3 * ci ** 0.5)
        dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device='cuda'))
        fl = 2 * 9 * ci * co * 16 * hw * hw
        a = t(lambda: ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)); b = t(lambda: ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1))
        print('  %4d->%-4d@%-3d conv %.3f ms %.1f TF | wgrad %.3f ms %.1f TF' % (ci, co, hw, a, fl / a / 1e9, b, fl / b / 1e9), flush=True)
    sys.exit(0)
for rnd in range(2):
    for name in sys.argv[1:]:
        print(name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, SSG_LIB_PATH=os.path.join(ROOT, 'ssunet-gan_amd', name)))
