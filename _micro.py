import sys, time, torch
import ssunet_gan_amd as S
from ssunet_gan_amd._lib import call, ptr, stream_ptr
dev = torch.device('cuda')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
scratch = torch.empty(4096 * 256, device=dev)
for blocks in (256, 512, 768, 1024):
    it = 2000
    ms = timeit(lambda: call('ssg_tool_mfma_peak_f32', ptr(scratch), blocks, it, stream_ptr()), 5)
    print('mfma peak blocks %d: %.1f TF' % (blocks, blocks * 4 * it * 16 * 4096 / ms / 1e9))
a = torch.empty(1 << 28, device=dev); b = torch.empty(1 << 28, device=dev)
ms = timeit(lambda: call('ssg_tool_copy_f32', ptr(a), ptr(b), a.numel(), stream_ptr()), 5)
print('copy: %.2f TB/s (read+write)' % (2 * a.numel() * 4 / ms / 1e9))
del a, b
shapes = [(16, 64, 64, 512, 3), (16, 128, 128, 256, 3), (16, 256, 256, 128, 3), (16, 384, 384, 64, 3), (16, 512, 512, 32, 3), (16, 192, 64, 512, 3), (16, 768, 768, 16, 3)]
for (n, ci, co, hw, k) in shapes:
    x = S.ops.new_nhwc(n, ci, hw, hw, dev); x.normal_()
    w = torch.randn(co, ci, k, k, device=dev) * 0.05
    dy = S.ops.new_nhwc(n, co, hw, hw, dev); dy.normal_()
    fl = 2.0 * n * hw * hw * ci * co * k * k
    t1 = timeit(lambda: S.ops._conv_fwd_impl(x, None, w, None, 1, k // 2, 0, 0.0))
    t2 = timeit(lambda: S.ops._conv_dgrad_impl(dy, w, 1, k // 2, hw, hw, 0, ci))
    t3 = timeit(lambda: S.ops._conv_wgrad_impl(x, None, dy, w.shape, 1, k // 2))
    print('%4d->%4d @%3d: fwd %6.3f ms %6.1f TF | dgrad %6.3f ms %6.1f TF | wgrad %6.3f ms %6.1f TF' % (ci, co, hw, t1, fl/t1/1e9, t2, fl/t2/1e9, t3, fl/t3/1e9))
