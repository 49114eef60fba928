import sys, time, torch
import ssunet_gan_amd as S
dev = torch.device('cuda')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(16, 64, 64, 512, 3), (16, 128, 128, 256, 3), (16, 256, 256, 128, 3), (16, 512, 512, 32, 3)]
for (n, ci, co, hw, k) in shapes:
    x = S.ops.new_nhwc(n, ci, hw, hw, dev); x.normal_()
    w = torch.randn(co, ci, k, k, device=dev) * 0.05
    fl = 2.0 * n * hw * hw * ci * co * k * k
    t1 = timeit(lambda: S.ops._conv_fwd_impl(x, None, w, None, 1, k // 2, 0, 0.0))
    print('%4d->%4d @%3d: fwd %6.3f ms %6.1f TF' % (ci, co, hw, t1, fl/t1/1e9))
