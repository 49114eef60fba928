"""HBM streaming rates of this device by access mix: pure write (fill), pure read (reduction), copy.  1-GiB buffers."""
import torch, time
dev = 'cuda'
n = 1 << 28                                     # 1 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.zero_()); print('fill  (write only): %.3f ms  %.2f TB/s' % (ms, n * 4 / ms / 1e9))
ms = t(lambda: a.fill_(1.5)); print('fill_ (write only): %.3f ms  %.2f TB/s' % (ms, n * 4 / ms / 1e9))
ms = t(lambda: a.sum()); print('sum   (read only) : %.3f ms  %.2f TB/s' % (ms, n * 4 / ms / 1e9))
ms = t(lambda: b.copy_(a)); print('copy  (read+write): %.3f ms  %.2f TB/s total' % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: torch.add(a, b, out=b)); print('add   (2 read + 1 write): %.3f ms  %.2f TB/s total' % (ms, 3 * n * 4 / ms / 1e9))
