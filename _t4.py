import sys, os, torch, importlib
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('ssunet_gan_amd')
from ssunet_gan_amd import ops
from ssunet_gan_amd._lib import ACT_NONE
dev = 'cuda'
torch.manual_seed(0)
for (ci, co, hw) in [(3, 64, 512), (64, 3, 512)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
    w = torch.randn(co, ci, 3, 3, device=dev)
    for _ in range(3):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops._conv_fwd_impl(x, None, w, None, 1, 1, ACT_NONE, 0.0)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get('SSG_T4_DEBUG', '0'), ci, co, hw, 'ms', e0.elapsed_time(e1) / 10, flush=True)
