import sys, torch, importlib
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module('ssunet_gan_amd')
from ssunet_gan_amd import ops
dev = 'cuda'
torch.manual_seed(0)
for (ci, co, hw) in [(64, 3, 512), (3, 64, 512), (128, 3, 256)]:
    x = ops.to_nhwc(torch.randn(16, ci, hw, hw, device=dev))
    dy = ops.to_nhwc(torch.randn(16, co, hw, hw, device=dev))
    for _ in range(3):
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dw = ops._conv_wgrad_impl(x, None, dy, (co, ci, 3, 3), 1, 1)
    e1.record(); torch.cuda.synchronize()
    print(ci, co, hw, 'ms', e0.elapsed_time(e1) / 5, flush=True)
