"""Diagnostic: relative RMS difference between the bf16 and the fp32 run of EfficientNet-B4 after every MBConv block (same
weights, same input).  Rounding noise grows smoothly (random walk); a kernel bug shows as a jump.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S  # noqa: E402

dev = torch.device('cuda', 0)
name, n, hw = (sys.argv[1] if len(sys.argv) > 1 else 'efficientnet-b4'), int(sys.argv[2]) if len(sys.argv) > 2 else 2, int(sys.argv[3]) if len(sys.argv) > 3 else 512
outs = {}
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(37)
    net = S.efficientnet_pytorch.EfficientNet.from_name(name, override_params=dict(drop_connect_rate=0.0)).to(dev).train().set_compute_dtype(dtype)
    acc = []
    for b in net._blocks:
        b.register_forward_hook(lambda m, i, o, acc=acc: acc.append(o.detach().float().cpu()))
    x = torch.randn(n, 3, hw, hw, generator=torch.Generator().manual_seed(23)).to(dev)
    with torch.no_grad():
        f = net.extract_features(x)
    acc.append(f.detach().float().cpu())
    outs[dtype] = acc
    del net
for i, (a, b) in enumerate(zip(outs[torch.float32], outs[torch.bfloat16])):
    d = (a - b).double()
    print('block %2d  shape %-22s rel rms err %.4f   max err / max %.4f' % (i, tuple(a.shape), (d.pow(2).mean().sqrt() / a.double().pow(2).mean().sqrt()).item(),
                                                                         (d.abs().max() / a.abs().max()).item()))
