"""EfficientNet-B4 extract_features fwd+bwd -- for rocprofv3 kernel stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
dev = 'cuda'
name, n, hw = (sys.argv[1] if len(sys.argv) > 1 else 'efficientnet-b4'), 4, 1024
torch.manual_seed(0)
enc = S.efficientnet_pytorch.EfficientNet.from_name(name).to(dev).train()
x = torch.randn(n, 3, hw, hw, device=dev)
for _ in range(3):
    enc.zero_grad(set_to_none=True)
    enc.extract_features(x).sum().backward()
torch.cuda.synchronize()
print('done')
