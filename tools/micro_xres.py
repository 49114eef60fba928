"""xResidualBlock(64,64) fwd+bwd on 16x64x256x256 -- for rocprofv3 kernel stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
dev = 'cuda'
torch.manual_seed(0)
blk = S.xresidualblock.xResidualBlock(64, 64).to(dev).train()
x = torch.randn(16, 64, 256, 256, device=dev, requires_grad=True)
for _ in range(3):
    blk.zero_grad(set_to_none=True)
    blk(x).sum().backward()
torch.cuda.synchronize()
print('done')
