"""Which stock torch kernels still run inside one G+D step, and from which source line (torch.profiler, with_stack)."""
import os, sys, collections
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
dev = torch.device('cuda')
torch.manual_seed(41)
G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
g = torch.Generator().manual_seed(7)
B = 4
inp = torch.randn(B, 3, 256, 256, generator=g).to(dev); tgt = (torch.rand(B, 3, 256, 256, generator=g) > 0.5).float().to(dev)
args = (inp, tgt, G, D, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3)
for _ in range(2):
    S.train_seg_gan.gan_step(*args)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    S.train_seg_gan.gan_step(*args)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
        continue
    where = '?'
    for fr in (ev.stack or []):
        if 'ssunet' in fr and 'site-packages' not in fr:
            where = fr.strip(); break
    cnt[(ev.name, where)] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print('%4d  %-28s %s' % (n, name, where[-110:]))
