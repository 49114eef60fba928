"""Per-shape profile of one G+D step: which kernel every conv / weight-gradient launch maps to, its time
and TFLOP/s (HIP events around each launch).  Usage: python tools/shapes.py [batch]"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(41)
G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
D = S.models_seg_gan.Discriminator(3,3,64,8,1024).to(dev).train()
og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
g = torch.Generator().manual_seed(7)
inp = torch.randn(B,3,512,512, generator=g).to(dev); tgt = (torch.rand(B,3,512,512, generator=g) > 0.5).float().to(dev)
args = (inp, tgt, G, D, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3)
S.train_seg_gan.gan_step(*args); torch.cuda.synchronize()
S.ops.PROFILE = []; S.ops.PROFILE_SHAPES = True
S.train_seg_gan.gan_step(*args); torch.cuda.synchronize()
agg = {}
for label, fl, e0, e1, _tag in S.ops.PROFILE:
    a = agg.setdefault(label, [0, 0, 0]); a[0] += fl; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(a[1] for a in agg.values())
print('total mfma ms %.1f' % tot)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-72s x%-3d %7.2f ms %5.1f%%  %6.1f TF' % (k, a[2], a[1], 100*a[1]/tot, a[0]/a[1]/1e9))

# algorithmic HBM bytes per launch of each kernel (input + output + weights, fp32), to set against the PMC traffic
import re
per_kernel = {}
for k, a in agg.items():
    m = re.match(r'(\S+) n(\d+) (\d+)x(\d+) cin(\d+) cout(\d+) (?:taps|k)(\d+)', k)
    if not m:
        continue
    name, n, gh, gw, cin, cout, taps = m.group(1), *map(int, m.groups()[1:])
    if name.startswith('wgrad'):
        taps = taps * taps                      # 'k3' -> 9 taps
    byts = 4.0 * (n * gh * gw * (cin + cout) + cin * cout * taps)
    p = per_kernel.setdefault(name, [0.0, 0])
    p[0] += byts * a[2]; p[1] += a[2]
print('\nalgorithmic bytes per launch (in + out + weights):')
for name, (b, c) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0]):
    print('%-40s launches %4d  avg %8.1f MB' % (name, c, b / c / 1e6))
