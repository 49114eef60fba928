"""CPU oracle for the wired ssUnet-GAN hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Plain torch fp32 ops on the CPU.  Every item cites the reference file:line it restates
(paths relative to /root/reference/scripts).  Parameter creation order and initialisers
are kept identical to the reference so that `torch.manual_seed(s)` regenerates the same
weights (the fixtures carry seeds, not 231 MB of weights).

Pinned by tests/golden/*.npz, generated from the imported reference by oracle/gen_golden.py.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

NB_FILTER = (64, 128, 256, 384, 512, 768)          # archs.py:568
SS_SCALE = 16                                      # archs.py:575


# ----------------------------------------------------------------------------- activation pattern (test aid)
class ActivationPattern(object):
    """The network is piecewise linear in its ReLU / LeakyReLU masks and max-pool argmax choices.  Parity tests use this to
    separate the two things that can differ between two fp32 implementations: (a) WHICH linear piece a near-tie lands on
    (|pre-activation| or top-2 pool gap below fp32 noise) and (b) the arithmetic on a given piece.
      mode 'record': the forward runs normally and appends, in call order, ('act', pre-activation) / ('pool', input, plane index);
      mode 'impose': items is a list of bool masks (act) / int64 window positions 0..3 (pool) in the same call order, and the
                     forward uses THEM instead of its own comparisons (so an fp64 run evaluates exactly the piece another
                     implementation was on).
    With PATTERN = None (the default) every op is the stock torch op: fixtures and the CPU baseline are unaffected."""

    def __init__(self, mode, items=None, keep=False):
        assert mode in ('record', 'impose')
        self.mode, self.items, self.pos = mode, (items if items is not None else []), 0
        # impose + keep: `seen` receives, in call order, the pre-activation (act) / pool input this run computed at each decision
        # while it FOLLOWED the imposed pattern up to there -- so a decision can be checked given identical upstream decisions
        # keep may also be a callable(k, x, item): the value is handed over as it is computed and nothing is retained (the
        # 16 x 512^2 forward has 2.2e9 decisions: 9 GB of pre-activations if they were all kept)
        self.keep, self.seen = keep, []

    def note(self, x):
        k = self.pos - 1
        if callable(self.keep):
            self.keep(k, x.detach(), self.items[k])
            self.seen.append(None)
        else:
            self.seen.append(x.detach().clone())


PATTERN = None


def _act(x, slope=0.0):
    P = PATTERN
    if P is None or P.mode == 'record':
        if P is not None:
            P.items.append(('act', x.detach().clone()))
        return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)
    m = P.items[P.pos]; P.pos += 1
    assert m.shape == x.shape, 'activation pattern out of step: %s vs %s' % (tuple(m.shape), tuple(x.shape))
    if P.keep:
        P.note(x)
    return torch.where(m, x, x * slope)


class _ReLU(nn.Module):
    def forward(self, x):
        return _act(x)


class _LeakyReLU(nn.Module):
    def __init__(self, slope):
        super().__init__()
        self.negative_slope = slope

    def forward(self, x):
        return _act(x, self.negative_slope)


class _MaxPoolIdx(nn.Module):
    """nn.MaxPool2d(2, 2, return_indices=True) (archs.py:571)."""

    def forward(self, x):
        P = PATTERN
        if P is None or P.mode == 'record':
            y, idx = F.max_pool2d(x, 2, 2, return_indices=True)
            if P is not None:
                P.items.append(('pool', x.detach().clone(), idx.clone()))
            return y, idx
        win = P.items[P.pos]; P.pos += 1
        if P.keep:
            P.note(x)
        n, c, h, w = x.shape
        assert win.shape == (n, c, h // 2, w // 2)
        xs = x.unfold(2, 2, 2).unfold(3, 2, 2).reshape(n, c, h // 2, w // 2, 4)
        y = xs.gather(-1, win.unsqueeze(-1)).squeeze(-1)
        oy = torch.arange(h // 2).view(1, 1, -1, 1); ox = torch.arange(w // 2).view(1, 1, 1, -1)
        idx = (2 * oy + win // 2) * w + 2 * ox + win % 2
        return y, idx


# ----------------------------------------------------------------------------- blocks
class ResBlockCPU(nn.Module):
    """archs.py:205-241 (BasicBlock): relu(bn1(conv3x3)) -> bn2(conv3x3) -> += 1x1 shortcut
    (no BN on the shortcut, :217-219) -> relu.  All convs bias-free."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, 1, bias=False))

    def forward(self, x):
        y = _act(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return _act(y + self.shortcut(x))


class SelfSpadeCPU(nn.Module):
    """normalization.py:67-122 (SPADE) as wired: self-conditioned modulation,
    out = x*(1+gamma(a)) + beta(a), a = relu(shared(x2map(x))).  The param-free norm
    is constructed (buffers appear in the state_dict) but never applied (:110)."""

    def __init__(self, norm_nc, label_nc, nhidden):
        super().__init__()
        self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)   # :81 ('batch')
        nh = int(max(nhidden, 4))                                      # :88
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, nh, 3, padding=1), _ReLU())
        self.x2map = nn.Conv2d(norm_nc, label_nc, 3, padding=1)
        self.mlp_gamma = nn.Conv2d(nh, norm_nc, 3, padding=1)
        self.mlp_beta = nn.Conv2d(nh, norm_nc, 3, padding=1)

    def forward(self, x, segmap=None):
        seg = self.x2map(x if segmap is None else segmap)
        a = self.mlp_shared(seg)
        return x * (1 + self.mlp_gamma(a)) + self.mlp_beta(a)


class UNetRSSv2CPU(nn.Module):
    """archs.py:559-671 (UNet_R_SS_v2).  Module creation order == reference order."""

    def __init__(self, num_classes, input_channels=3, deep_supervision=False, **kw):
        super().__init__()
        f = NB_FILTER
        sm = num_classes
        self.pool = _MaxPoolIdx()
        self.unpool = nn.MaxUnpool2d(2, stride=2)
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)

        def stage(name, cin, cout, head=None):
            setattr(self, 'conv' + name, ResBlockCPU(cin, cout))
            setattr(self, 'SPADE' + name, SelfSpadeCPU(cout, sm, cout / SS_SCALE))
            if head is not None:
                setattr(self, 'conv_head' + name, nn.Conv2d(cout, head, 1, 1, bias=False))

        stage('0_0', input_channels, f[0])
        stage('1_0', f[0], f[1])
        stage('2_0', f[1], f[2])
        stage('3_0', f[2], f[3])
        stage('4_0', f[3], f[4])
        stage('5_0', f[4], f[5], head=f[4])
        stage('4_1', f[4] + f[4], f[4], head=f[3])
        stage('3_1', f[3] + f[3], f[3], head=f[2])
        stage('2_1', f[2] + f[2], f[2])
        stage('1_1', f[1] + f[2], f[1])
        stage('0_1', f[0] + f[1], f[0])
        self.final = nn.Conv2d(f[0], num_classes, 1)
        nn.init.kaiming_uniform_(self.final.weight, mode='fan_in')     # :619-621
        self.final.bias.data.fill_(0)

    def forward(self, x):                                              # :623-671
        e0 = self.SPADE0_0(self.conv0_0(x))
        p0, _ = self.pool(e0)
        e1 = self.SPADE1_0(self.conv1_0(p0))
        p1, _ = self.pool(e1)
        e2 = self.SPADE2_0(self.conv2_0(p1))
        p2, i2 = self.pool(e2)
        e3 = self.SPADE3_0(self.conv3_0(p2))
        p3, i3 = self.pool(e3)
        e4 = self.SPADE4_0(self.conv4_0(p3))
        p4, i4 = self.pool(e4)
        e5 = self.conv_head5_0(self.SPADE5_0(self.conv5_0(p4)))
        d4 = self.SPADE4_1(self.conv4_1(torch.cat([e4, self.unpool(e5, i4)], 1)))
        d4 = self.conv_head4_1(d4)
        d3 = self.SPADE3_1(self.conv3_1(torch.cat([e3, self.unpool(d4, i3)], 1)))
        d3 = self.conv_head3_1(d3)
        d2 = self.SPADE2_1(self.conv2_1(torch.cat([e2, self.unpool(d3, i2)], 1)))
        d1 = self.SPADE1_1(self.conv1_1(torch.cat([e1, self.up(d2)], 1)))
        d0 = self.SPADE0_1(self.conv0_1(torch.cat([e0, self.up(d1)], 1)))
        return self.final(d0)


ARCHS = {'UNet_R_SS_v2': UNetRSSv2CPU}


class GeneratorCPU(nn.Module):
    """models_seg_gan.py:193-243: thin wrapper, self.net = archs[config['arch']](...)."""

    def __init__(self, config):
        super().__init__()
        self.net = ARCHS[config['arch']](config['num_classes'], config['input_channels'],
                                         config['deep_supervision'])

    def forward(self, x):
        return self.net(x)


class _ConvBlockCPU(nn.Module):
    """models_seg_gan.py:13-64 (ConvolutionalBlock), LeakyReLU(0.2) flavour only."""

    def __init__(self, cin, cout, k, stride, bn):
        super().__init__()
        layers = [nn.Conv2d(cin, cout, k, stride, k // 2)]
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(_LeakyReLU(0.2))
        self.conv_block = nn.Sequential(*layers)

    def forward(self, x):
        return self.conv_block(x)


class DiscriminatorCPU(nn.Module):
    """models_seg_gan.py:246-300: 8 conv blocks (block 0 without BN, odd blocks stride 2,
    channels 64,64,128,128,256,256,512,512) -> AdaptiveAvgPool(6,6) -> fc1 -> LReLU -> fc2."""

    def __init__(self, num_classes, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024):
        super().__init__()
        cin, blocks = num_classes, []
        for i in range(n_blocks):
            cout = (n_channels if i == 0 else cin * 2) if i % 2 == 0 else cin
            blocks.append(_ConvBlockCPU(cin, cout, kernel_size, 1 if i % 2 == 0 else 2, i != 0))
            cin = cout
        self.conv_blocks = nn.Sequential(*blocks)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((6, 6))
        self.fc1 = nn.Linear(cin * 36, fc_size)
        self.leaky_relu = _LeakyReLU(0.2)
        self.fc2 = nn.Linear(1024, 1)

    def forward(self, x):
        y = self.adaptive_pool(self.conv_blocks(x))
        return self.fc2(self.leaky_relu(self.fc1(y.view(x.size(0), -1))))


# ----------------------------------------------------------------------------- losses / metrics
def stable_bce(x, t):
    """losses.py:130-136 (StableBCELoss)."""
    return (x.clamp(min=0) - x * t + (1 + (-x.abs()).exp()).log()).mean()


def bce_dice_loss(x, t):
    """losses.py:274-302 (BCEDiceLoss): 0.5*bce + 1 - mean_n dice_n, with the inf/nan
    fallback 2*dice (:297-300)."""
    bce = stable_bce(x, t)
    n = t.size(0)
    p = torch.sigmoid(x).view(n, -1)
    tt = t.view(n, -1)
    dice_n = (2.0 * (p * tt).sum(1) + 1e-5) / (p.sum(1) + tt.sum(1) + 1e-5)
    dice = 1 - dice_n.sum() / n
    if torch.isinf(bce) or torch.isnan(bce):
        return 2.0 * dice
    return 0.5 * bce + dice


def iou_score(out, tgt):
    """metrics.py:6-22: hard IoU at 0.5 on sigmoid over the whole tensor (numpy)."""
    o = torch.sigmoid(out).data.cpu().numpy()
    t = tgt.data.cpu().numpy()
    o_ = o > 0.5
    o_[np.isnan(o)] = False
    t_ = t > 0.5
    return ((o_ & t_).sum() + 1e-5) / ((o_ | t_).sum() + 1e-5)


def dice_coef(out, tgt):
    """metrics.py:25-35: soft Dice over the whole flattened tensor (numpy)."""
    o = torch.sigmoid(out).view(-1).data.cpu().numpy()
    t = tgt.view(-1).data.cpu().numpy()
    return (2.0 * (o * t).sum() + 1e-5) / (o.sum() + t.sum() + 1e-5)


def clip_gradient(optimizer, c):
    """srgan_utils.py:186-195: elementwise clamp of every grad to [-c, c]."""
    for group in optimizer.param_groups:
        for p in group['params']:
            if p.grad is not None:
                p.grad.data.clamp_(-c, c)


# ----------------------------------------------------------------------------- the step
ALPHA, BETA, GRAD_CLIP = 1e-4, 1e-3, 0.8           # train_seg_gan.py:172-174


def gan_step(G, D, opt_g, opt_d, inp, tgt, num_classes=3, record=None):
    """One iteration of train_seg_gan.py:182-233 on CPU.  Returns dict(loss, closs, adv_g,
    adv_d, iou, dice, out).  `record(tag)` (optional) is called after each backward and
    each optimizer step so the fixture generator can snapshot grads/params."""
    bce_logits = nn.BCEWithLogitsLoss()
    mse = nn.MSELoss()
    out = G(inp)                                                       # :188
    out[torch.isnan(out)] = 0                                          # :190
    out_m = out[:, 1:num_classes].clone()
    tar_m = tgt[:, 1:num_classes].clone()
    loss = bce_dice_loss(out, tgt)                                     # :194
    closs = mse(out, tgt)                                              # :195
    iou = iou_score(out_m, tar_m)
    dice = dice_coef(out_m, tar_m)
    sd = D(out)                                                        # :202
    adv_g = bce_logits(sd, torch.ones_like(sd))
    total = loss + ALPHA * closs + BETA * adv_g
    opt_g.zero_grad()
    total.backward()
    if record:
        record('g_bwd')
    clip_gradient(opt_g, GRAD_CLIP)
    opt_g.step()
    if record:
        record('g_step')
    hr = D(tgt)                                                        # :217
    sr = D(out.detach())                                               # :218
    adv_d = bce_logits(sr, torch.zeros_like(sr)) + bce_logits(hr, torch.ones_like(hr))
    opt_d.zero_grad()                                                  # :225
    adv_d.backward()
    if record:
        record('d_bwd')
    clip_gradient(opt_d, GRAD_CLIP)
    opt_d.step()
    if record:
        record('d_step')
    return OrderedDict(loss=loss.item(), closs=closs.item(), adv_g=adv_g.item(),
                       adv_d=adv_d.item(), iou=float(iou), dice=float(dice), out=out.detach())


def make_models(seed=41, num_classes=3):
    """Model init exactly as train_seg_gan.py:35-36,448,463-468 (manual_seed(41), G then D,
    Adam lr=gan_lr=2e-5 for both; configs/config_v1.json:33)."""
    torch.manual_seed(seed)
    G = GeneratorCPU(dict(arch='UNet_R_SS_v2', num_classes=num_classes, input_channels=3,
                          deep_supervision=False))
    D = DiscriminatorCPU(num_classes, 3, 64, 8, 1024)
    opt_g = torch.optim.Adam(filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    opt_d = torch.optim.Adam(filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    return G, D, opt_g, opt_d


class SyncBatchNorm2dCPU(nn.BatchNorm2d):
    """batchnorm.py:40-127 (`_SynchronizedBatchNorm`) for W replicas evaluated in ONE process on the concatenated batch: the
    parallel-training branch (:57-80) with the whole batch's sums handed to `_compute_mean_std` (:115-127) --
    mean = sum/n, inv_std = clamp((ssum - sum*mean)/n, eps)^-1/2, running_var <- unbiased (n-1) estimate.  Eval mode is the
    stock batch norm (:52-55).  Pinned by tests/golden/step_dp_w2_n4_64.npz (oracle/gen_golden.py --only dp)."""

    def forward(self, x):
        if not self.training:
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, self.momentum, self.eps)
        shp = x.shape
        x = x.reshape(shp[0], self.num_features, -1)                            # :58-59
        size = x.size(0) * x.size(2)                                            # :62
        sum_ = x.sum(dim=0).sum(dim=-1); ssum = (x ** 2).sum(dim=0).sum(dim=-1)  # :63-64 (_sum_ft)
        assert size > 1                                                         # :118
        mean = sum_ / size; sumvar = ssum - sum_ * mean                         # :119-120
        unbias_var = sumvar / (size - 1); bias_var = sumvar / size              # :121-122
        self.running_mean = (1 - self.momentum) * self.running_mean + self.momentum * mean.data     # :124
        self.running_var = (1 - self.momentum) * self.running_var + self.momentum * unbias_var.data  # :125
        inv_std = bias_var.clamp(self.eps) ** -0.5                              # :127
        y = (x - mean.view(1, -1, 1)) * (inv_std * self.weight).view(1, -1, 1) + self.bias.view(1, -1, 1)   # :75
        return y.view(shp)


def convert_sync_batchnorm(module):
    """batchnorm.py:320-360 (`convert_model`): every BatchNorm2d -> the synchronised class, parameters and statistics carried over."""
    mod = module
    if isinstance(module, nn.BatchNorm2d) and not isinstance(module, SyncBatchNorm2dCPU):
        mod = SyncBatchNorm2dCPU(module.num_features, module.eps, module.momentum, module.affine)
        mod.running_mean = module.running_mean; mod.running_var = module.running_var
        if module.affine:
            mod.weight.data = module.weight.data.clone().detach(); mod.bias.data = module.bias.data.clone().detach()
    for name, child in module.named_children():
        mod.add_module(name, convert_sync_batchnorm(child))
    return mod


def synthetic_batch(n, h, w, seed=7, num_classes=3):
    """SURVEY.md 8(d) synthetic tiles: randn input, Bernoulli(0.5) masks."""
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn(n, 3, h, w, generator=g)
    tgt = (torch.rand(n, num_classes, h, w, generator=g) > 0.5).float()
    return inp, tgt


def stage1_step(model, optimizer, inp, tgt, clip=0.7, num_classes=3):
    """One iteration of train.py:79-115 (stage 1, deep_supervision off): forward, BCEDice, IoU/Dice on
    channels 1:, THEN the weight clamp to +-clip (:111-112), zero_grad, backward, optimizer.step()."""
    out = model(inp)
    out[torch.isnan(out)] = 0
    loss = bce_dice_loss(out, tgt)
    iou = iou_score(out[:, 1:num_classes].clone(), tgt[:, 1:num_classes].clone())
    dice = dice_coef(out[:, 1:num_classes].clone(), tgt[:, 1:num_classes].clone())
    for p in model.parameters():
        p.data.clamp_(-clip, clip)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return OrderedDict(loss=loss.item(), iou=float(iou), dice=float(dice), out=out.detach())
