"""Round-2 additions (VERDICT r1 "what's weak" + ADVICE r1), all on the GPU through the C-ABI:
  * StableBCELoss has a real gradient (ADVICE: ops.py:796 passed only g[0], g[1]);
  * the inf/nan-BCE fallback of BCEDiceLoss (losses.py:297-300) forward + backward on HIP;
  * eval-mode BN-fold cache sees running statistics rewritten by a training forward without an optimizer step;
  * the RCCL ("nccl") calls of the data-parallel path executed on hardware at world size 1 (SSG_DIST_FORCE=1): a 1-GPU box
    cannot host two RCCL ranks, so this is where init_process_group(nccl, device_id), all_reduce(AVG) on the flat gradient
    buckets, the fp64 sync-BN all-reduces and the metric-sum all-reduce actually run;
  * the literal statement sequence of the reference's train() body (train_seg_gan.py:188-233: boolean index_put on the
    logits, stock nn.MSELoss / nn.BCEWithLogitsLoss modules, numpy metrics, clip_gradient + optimizer.step()) on the HIP
    modules equals the fused gan_step (INTEGRATION.md 1 claims it runs unchanged).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol, what):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), '%s: max err %.3e (ref scale %.3e), %d bad' % (what, err.max().item(), b.abs().max().item(), int(bad.sum()))


def test_stable_bce_loss_has_gradient(pkg, dev):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 12, 20, generator=g) * 3
    t = (torch.rand(2, 3, 12, 20, generator=g) > 0.5).float()
    xr = x.clone().requires_grad_(True)
    lr = (xr.clamp(min=0) - xr * t + torch.log1p(torch.exp(-xr.abs()))).mean()          # losses.py:130-136
    lr.backward()
    xd = x.to(dev).requires_grad_(True)
    ld = pkg.losses.StableBCELoss()(xd, t.to(dev))
    ld.backward()
    assert abs(ld.item() - lr.item()) < 1e-6
    assert xd.grad.abs().max().item() > 0
    _close(xd.grad, xr.grad, 1e-5, 1e-9, 'StableBCELoss grad')
    # mixed use: all three differentiable components at once
    xd2 = x.to(dev).requires_grad_(True)
    res = pkg.ops.seg_loss(xd2, t.to(dev), 1)
    (0.7 * res[2] + 0.2 * res[1]).backward()
    xr2 = x.clone().requires_grad_(True)
    (0.7 * (xr2.clamp(min=0) - xr2 * t + torch.log1p(torch.exp(-xr2.abs()))).mean() + 0.2 * F.mse_loss(xr2, t)).backward()
    _close(xd2.grad, xr2.grad, 1e-5, 1e-9, 'bce+mse grad')
    assert pkg.losses.__all__ == ['BCEDiceLoss', 'LovaszHingeLoss']
    with pytest.raises(NotImplementedError):
        pkg.losses.LovaszHingeLoss()(xd, t.to(dev))


def test_bce_dice_inf_fallback_on_hip(pkg, dev):
    """losses.py:297-300: a non-finite BCE term switches the loss to 2*dice; gradients then flow through dice only."""
    from oracle import seg_gan_cpu as O
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 3, 8, 8, generator=g)
    t = (torch.rand(2, 3, 8, 8, generator=g) > 0.5).float()
    x[0, 1, 2, 3] = float('inf'); t[0, 1, 2, 3] = 1.0              # inf - inf*1 -> nan in the BCE sum
    xr = x.clone().requires_grad_(True)
    lr = O.bce_dice_loss(xr, t)
    lr.backward()
    xd = x.to(dev).requires_grad_(True)
    res = pkg.ops.seg_loss(xd, t.to(dev), 1)
    res[0].backward()
    assert res[6].item() == 0.0, 'finite flag should be off'
    assert np.isfinite(res[0].item()) and abs(res[0].item() - lr.item()) < 1e-5, (res[0].item(), lr.item())
    gr = xr.grad.clone(); gr[~torch.isfinite(gr)] = 0
    gd = xd.grad.detach().cpu().clone(); gd[~torch.isfinite(gd)] = 0
    _close(gd, gr, 1e-4, 1e-8, 'fallback grad')


def test_fold_cache_sees_running_stat_updates(pkg, dev):
    """ADVICE r1 (archs.py:59): train-mode forward under no_grad (no optimizer step) then eval() must not reuse stale folds."""
    torch.manual_seed(3)
    blk = pkg.archs.BasicBlock(8, 8).to(dev)
    x = torch.randn(2, 8, 16, 16, device=dev)
    blk.eval()
    with torch.no_grad():
        y0 = blk(x).clone()                      # fills the fold cache
    blk.train()
    with torch.no_grad():
        blk(x * 3 + 1)                           # rewrites running stats through raw pointers, no optimizer step
    blk.eval()
    with torch.no_grad():
        y1 = blk(x).clone()
        blk._fold_cache = None
        y2 = blk(x).clone()
    assert not torch.equal(y0, y1), 'running statistics changed, eval output must change'
    assert torch.equal(y1, y2), 'stale fold cache'


_RCCL_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.nn as nn, torch.distributed as dist
import ssunet_gan_amd as S
rank, world, local = S.dp.init_from_env()
assert S.dp.is_dist() and dist.get_backend() == 'nccl' and world == 1
dev = torch.device('cuda', 0)
torch.manual_seed(41)
G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
S.dp.broadcast_parameters(G); S.dp.broadcast_parameters(D)
S.dp.convert_sync_batchnorm(G); S.dp.convert_sync_batchnorm(D)
og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
g = torch.Generator().manual_seed(7)
inp = torch.randn(2, 3, 64, 64, generator=g).to(dev); tgt = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float().to(dev)
sg, sd = S.dp.grad_syncs(G, D)
assert sg is not None and sg.reduce and sg._avg_native
out = None
for _ in range(2):
    out = S.train_seg_gan.gan_step(inp, tgt, G, D, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3, sg, sd)
torch.cuda.synchronize()
res = dict(vals=[float(v) for v in out],
           g=[float(p.detach().double().abs().sum()) for p in G.parameters()],
           d=[float(p.detach().double().abs().sum()) for p in D.parameters()],
           rm=float(D.conv_blocks[1].conv_block[1].running_var.double().sum()))
dist.barrier(); dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
'''


@pytest.mark.timeout(900)
def test_rccl_path_runs_at_world_size_one(pkg, dev):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(SSG_DIST_FORCE='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29631',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_WORKER % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    # the same two steps in this process without a process group, sync-BN formula (clamp(var, eps)^-1/2, batchnorm.py:127)
    S = pkg
    torch.manual_seed(41)
    G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
    D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
    for m in list(G.modules()) + list(D.modules()):
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m._ssg_var_mode = 1
    og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(2, 3, 64, 64, generator=g).to(dev); tgt = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float().to(dev)
    for _ in range(2):
        out = S.train_seg_gan.gan_step(inp, tgt, G, D, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3)
    # a one-rank all-reduce is the identity and every kernel is deterministic: the two runs agree bit for bit
    # (IoU / Dice come back as fp64 ratios of the all-reduced sums in the distributed run, as fp32 scalars otherwise)
    assert [float(v) for v in out] == pytest.approx(got['vals'], rel=0, abs=2e-7)
    assert [float(p.detach().double().abs().sum()) for p in G.parameters()] == got['g']
    assert [float(p.detach().double().abs().sum()) for p in D.parameters()] == got['d']
    assert float(D.conv_blocks[1].conv_block[1].running_var.double().sum()) == got['rm']


def test_literal_reference_train_body_on_hip_modules(pkg, dev):
    """train_seg_gan.py:188-233 statement by statement (stock criteria, boolean index_put, numpy metrics, clip_gradient +
    optimizer.step()) on the HIP Generator / Discriminator == the fused gan_step, same seeds."""
    S = pkg

    def build():
        torch.manual_seed(41)
        G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
        D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
        return G, D, torch.optim.Adam(G.parameters(), lr=2e-5), torch.optim.Adam(D.parameters(), lr=2e-5)

    g = torch.Generator().manual_seed(7)
    input = torch.randn(2, 3, 64, 64, generator=g).to(dev); target = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float().to(dev)
    num_class = 3
    alpa, beta, grad_clip = 1e-4, 1e-3, 0.8
    generator, discriminator, optimizer_g, optimizer_d = build()
    criterion = S.losses.BCEDiceLoss()
    content_loss_criterion = nn.MSELoss().to(dev)
    adversarial_loss_criterion = nn.BCEWithLogitsLoss().to(dev)
    # ---- the reference's statements ----
    generator_output = generator(input)
    generator_output[torch.isnan(generator_output)] = 0
    if num_class > 1:
        fg_outputs = generator_output[:, 1:num_class, :, :]
        fg_targets = target[:, 1:num_class, :, :]
    loss = criterion(generator_output, target)
    content_loss = content_loss_criterion(generator_output, target)
    iou = S.metrics.iou_score(fg_outputs, fg_targets)
    dice = S.metrics.dice_coef(fg_outputs, fg_targets)
    seg_discriminated = discriminator(generator_output)
    adversarial_loss = adversarial_loss_criterion(seg_discriminated, torch.ones_like(seg_discriminated))
    perceptual_loss = loss + alpa * content_loss + beta * adversarial_loss
    optimizer_g.zero_grad()
    perceptual_loss.backward()
    if grad_clip is not None:
        S.srgan_utils.clip_gradient(optimizer_g, grad_clip)
    optimizer_g.step()                   # stock in-place update: tensor version counters move, packed-weight caches follow
    hr_discriminated = discriminator(target)
    sr_discriminated = discriminator(generator_output.detach())
    adversarial_loss = adversarial_loss_criterion(sr_discriminated, torch.zeros_like(sr_discriminated)) + \
        adversarial_loss_criterion(hr_discriminated, torch.ones_like(hr_discriminated))
    optimizer_d.zero_grad()
    adversarial_loss.backward()
    if grad_clip is not None:
        S.srgan_utils.clip_gradient(optimizer_d, grad_clip)
    optimizer_d.step()
    lit = dict(loss=float(loss), iou=float(iou), dice=float(dice), adv=float(adversarial_loss),
               g=[p.detach().clone() for p in generator.parameters()], d=[p.detach().clone() for p in discriminator.parameters()])
    # ---- the fused step ----
    G2, D2, og2, od2 = build()
    l2, i2, d2, c2, ag2, ad2 = S.train_seg_gan.gan_step(input, target, G2, D2, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(),
                                                         og2, od2, 3)
    assert abs(float(l2) - lit['loss']) < 1e-6 and abs(float(i2) - lit['iou']) < 1e-6 and abs(float(d2) - lit['dice']) < 1e-6
    assert abs(float(ad2) - lit['adv']) < 1e-6
    # same kernels underneath; the literal path adds torch's own elementwise ops (index_put, loss modules, Adam), so the
    # parameters agree to fp32 rounding of one Adam update (lr 2e-5), not bit for bit
    for a, b in zip(lit['g'], G2.parameters()):
        assert (a - b.detach()).abs().max().item() <= 4.1e-5
    for a, b in zip(lit['d'], D2.parameters()):
        assert (a - b.detach()).abs().max().item() <= 4.1e-5
    frac_g = np.mean([((a - b.detach()).abs() > 1e-7).float().mean().item() for a, b in zip(lit['g'], G2.parameters())])
    assert frac_g < 0.02, 'more than 2%% of G weights stepped differently: %.4f' % frac_g
