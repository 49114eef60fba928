#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own modules (TEST INFRASTRUCTURE).

Runs only in the build container, where /root/reference exists.  It imports the reference's
`archs`, `models_seg_gan`, `losses`, `metrics`, `normalization` as-is (with an empty
placeholder for the absent `torchvision`, which those files import but never touch on this
path -- SURVEY.md 8c) and drives them through the exact train_seg_gan.py:182-233 sequence.
The reference's driver files (train_seg_gan.py, srgan_utils.py) need cv2/albumentations/
tensorboardX and cannot be imported, so the *sequence* is the restated one, the *modules*
are the reference's.

Fixtures carry seeds + inputs + outputs (+ per-parameter digests), never weights or code.

    python oracle/gen_golden.py [--only step64|step256|blocks|dp|...]
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/scripts'
OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit('reference not present; fixtures can only be generated in the build container')
    warnings.filterwarnings('ignore')
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tv.models = tvm
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)
    sys.path.insert(0, REF)
    import archs, models_seg_gan, losses, metrics, normalization  # noqa
    return archs, models_seg_gan, losses, metrics, normalization


def digest(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def param_digests(module, grads=False):
    rows = []
    for _, p in module.named_parameters():
        src = p.grad if grads else p
        rows.append(digest(src) if src is not None else np.zeros(3))
    return np.stack(rows)


def buffer_digests(module):
    return np.stack([digest(b.float()) for _, b in module.named_buffers()])


def synthetic_batch(n, h, w, seed=7, num_classes=3):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn(n, 3, h, w, generator=g)
    tgt = (torch.rand(n, num_classes, h, w, generator=g) > 0.5).float()
    return inp, tgt


def ref_models(mods, seed=41):
    archs, msg, losses, metrics, _ = mods
    torch.manual_seed(seed)
    G = msg.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False))
    D = msg.Discriminator(3, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024)
    og = torch.optim.Adam(params=filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    od = torch.optim.Adam(params=filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    return G, D, og, od


def ref_step(mods, G, D, og, od, inp, tgt, rec):
    """train_seg_gan.py:182-233 with the reference's modules; clip_gradient restated
    (srgan_utils.py:186-195 is not importable: needs torchvision/PIL module constants)."""
    _, _, losses, metrics, _ = mods
    crit = losses.BCEDiceLoss()
    adv_c = nn.BCEWithLogitsLoss()
    con_c = nn.MSELoss()
    G.train(); D.train()
    out = G(inp)
    out[torch.isnan(out)] = 0
    out_m = out[:, 1:3].clone(); tar_m = tgt[:, 1:3].clone()
    loss = crit(out, tgt)
    closs = con_c(out, tgt)
    iou = metrics.iou_score(out_m, tar_m)
    dice = metrics.dice_coef(out_m, tar_m)
    sd = D(out)
    adv_g = adv_c(sd, torch.ones_like(sd))
    tot = loss + 1e-4 * closs + 1e-3 * adv_g
    og.zero_grad(); tot.backward()
    rec['g_bwd_G'] = param_digests(G, True); rec['g_bwd_D'] = param_digests(D, True)
    for grp in og.param_groups:
        for p in grp['params']:
            if p.grad is not None:
                p.grad.data.clamp_(-0.8, 0.8)
    og.step()
    rec['g_step_G'] = param_digests(G)
    hr = D(tgt); sr = D(out.detach())
    adv_d = adv_c(sr, torch.zeros_like(sr)) + adv_c(hr, torch.ones_like(hr))
    od.zero_grad(); adv_d.backward()
    rec['d_bwd_D'] = param_digests(D, True)
    for grp in od.param_groups:
        for p in grp['params']:
            if p.grad is not None:
                p.grad.data.clamp_(-0.8, 0.8)
    od.step()
    rec['d_step_D'] = param_digests(D)
    rec['bufs_G'] = buffer_digests(G); rec['bufs_D'] = buffer_digests(D)
    rec['scalars'] = np.array([loss.item(), closs.item(), adv_g.item(), adv_d.item(), float(iou), float(dice)])
    rec['logits'] = out.detach().numpy().copy()
    rec['min_abs_logit'] = np.array(out.detach().abs().min().item())
    rec['sd'] = sd.detach().numpy().copy(); rec['hr'] = hr.detach().numpy().copy(); rec['sr'] = sr.detach().numpy().copy()


def gen_step(mods, name, n, h, w, steps, keep_logits=True):
    G, D, og, od = ref_models(mods)
    inp, tgt = synthetic_batch(n, h, w)
    data = dict(seed_model=np.array(41), seed_batch=np.array(7), shape=np.array([n, 3, h, w]),
                init_G=param_digests(G), init_D=param_digests(D),
                param_names_G=np.array([k for k, _ in G.named_parameters()]),
                param_names_D=np.array([k for k, _ in D.named_parameters()]),
                state_keys_G=np.array(list(G.state_dict().keys())),
                state_keys_D=np.array(list(D.state_dict().keys())))
    if keep_logits:
        data['input'] = inp.numpy(); data['target'] = tgt.numpy()
    for s in range(steps):
        rec = {}
        ref_step(mods, G, D, og, od, inp, tgt, rec)
        if not keep_logits:
            lg = rec.pop('logits')
            rec['logits_ds'] = lg[:, :, ::8, ::8].copy()        # down-sampled view + digest
            rec['logits_digest'] = digest(torch.from_numpy(lg))
        for k, v in rec.items():
            data['s%d_%s' % (s, k)] = v
        print(name, 'step', s, rec['scalars'])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)


def gen_step_sn(mods):
    """BASELINE config 5, "spectral-norm discriminator on": the reference's own scripts/spectral_norm.py (vendored torch
    spectral_norm; imported by nobody in the reference -- SURVEY.md 0) applied to the eight discriminator convs, then one
    G+D step of train_seg_gan.py:182-233 on 2 x 3 x 64 x 64 (two steps: u/v carry over)."""
    import spectral_norm as ref_sn
    G, D, _, _ = ref_models(mods)
    convs = [m for m in D.modules() if isinstance(m, nn.Conv2d)]
    assert len(convs) == 8
    for m in convs:                                   # consumes the RNG for u, v right after the model init (seed 41)
        ref_sn.spectral_norm(m)
    og = torch.optim.Adam(params=filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    od = torch.optim.Adam(params=filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    inp, tgt = synthetic_batch(2, 64, 64)
    data = dict(seed_model=np.array(41), seed_batch=np.array(7), shape=np.array([2, 3, 64, 64]),
                state_keys_D=np.array(list(D.state_dict().keys())), param_names_D=np.array([k for k, _ in D.named_parameters()]),
                init_D=param_digests(D), init_uv=buffer_digests(D))
    for s in range(2):
        rec = {}
        ref_step(mods, G, D, og, od, inp, tgt, rec)
        for k, v in rec.items():
            data['s%d_%s' % (s, k)] = v
        print('step_sn', s, rec['scalars'])
    np.savez_compressed(os.path.join(OUT, 'step_sn_n2_64.npz'), **data)


def _grad_pack(mod, x, extra_inputs=()):
    x = x.clone().requires_grad_(True)
    y = mod(x, *extra_inputs) if extra_inputs else mod(x)
    g = torch.Generator().manual_seed(99)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    return y.detach().numpy(), dy.numpy(), x.grad.numpy(), param_digests(mod, True)


def gen_blocks(mods):
    """Block-level fixtures: seeds regenerate weights; store x, y, dy, dx, grad digests, buffers."""
    archs, msg, losses, metrics, norm = mods
    data = {}
    g = torch.Generator().manual_seed(5)
    # BasicBlock (archs.py:205-241), two shapes incl. a concat-sized input
    for tag, (cin, cout, hw) in dict(bb_a=(8, 16, 12), bb_b=(48, 32, 8), bb_c=(3, 64, 16)).items():
        torch.manual_seed(11)
        m = archs.BasicBlock(cin, cout); m.train()
        x = torch.randn(2, cin, hw, hw, generator=g)
        y, dy, dx, gd = _grad_pack(m, x)
        data.update({tag + '_x': x.numpy(), tag + '_y': y, tag + '_dy': dy, tag + '_dx': dx, tag + '_gd': gd,
                     tag + '_bufs': buffer_digests(m), tag + '_cfg': np.array([cin, cout, hw])})
    # SPADE as wired (normalization.py:67-122); nhidden = C/16 -> max(.,4)
    for tag, (c, hw) in dict(sp_a=(64, 8), sp_b=(128, 6)).items():
        torch.manual_seed(12)
        m = norm.SPADE('spadebatch3x3', c, 3, c / 16); m.train()
        x = torch.randn(2, c, hw, hw, generator=g)
        x2 = x.clone().requires_grad_(True)
        y = m(x2, x2)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
        y.backward(dy)
        data.update({tag + '_x': x.numpy(), tag + '_y': y.detach().numpy(), tag + '_dy': dy.numpy(),
                     tag + '_dx': x2.grad.numpy(), tag + '_gd': param_digests(m, True), tag + '_cfg': np.array([c, hw])})
    # ConvolutionalBlock (models_seg_gan.py:13-64): stride 1 no-BN, stride 2 with BN
    for tag, (cin, cout, s, bn, hw) in dict(cb_a=(3, 16, 1, False, 10), cb_b=(16, 16, 2, True, 10),
                                            cb_c=(8, 24, 2, True, 7)).items():
        torch.manual_seed(13)
        m = msg.ConvolutionalBlock(cin, cout, 3, s, bn, 'LeakyReLu'); m.train()
        x = torch.randn(2, cin, hw, hw, generator=g)
        y, dy, dx, gd = _grad_pack(m, x)
        data.update({tag + '_x': x.numpy(), tag + '_y': y, tag + '_dy': dy, tag + '_dx': dx, tag + '_gd': gd,
                     tag + '_cfg': np.array([cin, cout, s, int(bn), hw])})
    # Discriminator on a 96x96 input (adaptive pool 6x6 is then the identity) and 64x64 (2x2 -> 6x6)
    for tag, hw in dict(d_96=96, d_64=64).items():
        torch.manual_seed(14)
        m = msg.Discriminator(3, 3, 8, 8, 1024); m.train()      # narrow D (n_channels=8) to keep it small
        x = torch.randn(2, 3, hw, hw, generator=g)
        y, dy, dx, gd = _grad_pack(m, x)
        data.update({tag + '_x': x.numpy(), tag + '_y': y, tag + '_dy': dy, tag + '_dx': dx, tag + '_gd': gd})
    # Losses / metrics (losses.py:274-302,130-136; metrics.py:6-35)
    x = torch.randn(3, 3, 16, 16, generator=g) * 3
    t = (torch.rand(3, 3, 16, 16, generator=g) > 0.5).float()
    x2 = x.clone().requires_grad_(True)
    l = losses.BCEDiceLoss()(x2, t); l.backward()
    data.update(loss_x=x.numpy(), loss_t=t.numpy(), loss_val=np.array(l.item()), loss_dx=x2.grad.numpy(),
                loss_bce=np.array(losses.StableBCELoss()(x, t).item()),
                loss_iou=np.array(metrics.iou_score(x[:, 1:].clone(), t[:, 1:].clone())),
                loss_dice=np.array(metrics.dice_coef(x[:, 1:].clone(), t[:, 1:].clone())))
    np.savez_compressed(os.path.join(OUT, 'blocks.npz'), **data)
    print('blocks.npz written,', len(data), 'arrays')


def gen_stage1(mods):
    """Stage-1 trainer (train.py:68-137) on the reference's UNet_R_SS_v2 + BCEDiceLoss: Adam(lr 1e-4,
    weight_decay 1e-7) and the per-step weight clamp +-0.7 (config_v1.json:30-41); 2 steps at 2x3x64x64."""
    archs, msg, losses, metrics, _ = mods
    torch.manual_seed(41)
    model = archs.UNet_R_SS_v2(3, 3, False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    crit = losses.BCEDiceLoss()
    inp, tgt = synthetic_batch(2, 64, 64)
    data = {}
    model.train()
    for s in range(2):
        out = model(inp)
        out[torch.isnan(out)] = 0
        loss = crit(out, tgt)
        iou = metrics.iou_score(out[:, 1:3].clone(), tgt[:, 1:3].clone()); dice = metrics.dice_coef(out[:, 1:3].clone(), tgt[:, 1:3].clone())
        for p in model.parameters():
            p.data.clamp_(-0.7, 0.7)
        opt.zero_grad(); loss.backward()
        data['s%d_grads' % s] = param_digests(model, True)
        opt.step()
        data['s%d_params' % s] = param_digests(model)
        data['s%d_scalars' % s] = np.array([loss.item(), float(iou), float(dice)])
        data['s%d_logits' % s] = out.detach().numpy().copy()
        print('stage1 step', s, data['s%d_scalars' % s])
    np.savez_compressed(os.path.join(OUT, 'stage1_n2_64.npz'), **data)


def gen_archs(mods):
    """Every exported arch the build provides: logits, input gradient and
    per-parameter gradient digests on 4 x 3 x 64 x 64 (large enough that the deepest batch norms see
    >= 16 samples), seeds regenerate the weights."""
    archs = mods[0]
    data = {}
    g = torch.Generator().manual_seed(51)
    x = torch.randn(4, 3, 64, 64, generator=g)
    data['x'] = x.numpy()
    for name, ds in (('UNet', False), ('NestedUNet', False), ('NestedUNet', True), ('SSUNet', False), ('UNet_ori', False),
                     ('UNet_B_SS', False), ('UNet_R_SS', False), ('AttUNet', False)):
        torch.manual_seed(52)
        m = archs.__dict__[name](3, 3, ds); m.train()
        xr = x.clone().requires_grad_(True)
        out = m(xr)
        outs = out if isinstance(out, list) else [out]
        tot = 0
        for i, o in enumerate(outs):
            dy = torch.randn(o.shape, generator=torch.Generator().manual_seed(99 + i))
            tot = tot + (o * dy).sum()
        tot.backward()
        tag = name + ('_ds' if ds else '')
        data[tag + '_y'] = np.stack([o.detach().numpy() for o in outs])[..., ::2, ::2].copy()     # every 2nd pixel: keeps the file small
        data[tag + '_dx'] = xr.grad.numpy()[..., ::2, ::2].copy()
        data[tag + '_gd'] = param_digests(m, True)
        print(tag, data[tag + '_y'].shape)
    np.savez_compressed(os.path.join(OUT, 'archs.npz'), **data)


def gen_infer(mods):
    """Eval-mode generator as the sliding-window API drives it (aerial_image_segmentation_api.py:376-390):
    one train-mode forward to give the batch norms non-trivial running statistics, then model.eval(),
    one patch per forward, sigmoid.  Also the patch order of patch_gen on a small image."""
    archs = mods[0]
    torch.manual_seed(41)
    model = archs.UNet_R_SS_v2(3, 3, False)
    model.train()
    inp, _ = synthetic_batch(2, 64, 64)
    with torch.no_grad():
        model(inp)
    model.eval()
    g = torch.Generator().manual_seed(61)
    patches = torch.rand(5, 3, 64, 64, generator=g)
    outs = []
    with torch.no_grad():
        for p_ in patches:
            outs.append(torch.sigmoid(model(p_.unsqueeze(0)))[0].numpy())
    # patch_gen order: needs cv2-free import of the function only -> restate the call through the module source
    import importlib.util, types as _t
    src = open(os.path.join(REF, 'aerial_image_segmentation_api.py')).read()
    start = src.index('def patch_gen('); end = src.index('def patch_merge(')
    ns = {'math': __import__('math')}
    exec(compile(src[start:end], 'patch_gen_extract', 'exec'), ns)         # runs the reference's own function text in memory
    img = np.arange(40 * 56 * 1).reshape(40, 56, 1)
    ip, _ = ns['patch_gen'](img, img, 16, 0.5)
    origins = np.array([[int(p_[0, 0, 0]) // 56, int(p_[0, 0, 0]) % 56] for p_ in ip])
    np.savez_compressed(os.path.join(OUT, 'infer.npz'), patches=patches.numpy(), probs=np.stack(outs), origins=origins,
                        img_hw=np.array([40, 56]), p_size=np.array(16))
    print('infer.npz', np.stack(outs).shape, origins.shape)


def gen_unwired(mods):
    """Golden vectors for the named-but-unwired blocks, from the
    reference's own batchnorm.py, archs.up_conv, xresidualblock.py, spectral_norm.py and
    efficientnet_pytorch (all import as-is)."""
    import batchnorm as ref_bn, xresidualblock as ref_x, spectral_norm as ref_sn, efficientnet_pytorch as ref_e
    archs = mods[0]
    g = torch.Generator().manual_seed(21)
    data = {}
    # A9: _compute_mean_std on reduced sums of two "replicas" + the fused output formula (batchnorm.py:75,115-127)
    torch.manual_seed(31)
    bn = ref_bn.SynchronizedBatchNorm2d(8)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(8, generator=g) + 0.5); bn.bias.copy_(torch.randn(8, generator=g))
    xa = torch.randn(2, 8, 6, 6, generator=g) * 2 + 1; xb = torch.randn(2, 8, 6, 6, generator=g) * 2 + 1
    parts = [xa.view(2, 8, -1), xb.view(2, 8, -1)]
    s = sum(p.sum(dim=0).sum(dim=-1) for p in parts); ss = sum((p ** 2).sum(dim=0).sum(dim=-1) for p in parts)
    mean, inv_std = bn._compute_mean_std(s, ss, 2 * 2 * 36)
    ya = (xa - mean.view(1, 8, 1, 1)) * (inv_std * bn.weight).view(1, 8, 1, 1) + bn.bias.view(1, 8, 1, 1)
    data.update(sbn_xa=xa.numpy(), sbn_xb=xb.numpy(), sbn_w=bn.weight.detach().numpy(), sbn_b=bn.bias.detach().numpy(),
                sbn_mean=mean.numpy(), sbn_inv_std=inv_std.numpy(), sbn_ya=ya.detach().numpy(),
                sbn_running_mean=bn.running_mean.numpy(), sbn_running_var=bn.running_var.numpy())
    # A13: up_conv
    torch.manual_seed(32)
    m = archs.up_conv(16, 8); m.train()
    x = torch.randn(2, 16, 5, 6, generator=g)
    y, dy, dx, gd = _grad_pack(m, x)
    data.update(up_x=x.numpy(), up_y=y, up_dy=dy, up_dx=dx, up_gd=gd)
    # A11: xResidualBlock (64 channels as in SURVEY 8a, and a small one)
    for tag, (c, hw) in dict(xr_a=(16, 12), xr_b=(64, 10)).items():
        torch.manual_seed(33)
        m = ref_x.xResidualBlock(c, c); m.train()
        x = torch.randn(2, c, hw, hw, generator=g)
        y, dy, dx, gd = _grad_pack(m, x)
        data.update({tag + '_x': x.numpy(), tag + '_y': y, tag + '_dy': dy, tag + '_dx': dx, tag + '_gd': gd,
                     tag + '_cfg': np.array([c, hw]), tag + '_nparams': np.array(sum(p.numel() for p in m.parameters()))})
    # A12: spectral_norm on a conv: u/v/sigma/weight after 1 and 2 training forwards, eval forward, grads
    torch.manual_seed(34)
    conv = ref_sn.spectral_norm(nn.Conv2d(8, 12, 3, padding=1))
    x = torch.randn(2, 8, 6, 6, generator=g)
    data.update(sn_x=x.numpy(), sn_w_orig=conv.weight_orig.detach().numpy().copy(), sn_u0=conv.weight_u.numpy().copy(),
                sn_v0=conv.weight_v.numpy().copy(), sn_bias=conv.bias.detach().numpy().copy())
    conv.train()
    for it in (1, 2):
        xr = x.clone().requires_grad_(True)
        conv.zero_grad()
        y = conv(xr)
        dyv = torch.randn(y.shape, generator=torch.Generator().manual_seed(99))
        y.backward(dyv)
        data.update({'sn_y%d' % it: y.detach().numpy(), 'sn_u%d' % it: conv.weight_u.numpy().copy(), 'sn_v%d' % it: conv.weight_v.numpy().copy(),
                     'sn_w%d' % it: conv.weight.detach().numpy().copy(), 'sn_dworig%d' % it: conv.weight_orig.grad.numpy().copy(),
                     'sn_dx%d' % it: xr.grad.numpy().copy()})
    data['sn_dy'] = dyv.numpy()
    conv.eval()
    data['sn_y_eval'] = conv(x).detach().numpy()
    # A10: MBConvBlock variants (train mode, drop_connect off) + EfficientNet-B0 extract_features on a small image
    from efficientnet_pytorch.utils import BlockArgs, GlobalParams
    gp = GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, dropout_rate=0.2, num_classes=10, width_coefficient=1.0,
                      depth_coefficient=1.0, depth_divisor=8, min_depth=None, drop_connect_rate=0.2, image_size=224)
    cases = dict(mb_a=(3, 1, 16, 16, 1, 12), mb_b=(3, 2, 16, 24, 6, 12), mb_c=(5, 2, 24, 40, 6, 11), mb_d=(5, 1, 40, 40, 6, 8))
    for tag, (k, s_, inp, out, e, hw) in cases.items():
        torch.manual_seed(35)
        ba = BlockArgs(kernel_size=k, num_repeat=1, input_filters=inp, output_filters=out, expand_ratio=e, id_skip=True,
                       stride=[s_], se_ratio=0.25)
        m = ref_e.model.MBConvBlock(ba, gp); m.train()
        x = torch.randn(2, inp, hw, hw, generator=g)
        y, dy, dx, gd = _grad_pack(m, x)
        data.update({tag + '_x': x.numpy(), tag + '_y': y, tag + '_dy': dy, tag + '_dx': dx, tag + '_gd': gd,
                     tag + '_cfg': np.array([k, s_, inp, out, e, hw]), tag + '_bufs': buffer_digests(m)})
    torch.manual_seed(36)
    net = ref_e.EfficientNet.from_name('efficientnet-b0', override_params=dict(drop_connect_rate=0.0)); net.train()
    x = torch.randn(2, 3, 64, 64, generator=g)
    xr = x.clone().requires_grad_(True)
    f = net.extract_features(xr)
    dyf = torch.randn(f.shape, generator=torch.Generator().manual_seed(99))
    f.backward(dyf)
    feat_params = [p for n_, p in net.named_parameters() if not n_.startswith('_fc')]
    data.update(eff_x=x.numpy(), eff_feat=f.detach().numpy(), eff_dy=dyf.numpy(), eff_dx=xr.grad.numpy(),
                eff_gd=np.stack([digest(p.grad) for p in feat_params]), eff_init=np.stack([digest(p) for p in net.parameters()]),
                eff_keys=np.array(list(net.state_dict().keys())))
    net.eval()
    with torch.no_grad():
        data['eff_feat_eval'] = net.extract_features(x).numpy()
    np.savez_compressed(os.path.join(OUT, 'unwired.npz'), **data)
    print('unwired.npz written,', len(data), 'arrays')


def gen_effb4(mods):
    """BASELINE config 4 at full size: EfficientNet-B4 `extract_features` forward + backward on 4 x 3 x 1024 x 1024 (the
    reference defines no B4 U-Net: SURVEY.md 0), drop_connect_rate 0 (the RNG-free setting), train mode.  The feature map
    (4 x 1792 x 32 x 32) and the input gradient are stored as digests plus a strided subset; inputs come from seeds."""
    import efficientnet_pytorch as ref_e
    torch.manual_seed(37)
    net = ref_e.EfficientNet.from_name('efficientnet-b4', override_params=dict(drop_connect_rate=0.0)); net.train()
    g = torch.Generator().manual_seed(23)
    x = torch.randn(4, 3, 1024, 1024, generator=g)
    xr = x.clone().requires_grad_(True)
    f = net.extract_features(xr)
    dyf = torch.randn(f.shape, generator=torch.Generator().manual_seed(99))
    f.backward(dyf)
    feat_params = [(n_, p) for n_, p in net.named_parameters() if not n_.startswith('_fc')]
    data = dict(shape=np.array(list(x.shape)), seed_model=np.array(37), seed_x=np.array(23), seed_dy=np.array(99),
                feat_shape=np.array(list(f.shape)), feat_digest=digest(f), feat_sub=f.detach()[:, ::16, ::4, ::4].numpy().copy(),
                dx_digest=digest(xr.grad), dx_sub=xr.grad[:, :, ::32, ::32].numpy().copy(),
                gd=np.stack([digest(p.grad) for _, p in feat_params]), names=np.array([n_ for n_, _ in feat_params]),
                init=np.stack([digest(p) for p in net.parameters()]), bufs=buffer_digests(net))
    np.savez_compressed(os.path.join(OUT, 'effb4_n4_1024.npz'), **data)
    print('effb4_n4_1024.npz written; feature digest', data['feat_digest'])


def gen_dp(mods, world=2, b=2, h=64, w=64, steps=2):
    """Data-parallel oracle ("sync-BN-at-W == single-process batch W*b"): the reference's modules in ONE
    process on the concatenated batch of W ranks, every BatchNorm2d converted by the reference's own `batchnorm.convert_model`
    to its SynchronizedBatchNorm2d and run through that class's parallel-training branch (batchnorm.py:57-80) with the statistics
    of the whole batch handed to ITS `_compute_mean_std` (batchnorm.py:115-127: biased variance clamped at eps for the
    normalisation, unbiased variance into the running estimate).  Only the device transport between the replicas
    (`SyncMaster.run_master` -> ReduceAddCoalesced / Broadcast, comm.py + batchnorm.py:92-113: CUDA-only, and the identity
    for a single contributor) is replaced by a direct call.  The step sequence is ref_step (train_seg_gan.py:182-233)."""
    import batchnorm as ref_bn
    G, D, _, _ = ref_models(mods)
    G = ref_bn.convert_model(G); D = ref_bn.convert_model(D)
    nconv = 0
    for m in list(G.modules()) + list(D.modules()):
        if isinstance(m, ref_bn._SynchronizedBatchNorm):
            m._is_parallel = True; m._parallel_id = 0
            m._sync_master.run_master = (lambda mod: (lambda msg: mod._compute_mean_std(msg.sum, msg.ssum, msg.sum_size)))(m)
            nconv += 1
    og = torch.optim.Adam(params=filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    od = torch.optim.Adam(params=filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    n = world * b
    inp, tgt = synthetic_batch(n, h, w)
    data = dict(seed_model=np.array(41), seed_batch=np.array(7), shape=np.array([n, 3, h, w]), world=np.array(world),
                n_sync_bn=np.array(nconv), input=inp.numpy(), target=tgt.numpy(),
                param_names_G=np.array([k for k, _ in G.named_parameters()]),
                param_names_D=np.array([k for k, _ in D.named_parameters()]))
    for s in range(steps):
        rec = {}
        ref_step(mods, G, D, og, od, inp, tgt, rec)
        for k, v in rec.items():
            data['s%d_%s' % (s, k)] = v
        print('step_dp', 'step', s, rec['scalars'])
    np.savez_compressed(os.path.join(OUT, 'step_dp_w%d_n%d_%d.npz' % (world, n, h)), **data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    mods = import_reference()
    if a.only in (None, 'blocks'):
        gen_blocks(mods)
    if a.only in (None, 'unwired'):
        gen_unwired(mods)
    if a.only in (None, 'stage1'):
        gen_stage1(mods)
    if a.only in (None, 'archs'):
        gen_archs(mods)
    if a.only in (None, 'infer'):
        gen_infer(mods)
    if a.only in (None, 'step64'):
        gen_step(mods, 'step_n2_64', 2, 64, 64, steps=2)
    if a.only in (None, 'step256'):
        gen_step(mods, 'step_n4_256', 4, 256, 256, steps=1, keep_logits=False)
    if a.only in (None, 'stepsn'):
        gen_step_sn(mods)
    if a.only in (None, 'dp'):
        gen_dp(mods)
    if a.only == 'effb4':
        gen_effb4(mods)                 # ~15 GB RSS, a minute of the reference on 8 cores: on request only
    if a.only == 'step512':
        # BASELINE config 2 at full size (16 x 3 x 512 x 512): ~45 GB RSS and a few minutes of the reference on 8 cores, so
        # it only runs on request (`--only step512`); logits are stored down-sampled 8x (as step256) + digests + scalars.
        gen_step(mods, 'step_n16_512', 16, 512, 512, steps=1, keep_logits=False)


if __name__ == '__main__':
    main()
