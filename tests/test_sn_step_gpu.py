"""BASELINE config 5, "spectral-norm discriminator on": spectral_norm() on the eight discriminator convs, then the G+D step,
against the reference's run of the same thing (tests/golden/step_sn_n2_64.npz: reference spectral_norm.py + models_seg_gan.py
through the train_seg_gan.py:182-233 sequence, two steps so that the in-place u / v power-iteration state carries over)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN


def _digests(ts):
    rows = []
    for t in ts:
        t = t.detach().double().cpu()
        rows.append([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])
    return np.array(rows)


def _build(pkg):
    torch.manual_seed(41)
    G = pkg.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False))
    D = pkg.models_seg_gan.Discriminator(3, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024)
    convs = [m for m in D.modules() if isinstance(m, nn.Conv2d)]
    for m in convs:                                   # on the host, as the fixture: same RNG draws for u, v
        pkg.spectral_norm.spectral_norm(m)
    return G, D, convs


def test_sn_discriminator_state_dict_surface(pkg):
    gold = np.load(os.path.join(GOLDEN, 'step_sn_n2_64.npz'))
    G, D, convs = _build(pkg)
    assert list(D.state_dict().keys()) == [str(k) for k in gold['state_keys_D']]
    assert [k for k, _ in D.named_parameters()] == [str(k) for k in gold['param_names_D']]
    assert np.allclose(_digests(D.parameters()), gold['init_D'], rtol=1e-9, atol=1e-12)
    assert np.allclose(_digests(D.buffers()), gold['init_uv'], rtol=1e-6, atol=1e-9), 'u / v initial vectors differ from the reference'
    sd = D.state_dict()
    assert sd._metadata['conv_blocks.0.conv_block.0']['spectral_norm'] == {'weight.version': 1}
    # round trip, and an unversioned (pre-version-1) checkpoint: weight present, weight_v absent
    G2, D2, _ = _build(pkg)
    D2.load_state_dict(sd)
    old = {k: v.clone() for k, v in sd.items()}
    truth = {}
    for i in range(8):
        key = 'conv_blocks.%d.conv_block.0.' % i
        wm = old[key + 'weight_orig'].reshape(old[key + 'weight_orig'].shape[0], -1)
        u = torch.nn.functional.normalize(torch.mv(wm, old[key + 'weight_v']), dim=0)      # the invariant of a trained checkpoint
        old[key + 'weight_u'] = u
        sigma = torch.dot(u, torch.mv(wm, old[key + 'weight_v']))
        old[key + 'weight'] = old[key + 'weight_orig'] / sigma
        del old[key + 'weight_v']
        truth[i] = (wm, u, sigma)
    D2.load_state_dict(old)                                                                # a plain dict has no _metadata: unversioned
    for i, (wm, u, sigma) in truth.items():
        v_rec = D2.conv_blocks[i].conv_block[0].weight_v
        su = torch.dot(u, torch.mv(wm, v_rec))
        assert abs(su.item() - sigma.item()) < 1e-3 * abs(sigma.item()), i
        assert torch.allclose(torch.nn.functional.normalize(torch.mv(wm, v_rec), dim=0), u, atol=1e-3), i


@pytest.mark.gpu
def test_sn_discriminator_in_the_gan_step(pkg, dev):
    gold = np.load(os.path.join(GOLDEN, 'step_sn_n2_64.npz'))
    G, D, convs = _build(pkg)
    G.to(dev).train(); D.to(dev).train()
    og = torch.optim.Adam(params=filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    od = torch.optim.Adam(params=filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(2, 3, 64, 64, generator=g).to(dev); tgt = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float().to(dev)
    calls = []
    D.register_forward_hook(lambda m, i, o: calls.append(o.detach().cpu().numpy().copy()))
    fused = []
    orig_sn = pkg.ops.conv2d_sn
    pkg.ops.conv2d_sn = lambda *a, **k: (fused.append(1), orig_sn(*a, **k))[1]
    for s in range(2):
        del calls[:]
        loss, iou, dice, closs, adv_g, adv_d = pkg.train_seg_gan.gan_step(inp, tgt, G, D, pkg.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(),
                                                                         nn.MSELoss(), og, od, 3)
        got = np.array([loss.item(), closs.item(), adv_g.item(), adv_d.item(), iou.item(), dice.item()])
        ref = gold['s%d_scalars' % s]
        tol = np.array([2e-5, 5e-5, 1e-4, 2e-4, 1e-4, 1e-4]) if s == 0 else np.array([2e-3, 5e-3, 3e-2, 3e-2, 5e-3, 2e-3])
        assert (np.abs(got - ref) < tol).all(), 'step %d scalars %s vs %s' % (s, got, ref)
        if s == 0:
            for k, v in zip(('sd', 'hr', 'sr'), calls):
                assert np.abs(v - gold['s0_' + k]).max() < 2e-4, (k, v, gold['s0_' + k])
            # D-step gradients of weight_orig flow through W/sigma (spectral_norm.py:86-88): golden digests of the same backward
            gd = _digests([p.grad for p in D.parameters()])
            ref_g = gold['s0_d_bwd_D']
            names = [k for k, _ in D.named_parameters()]
            rel = np.abs(gd[:, 2] - ref_g[:, 2]) / (ref_g[:, 2] + 1e-12)
            big = ref_g[:, 2] > 1e-6
            assert np.median(rel[big]) < 2e-3 and rel[big].max() < 0.1, [(names[i], rel[i]) for i in np.argsort(-rel * big)[:4]]
        # u, v after 3 power iterations per step (three D forwards), running stats
        bd = _digests(D.buffers())
        ref_b = gold['s%d_bufs_D' % s]
        assert np.allclose(bd[:, 1], ref_b[:, 1], rtol=5e-4 * (s + 1), atol=1e-4), 'u / v / running stats after step %d' % s
        pd = _digests(D.parameters())
        rd = gold['s%d_d_step_D' % s]
        numel = np.array([p.numel() for p in D.parameters()])
        assert (np.abs(pd[:, 1] - rd[:, 1]) <= 1e-5 * rd[:, 1] + (0.5 * numel + 2) * 2e-5 * (s + 1)).all()
    pkg.ops.conv2d_sn = orig_sn
    # the eight convs ran with the spectral norm inside the conv (sigma rides the weight pack; W / sigma never written out)
    assert len(fused) == 2 * 3 * 8, len(fused)
    # eval mode: no power iteration (spectral_norm.py:99-101 passes module.training)
    D.eval()
    u0 = [m.weight_u.clone() for m in convs]
    with torch.no_grad():
        D(tgt)
    assert all(torch.equal(a, m.weight_u) for a, m in zip(u0, convs))
