"""GPU parity of the named-but-unwired per-op blocks against golden vectors
generated from the reference's own modules (tests/golden/unwired.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN, 'unwired.npz'))


def _digest(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def _check(mod, gold, tag, dev, atol=3e-5, rtol=3e-5, gtol=2e-4, params=None):
    x = torch.from_numpy(gold[tag + '_x']).to(dev).requires_grad_(True)
    y = mod(x)
    yg = gold[tag + '_y']
    assert tuple(y.shape) == yg.shape
    err = np.abs(y.detach().cpu().numpy() - yg).max()
    assert err <= atol + rtol * np.abs(yg).max(), '%s fwd err %.3e' % (tag, err)
    y.backward(torch.from_numpy(gold[tag + '_dy']).to(dev))
    dxg = gold[tag + '_dx']
    err = np.abs(x.grad.cpu().numpy() - dxg).max()
    assert err <= atol + 2e-4 * np.abs(dxg).max(), '%s dx err %.3e (max %.3e)' % (tag, err, np.abs(dxg).max())
    gd = np.stack([_digest(p.grad) for p in (params or list(mod.parameters()))])
    ref = gold[tag + '_gd']
    l2err = np.abs(gd[:, 2] - ref[:, 2])
    assert (l2err <= gtol * ref[:, 2] + 2e-4).all(), '%s grad digests: worst %s' % (tag, l2err.max())


def test_sync_bn_formula_single_process_equivalent(pkg, dev, gold):
    """A9: two replicas' batches through the sync branch (clamp(var,eps)^-1/2, batchnorm.py:127) ==
    one process over the concatenated batch, which is how the kernels evaluate it."""
    bn = pkg.batchnorm.SynchronizedBatchNorm2d(8).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(gold['sbn_w'])); bn.bias.copy_(torch.from_numpy(gold['sbn_b']))
    x = torch.cat([torch.from_numpy(gold['sbn_xa']), torch.from_numpy(gold['sbn_xb'])], 0).to(dev)
    y = pkg.ops.batch_norm_act(x, bn, var_mode=1)
    assert np.abs(y[:2].detach().cpu().numpy() - gold['sbn_ya']).max() < 2e-5
    assert np.allclose(bn.running_mean.cpu().numpy(), gold['sbn_running_mean'], atol=1e-6)
    assert np.allclose(bn.running_var.cpu().numpy(), gold['sbn_running_var'], rtol=1e-5, atol=1e-6)
    from oracle import unwired_ops_cpu as U
    outs, mean, _ = U.sync_bn_forward([torch.from_numpy(gold['sbn_xa']), torch.from_numpy(gold['sbn_xb'])],
                                      torch.from_numpy(gold['sbn_w']), torch.from_numpy(gold['sbn_b']))
    assert np.abs(outs[0].numpy() - gold['sbn_ya']).max() < 1e-5 and np.allclose(mean.numpy(), gold['sbn_mean'], atol=1e-6)
    # module surface of the vendored package
    m = pkg.batchnorm.convert_model(nn.Sequential(nn.Conv2d(3, 8, 1), nn.BatchNorm2d(8)))
    assert isinstance(m[1], pkg.batchnorm.SynchronizedBatchNorm2d)


def test_up_conv(pkg, dev, gold):
    torch.manual_seed(32)
    m = pkg.archs.up_conv(16, 8).to(dev).train()
    _check(m, gold, 'up', dev)


@pytest.mark.parametrize('tag', ['xr_a', 'xr_b'])
def test_xresidualblock(pkg, dev, gold, tag):
    c, hw = [int(v) for v in gold[tag + '_cfg']]
    torch.manual_seed(33)
    m = pkg.xresidualblock.xResidualBlock(c, c).to(dev).train()
    assert sum(p.numel() for p in m.parameters()) == int(gold[tag + '_nparams'])
    _check(m, gold, tag, dev, atol=5e-5, rtol=5e-5, gtol=1e-3)


def test_spectral_norm(pkg, dev, gold):
    torch.manual_seed(34)
    conv = pkg.spectral_norm.spectral_norm(nn.Conv2d(8, 12, 3, padding=1))
    assert np.array_equal(conv.weight_orig.detach().numpy(), gold['sn_w_orig'])
    assert np.allclose(conv.weight_u.numpy(), gold['sn_u0'], atol=1e-7) and np.allclose(conv.weight_v.numpy(), gold['sn_v0'], atol=1e-7)
    assert set(conv.state_dict().keys()) == {'bias', 'weight_orig', 'weight_u', 'weight_v'}
    conv.to(dev).train()
    x = torch.from_numpy(gold['sn_x']).to(dev)
    dy = torch.from_numpy(gold['sn_dy']).to(dev)
    for it in (1, 2):
        xr = x.clone().requires_grad_(True)
        conv.zero_grad()
        conv._forward_pre_hooks[next(iter(conv._forward_pre_hooks))](conv, None)     # the registered hook
        y = pkg.ops.conv2d(xr, conv.weight, conv.bias, 1, 1)
        y.backward(dy)
        assert np.abs(conv.weight_u.cpu().numpy() - gold['sn_u%d' % it]).max() < 2e-6
        assert np.abs(conv.weight_v.cpu().numpy() - gold['sn_v%d' % it]).max() < 2e-6
        assert np.abs(conv.weight.detach().cpu().numpy() - gold['sn_w%d' % it]).max() < 2e-6
        assert np.abs(y.detach().cpu().numpy() - gold['sn_y%d' % it]).max() < 2e-5
        assert np.abs(conv.weight_orig.grad.cpu().numpy() - gold['sn_dworig%d' % it]).max() < 1e-4 * np.abs(gold['sn_dworig%d' % it]).max() + 1e-6
        assert np.abs(xr.grad.cpu().numpy() - gold['sn_dx%d' % it]).max() < 2e-5
    conv.eval()
    conv._forward_pre_hooks[next(iter(conv._forward_pre_hooks))](conv, None)
    y = pkg.ops.conv2d(x, conv.weight, conv.bias, 1, 1)
    assert np.abs(y.detach().cpu().numpy() - gold['sn_y_eval']).max() < 2e-5
    assert np.abs(conv.weight_u.cpu().numpy() - gold['sn_u2']).max() < 2e-6          # eval: no power iteration


@pytest.mark.parametrize('tag', ['mb_a', 'mb_b', 'mb_c', 'mb_d'])
def test_mbconv_block(pkg, dev, gold, tag):
    E = pkg.efficientnet_pytorch
    k, s, inp, out, e, hw = [int(v) for v in gold[tag + '_cfg']]
    gp = E.GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, dropout_rate=0.2, num_classes=10, width_coefficient=1.0,
                        depth_coefficient=1.0, depth_divisor=8, min_depth=None, drop_connect_rate=0.2, image_size=224)
    ba = E.BlockArgs(kernel_size=k, num_repeat=1, input_filters=inp, output_filters=out, expand_ratio=e, id_skip=True, stride=[s], se_ratio=0.25)
    torch.manual_seed(35)
    m = E.MBConvBlock(ba, gp).to(dev).train()
    _check(m, gold, tag, dev, atol=5e-5, rtol=1e-4, gtol=2e-3)
    bufs = np.stack([_digest(b.float()) for b in m.buffers()])
    assert np.allclose(bufs[:, 1], gold[tag + '_bufs'][:, 1], rtol=1e-4, atol=1e-5)


def test_efficientnet_b0_extract_features(pkg, dev, gold):
    E = pkg.efficientnet_pytorch
    torch.manual_seed(36)
    net = E.EfficientNet.from_name('efficientnet-b0', override_params=dict(drop_connect_rate=0.0))
    assert list(net.state_dict().keys()) == [str(k) for k in gold['eff_keys']]
    init = np.stack([_digest(p) for p in net.parameters()])
    assert np.allclose(init, gold['eff_init'], rtol=1e-9, atol=1e-12)
    net.to(dev).train()
    x = torch.from_numpy(gold['eff_x']).to(dev).requires_grad_(True)
    f = net.extract_features(x)
    ref = gold['eff_feat']
    e = np.abs(f.detach().cpu().numpy() - ref)
    assert e.max() < 2e-3 * np.abs(ref).max() and np.median(e) < 2e-5, 'features: max %.3e median %.3e' % (e.max(), np.median(e))
    f.backward(torch.from_numpy(gold['eff_dy']).to(dev))
    dxe = np.abs(x.grad.cpu().numpy() - gold['eff_dx'])
    assert dxe.max() < 5e-3 * np.abs(gold['eff_dx']).max()
    params = [p for n, p in net.named_parameters() if not n.startswith('_fc')]
    gd = np.stack([_digest(p.grad) for p in params])
    rel = np.abs(gd[:, 2] - gold['eff_gd'][:, 2]) / (gold['eff_gd'][:, 2] + 1e-12)
    names = [n for n, _ in net.named_parameters() if not n.startswith('_fc')]
    # a BN bias that feeds (through a 1x1 conv) another batch norm has an analytically ZERO gradient:
    # the reference's value there (~1e-4) is its own fp32 noise, so those are bounded absolutely
    big = gold['eff_gd'][:, 2] > 2e-3
    assert np.median(rel) < 1e-3 and rel[big].max() < 0.05, 'grad digests: median %.3e; worst %s' % (
        np.median(rel), [(names[i], rel[i], gold['eff_gd'][i, 2]) for i in np.argsort(-rel * big)[:4]])
    assert np.abs(gd[~big, 2] - gold['eff_gd'][~big, 2]).max() < 1e-3
    net.eval()
    with torch.no_grad():
        fe = net.extract_features(x.detach())
    assert np.abs(fe.cpu().numpy() - gold['eff_feat_eval']).max() < 1e-4 * np.abs(gold['eff_feat_eval']).max() + 1e-5
    with pytest.raises(NotImplementedError):
        net(x.detach())
