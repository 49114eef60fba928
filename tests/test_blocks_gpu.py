"""Block-level parity against golden vectors generated from the REFERENCE's own modules
(tests/golden/blocks.npz, made by oracle/gen_golden.py).  Weights are regenerated from the
recorded seeds, which also proves the constructors create parameters in the reference's order."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN, 'blocks.npz'))


def _digest(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def _check(mod, gold, tag, dev, fwd=None, atol=2e-5, rtol=2e-5):
    x = torch.from_numpy(gold[tag + '_x']).to(dev).requires_grad_(True)
    y = (fwd or mod)(x)
    yg = gold[tag + '_y']
    assert tuple(y.shape) == yg.shape
    err = np.abs(y.detach().cpu().numpy() - yg).max()
    assert err <= atol + rtol * np.abs(yg).max(), '%s fwd err %.3e' % (tag, err)
    y.backward(torch.from_numpy(gold[tag + '_dy']).to(dev))
    dxg = gold[tag + '_dx']
    err = np.abs(x.grad.cpu().numpy() - dxg).max()
    assert err <= atol + 1e-4 * np.abs(dxg).max(), '%s dx err %.3e (max %.3e)' % (tag, err, np.abs(dxg).max())
    gd = np.stack([_digest(p.grad) for p in mod.parameters()])
    ref = gold[tag + '_gd']
    l2err = np.abs(gd[:, 2] - ref[:, 2])
    assert (l2err <= 1e-4 * ref[:, 2] + 1e-5).all(), '%s grad digests: %s' % (tag, l2err.max())
    sumerr = np.abs(gd[:, 0] - ref[:, 0])
    assert (sumerr <= 2e-4 * ref[:, 1] + 2e-4).all(), '%s grad sums: %s' % (tag, sumerr.max())


@pytest.mark.parametrize('tag', ['bb_a', 'bb_b', 'bb_c'])
def test_basic_block(pkg, dev, gold, tag):
    cin, cout, hw = [int(v) for v in gold[tag + '_cfg']]
    torch.manual_seed(11)
    m = pkg.archs.BasicBlock(cin, cout).to(dev).train()
    _check(m, gold, tag, dev)
    bufs = np.stack([_digest(b.float()) for b in m.buffers()])
    assert np.allclose(bufs, gold[tag + '_bufs'], rtol=1e-4, atol=1e-5)


def test_basic_block_concat_equals_cat(pkg, dev, gold):
    """Two-pointer input == torch.cat input (archs.py:651-667)."""
    torch.manual_seed(11)
    m = pkg.archs.BasicBlock(48, 32).to(dev).train()
    x = torch.from_numpy(gold['bb_b_x']).to(dev)
    xa = x[:, :32].contiguous().requires_grad_(True); xb = x[:, 32:].contiguous().requires_grad_(True)
    y = m(xa, xb)
    assert np.abs(y.detach().cpu().numpy() - gold['bb_b_y']).max() < 5e-5
    y.backward(torch.from_numpy(gold['bb_b_dy']).to(dev))
    dx = torch.cat([xa.grad, xb.grad], 1).cpu().numpy()
    assert np.abs(dx - gold['bb_b_dx']).max() < 5e-5


@pytest.mark.parametrize('cin,cout,h,w,nb', [(64, 64, 48, 64, 4), (128, 128, 40, 72, 2), (64, 128, 33, 50, 3)])
def test_basic_block_bn_apply_fused_into_conv2(pkg, dev, cin, cout, h, w, nb):
    """relu(bn1(conv1(x))) applied on conv2's input (ssg_conv_desc.in_scale, archs.py:229-230) instead of being written out: the
    transform is bn_apply's own expression on the same fp32 values, so the block's output, its input gradient and every parameter
    gradient keep their bits; the materialising kernel must not run, and the halo pixels outside the image stay zeros of the
    ACTIVATED tensor (a shift > 0 would otherwise leak relu(shift) into the border outputs)."""
    ops = pkg.ops
    call = pkg._lib.call
    torch.manual_seed(5)
    m = pkg.archs.BasicBlock(cin, cout).to(dev).train()
    with torch.no_grad():
        m.bn1.bias.fill_(0.7)                              # relu(shift) != 0: the padding must not be transformed
        m.bn1.weight.uniform_(0.5, 1.5)
    x0 = torch.randn(nb, cin, h, w, device=dev)
    dy = torch.randn(nb, cout, h, w, device=dev)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    applied = []
    real_apply = ops._bn_apply

    def counting_apply(x, stats, res, act, slope):
        if res is None:                                    # bn1's apply (bn2's carries the shortcut as its residual)
            applied.append(tuple(x.shape))
        return real_apply(x, stats, res, act, slope)
    outs = {}
    call('ssg_conv_set_k32_mode', 2)
    call('ssg_wgrad_set_k32_mode', 1)
    ops._bn_apply = counting_apply
    saved_fuse = ops.BN_FUSE_INPUT
    try:
        for fused in (False, True):
            m.load_state_dict(state)
            ops.BN_FUSE_INPUT = fused
            del applied[:]
            x = x0.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            y = m(x)
            y.backward(dy)
            outs[fused] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in m.parameters()], len(applied),
                           [b.clone() for b in m.buffers()])
    finally:
        ops._bn_apply = real_apply
        ops.BN_FUSE_INPUT = saved_fuse
        call('ssg_conv_set_k32_mode', 1)
    assert outs[True][3] == 0, 'bn1 was written out %d times on the fused route (forward or weight gradient declined)' % outs[True][3]
    assert outs[False][3] == 1
    assert torch.equal(outs[True][0], outs[False][0]), (outs[True][0] - outs[False][0]).abs().max().item()
    assert torch.equal(outs[True][1], outs[False][1]), (outs[True][1] - outs[False][1]).abs().max().item()
    for a, b in zip(outs[True][2], outs[False][2]):
        assert torch.equal(a, b), (a - b).abs().max().item()
    for a, b in zip(outs[True][4], outs[False][4]):
        assert torch.equal(a, b)
    # and against stock torch autograd on the same weights (fp32 tolerance of the split kernels)
    ref = torch.nn.Sequential()
    import torch.nn.functional as F
    xr = x0.clone().requires_grad_(True)
    sd = {k: v.to(dev) for k, v in state.items()}
    c1 = F.conv2d(xr, sd['conv1.weight'], None, 1, 1)
    y1 = F.relu(F.batch_norm(c1, None, None, sd['bn1.weight'], sd['bn1.bias'], True, 0.1, 1e-5))
    c2 = F.conv2d(y1, sd['conv2.weight'], None, 1, 1)
    o = F.batch_norm(c2, None, None, sd['bn2.weight'], sd['bn2.bias'], True, 0.1, 1e-5)
    o = F.relu(o + (F.conv2d(xr, sd['shortcut.0.weight'], None, 1, 0) if 'shortcut.0.weight' in sd else xr))
    # forward output against stock torch (the backward is pinned by the bit-for-bit agreement with the materialising route, which the
    # golden vectors cover: a ReLU whose pre-activation is within rounding of zero flips between implementations and, through the
    # batch-norm sums of so few pixels, moves every gradient element of its channel -- no tight bound on dx holds against torch)
    err = (outs[True][0] - o.detach()).abs()
    assert err.max().item() <= 1e-4 * max(1.0, o.abs().max().item()), err.max().item()


@pytest.mark.parametrize('cin,cout,h,w,nb', [(64, 64, 48, 64, 4), (128, 128, 40, 72, 2), (64, 128, 33, 50, 3)])
def test_basic_block_bn1_backward_sums_from_the_dgrad_epilogue(pkg, dev, cin, cout, h, w, nb):
    """bn1's backward sums (sum g, sum g * xhat) from the epilogue of conv2's input gradient (ssg_conv_desc.bwd_x, archs.py:229-230) instead
    of a reduce pass over (dy1, c1): the forward is untouched (same bits), every gradient agrees with the two-pass route to fp32
    accumulation error (the sums are formed in a different order: fp32 inside a 16-lane group of the tile, fp64 beyond)."""
    ops = pkg.ops
    call = pkg._lib.call
    torch.manual_seed(6)
    m = pkg.archs.BasicBlock(cin, cout).to(dev).train()
    x0 = torch.randn(nb, cin, h, w, device=dev)
    dy = torch.randn(nb, cout, h, w, device=dev)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    used = []
    real = ops._bn_bwd_from_partials

    def counting(*a, **k):
        used.append(1)
        return real(*a, **k)
    outs = {}
    saved = ops.BN_BWD_EPILOGUE
    call('ssg_conv_set_k32_mode', 2)
    ops._bn_bwd_from_partials = counting
    try:
        for on in (False, True):
            m.load_state_dict(state)
            ops.BN_BWD_EPILOGUE = on
            del used[:]
            x = x0.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            y = m(x)
            y.backward(dy)
            outs[on] = (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in m.parameters()], len(used))
    finally:
        ops._bn_bwd_from_partials = real
        ops.BN_BWD_EPILOGUE = saved
        call('ssg_conv_set_k32_mode', 1)
    assert outs[True][3] == 1 and outs[False][3] == 0, (outs[True][3], outs[False][3])
    assert torch.equal(outs[True][0], outs[False][0])
    for a, b, what in [(outs[True][1], outs[False][1], 'dx')] + [(a, b, 'param %d' % i) for i, (a, b) in enumerate(zip(outs[True][2], outs[False][2]))]:
        err = (a - b).abs().max().item()
        assert err <= 2e-5 * max(1.0, b.abs().max().item()), '%s: %.3e (max %.3e)' % (what, err, b.abs().max().item())


@pytest.mark.parametrize('tag', ['sp_a', 'sp_b'])
def test_spade(pkg, dev, gold, tag):
    c, hw = [int(v) for v in gold[tag + '_cfg']]
    torch.manual_seed(12)
    m = pkg.normalization.SPADE('spadebatch3x3', c, 3, c / 16).to(dev).train()
    _check(m, gold, tag, dev, fwd=lambda x: m(x, x))


@pytest.mark.parametrize('c,h,w', [(64, 256, 257), (128, 130, 515)])
def test_spade_fused_gamma_beta_modulate(pkg, dev, c, h, w):
    """SPADE at a size the fused kernel takes (4-channel hidden activation, >= 65536 pixels: gamma|beta conv + modulation in
    ssg_spade_conv_modulate_f32, gamma-only tensor kept for the backward) against the block's arithmetic in torch fp64 on the
    CPU (normalization.py:110-120), forward and every gradient."""
    import torch.nn.functional as F
    torch.manual_seed(21)
    # ragged widths: edge strips of the 32-pixel kernel; c = 64 -> nhidden 4 (8-byte loads), c = 128 -> nhidden 8 (16-byte loads)
    m = pkg.normalization.SPADE('spadebatch3x3', c, 3, c / 16).to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, c, h, w, generator=g); dy = torch.randn(1, c, h, w, generator=g)
    xd = x.to(dev).requires_grad_(True)
    pkg.ops.PROFILE = []
    try:
        yd = m(xd, xd)
        labels = [r[0] for r in pkg.ops.PROFILE]
    finally:
        pkg.ops.PROFILE = None
    assert 'thin32_cin_kernel<spade>' in labels, labels
    yd.backward(dy.to(dev))
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.named_parameters()}
    xr = x.double().requires_grad_(True)
    seg = F.conv2d(xr, P['x2map.weight'], P['x2map.bias'], 1, 1)
    a = F.relu(F.conv2d(seg, P['mlp_shared.0.weight'], P['mlp_shared.0.bias'], 1, 1))
    gam = F.conv2d(a, P['mlp_gamma.weight'], P['mlp_gamma.bias'], 1, 1); bet = F.conv2d(a, P['mlp_beta.weight'], P['mlp_beta.bias'], 1, 1)
    yr = xr * (1 + gam) + bet
    yr.backward(dy.double())

    def close(got, ref, nm, rtol=2e-5):
        e = (got.detach().cpu().double() - ref).abs().max().item()
        assert e <= rtol * ref.abs().max().item() + 1e-6, '%s: max err %.3e (ref max %.3e)' % (nm, e, ref.abs().max().item())
    close(yd, yr.detach(), 'out')
    close(xd.grad, xr.grad, 'dx', 5e-5)
    for k, v in m.named_parameters():
        close(v.grad, P[k].grad, k, 2e-4)


@pytest.mark.parametrize('tag', ['cb_a', 'cb_b', 'cb_c'])
def test_convolutional_block(pkg, dev, gold, tag):
    cin, cout, s, bn, hw = [int(v) for v in gold[tag + '_cfg']]
    torch.manual_seed(13)
    m = pkg.models_seg_gan.ConvolutionalBlock(cin, cout, 3, s, bool(bn), 'LeakyReLu').to(dev).train()
    _check(m, gold, tag, dev)


@pytest.mark.parametrize('tag', ['d_96', 'd_64'])
def test_discriminator_small(pkg, dev, gold, tag):
    torch.manual_seed(14)
    m = pkg.models_seg_gan.Discriminator(3, 3, 8, 8, 1024).to(dev).train()
    _check(m, gold, tag, dev, atol=5e-5, rtol=1e-4)


def test_losses_and_metrics(pkg, dev, gold):
    x = torch.from_numpy(gold['loss_x']).to(dev).requires_grad_(True)
    t = torch.from_numpy(gold['loss_t']).to(dev)
    l = pkg.losses.BCEDiceLoss()(x, t)
    assert abs(l.item() - float(gold['loss_val'])) < 1e-5
    l.backward()
    assert np.abs(x.grad.cpu().numpy() - gold['loss_dx']).max() < 1e-8 + 1e-4 * np.abs(gold['loss_dx']).max()
    assert abs(pkg.losses.StableBCELoss()(x.detach(), t).item() - float(gold['loss_bce'])) < 1e-5
    xm, tm = x.detach()[:, 1:].clone(), t[:, 1:].clone()
    assert abs(pkg.metrics.iou_score(xm, tm) - float(gold['loss_iou'])) < 1e-6
    assert abs(pkg.metrics.dice_coef(xm, tm) - float(gold['loss_dice'])) < 1e-5
