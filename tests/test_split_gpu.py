"""fp32 convolution on the bf16 matrix pipe (csrc/conv_igemm_halo_x3.hip, SSG_MFMA_SPLIT=1): every fp32 operand is split
into three bf16 terms and the six leading products are accumulated in fp32.  The claim to hold: fp32-class accuracy --
the split kernel must be as close to an fp64 reference as the fp32-MFMA kernel is (same order of magnitude of max and
rms error), on forward and input gradient, incl. two-pointer inputs, residual epilogue and batch-norm partial rows."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(ops, split, fn):
    old = ops.MFMA_SPLIT
    ops.MFMA_SPLIT = split
    try:
        return fn()
    finally:
        ops.MFMA_SPLIT = old


@pytest.fixture(params=['x3', 'k32'])
def family(request, pkg):
    """Both split-operand families meet the same bounds: the 16-channel-chunk x3 kernels on v_mfma_f32_32x32x16_bf16 and the round-4
    k32 kernels on v_mfma_f32_16x16x32_bf16 (forced on: at these sizes the default dispatch keeps most launches on x3)."""
    k32 = request.param == 'k32'
    pkg._lib.call('ssg_conv_set_k32_mode', 2 if k32 else 0)
    pkg._lib.call('ssg_wgrad_set_k32_mode', 1 if k32 else 0)
    yield request.param
    pkg._lib.call('ssg_conv_set_k32_mode', 1)
    pkg._lib.call('ssg_wgrad_set_k32_mode', 1)


@pytest.mark.parametrize('c1,c2,co,hw,nb', [(128, 0, 128, 128, 6), (64, 0, 64, 128, 4), (64, 128, 64, 128, 4), (128, 256, 128, 128, 6), (256, 0, 384, 64, 8)])
def test_split_conv_matches_fp64_like_fp32_mfma(pkg, dev, c1, c2, co, hw, nb, family):
    ops = pkg.ops
    torch.manual_seed(21)
    torch.set_num_threads(16)
    xc = torch.randn(nb, c1 + c2, hw, hw) * 1.5 + 0.3
    wc = torch.randn(co, c1 + c2, 3, 3) / (3 * (c1 + c2) ** 0.5)
    rc = torch.randn(nb, co, hw, hw)
    ref = F.conv2d(xc.double(), wc.double(), None, 1, 1) + rc.double()
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    w = wc.to(dev); res = ops.to_nhwc(rc.to(dev))

    def fwd():
        return ops._conv_fwd_impl(x1, x2, w, None, 1, 1, 0, 0.0, res=res)
    y32 = _run(ops, False, fwd).cpu().double()
    ops.PROFILE = []
    try:
        y3 = _run(ops, True, fwd).cpu().double()
        labels = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert labels and ('halo_x3' if family == 'x3' else 'halo_k32') in labels[0], 'the %s kernel did not run: %s' % (family, labels)
    e32 = (y32 - ref).abs(); e3 = (y3 - ref).abs()
    assert e3.max().item() <= 2.0 * e32.max().item() + 1e-6, (e3.max().item(), e32.max().item())
    assert e3.pow(2).mean().sqrt().item() <= 2.0 * e32.pow(2).mean().sqrt().item() + 1e-8

    # input gradient (same kernel, transposed pack) and batch-norm partial rows
    dyc = torch.randn(nb, co, hw, hw)
    dy = ops.to_nhwc(dyc.to(dev))
    gref = torch.nn.grad.conv2d_input((nb, c1 + c2, hw, hw), wc.double(), dyc.double(), 1, 1)[:, :c1]

    def dgrad():
        return ops._conv_dgrad_impl(dy, w, 1, 1, hw, hw, 0, c1)
    g32 = _run(ops, False, dgrad).cpu().double(); g3 = _run(ops, True, dgrad).cpu().double()
    assert (g3 - gref).abs().max().item() <= 2.0 * (g32 - gref).abs().max().item() + 1e-6

    # weight gradient: both operands are activations, both split on the fly (wgrad_halo_x3_kernel)
    wref = torch.nn.grad.conv2d_weight(xc.double(), wc.shape, dyc.double(), 1, 1)

    def wgrad():
        return ops._conv_wgrad_impl(x1, x2, dy, tuple(w.shape), 1, 1)
    w32 = _run(ops, False, wgrad).cpu().double()
    ops.PROFILE = []
    try:
        w3 = _run(ops, True, wgrad).cpu().double()
        wl = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert wl and ('wgrad_halo_x3' if family == 'x3' else 'wgrad_k32') in wl[0], 'the %s weight-gradient kernel did not run: %s' % (family, wl)
    ew32 = (w32 - wref).abs(); ew3 = (w3 - wref).abs()
    assert ew3.max().item() <= 2.0 * ew32.max().item() + 1e-6 * wref.abs().max().item(), (ew3.max().item(), ew32.max().item())
    assert ew3.pow(2).mean().sqrt().item() <= 2.0 * ew32.pow(2).mean().sqrt().item() + 1e-8

    def stats():
        return ops._conv_fwd_impl(x1, x2, w, None, 1, 1, 0, 0.0, want_bn=True)
    yb, pb = _run(ops, True, stats)
    assert pb is not None and pb.numel() > 0
    # the partial rows are the column sums of what the kernel wrote (fp32 deviations from a pivot, widened to fp64)
    yd = yb.double()
    s1 = yd.sum((0, 2, 3)); s2 = (yd * yd).sum((0, 2, 3))
    tot = pb.sum(0)
    assert torch.allclose(tot[0], s1, rtol=1e-7, atol=1e-6 * yd.abs().sum((0, 2, 3)).max().item())
    assert torch.allclose(tot[1], s2, rtol=1e-6)


def test_split_step_meets_the_step_tolerances(pkg, dev):
    """The whole G+D step with the split kernels on: the golden 2 x 64^2 step at the tolerances of tests/test_step_gpu.py."""
    import os
    import numpy as np
    import torch.nn as nn
    from conftest import GOLDEN
    ops = pkg.ops
    gold = np.load(os.path.join(GOLDEN, 'step_n2_64.npz'))
    torch.manual_seed(41)
    G = pkg.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
    D = pkg.models_seg_gan.Discriminator(3, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024).to(dev).train()
    og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(2, 3, 64, 64, generator=g).to(dev); tgt = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float().to(dev)
    tap = {}
    G.register_forward_hook(lambda m, i, o: tap.__setitem__('logits', o.detach().clone()))
    out = _run(ops, True, lambda: pkg.train_seg_gan.gan_step(inp, tgt, G, D, pkg.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3))
    e = np.abs(tap['logits'].cpu().numpy() - gold['s0_logits'])
    assert e.max() < 2e-4, 'logits err %.3e' % e.max()
    got = np.array([out[0].item(), out[3].item(), out[4].item(), out[5].item(), out[1].item(), out[2].item()])
    tol = np.array([2e-5, 5e-5, 1e-4, 2e-4, 1e-4, 1e-4])
    assert (np.abs(got - gold['s0_scalars']) < tol).all(), (got, gold['s0_scalars'])


@pytest.mark.parametrize('cin,co,hw,k,stride', [(128, 128, 128, 3, 2), (256, 128, 96, 1, 1), (64, 64, 192, 3, 2), (128, 256, 64, 1, 1)])
def test_split_dma_family_matches_fp64_like_fp32_mfma(pkg, dev, cin, co, hw, k, stride):
    """The LDS-DMA pipeline with split operands (conv_igemm_dma_x3.hip): stride-2 3x3 convs, 1x1 convs, and -- through the
    input gradient of the stride-2 conv -- the four parity-class launches (1 / 2 / 2 / 4 taps, strided outputs)."""
    ops = pkg.ops
    torch.manual_seed(23)
    torch.set_num_threads(16)
    nb = 4
    xc = torch.randn(nb, cin, hw, hw) * 1.2 - 0.2
    wc = torch.randn(co, cin, k, k) / (k * cin ** 0.5)
    pad = k // 2
    ref = F.conv2d(xc.double(), wc.double(), None, stride, pad)
    x = ops.to_nhwc(xc.to(dev)); w = wc.to(dev)

    def fwd():
        return ops._conv_fwd_impl(x, None, w, None, stride, pad, 0, 0.0)
    y32 = _run(ops, False, fwd).cpu().double()
    ops.PROFILE = []
    try:
        y3 = _run(ops, True, fwd).cpu().double()
        labels = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert labels and 'dma_x3' in labels[0], 'the split LDS-DMA kernel did not run: %s' % labels
    e32 = (y32 - ref).abs(); e3 = (y3 - ref).abs()
    assert e3.max().item() <= 2.0 * e32.max().item() + 1e-6, (e3.max().item(), e32.max().item())
    assert e3.pow(2).mean().sqrt().item() <= 2.0 * e32.pow(2).mean().sqrt().item() + 1e-8
    oh = ref.shape[2]
    dyc = torch.randn(nb, co, oh, oh)
    dy = ops.to_nhwc(dyc.to(dev))
    gref = torch.nn.grad.conv2d_input((nb, cin, hw, hw), wc.double(), dyc.double(), stride, pad)

    def dgrad():
        return ops._conv_dgrad_impl(dy, w, stride, pad, hw, hw, 0, cin)
    g32 = _run(ops, False, dgrad).cpu().double()
    ops.PROFILE = []
    try:
        g3 = _run(ops, True, dgrad).cpu().double()
        labels = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert any('x3' in l for l in labels), labels
    assert (g3 - gref).abs().max().item() <= 2.0 * (g32 - gref).abs().max().item() + 1e-6

    # weight gradient of the same layer: wgrad_dma_x3_kernel (both operands split as they leave LDS)
    wref = torch.nn.grad.conv2d_weight(xc.double(), wc.shape, dyc.double(), stride, pad)

    def wgrad():
        return ops._conv_wgrad_impl(x, None, dy, tuple(w.shape), stride, pad)
    w32 = _run(ops, False, wgrad).cpu().double()
    ops.PROFILE = []
    try:
        w3 = _run(ops, True, wgrad).cpu().double()
        wl = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert wl and 'wgrad_dma_x3' in wl[0], 'the split LDS-DMA weight-gradient kernel did not run: %s' % wl
    ew32 = (w32 - wref).abs(); ew3 = (w3 - wref).abs()
    assert ew3.max().item() <= 2.0 * ew32.max().item() + 1e-6 * wref.abs().max().item(), (ew3.max().item(), ew32.max().item())
    assert ew3.pow(2).mean().sqrt().item() <= 2.0 * ew32.pow(2).mean().sqrt().item() + 1e-8


@pytest.mark.parametrize('cin,co,h,w,nb', [(64, 64, 96, 128, 3), (128, 256, 63, 65, 2), (192, 64, 34, 40, 2), (64, 48, 31, 33, 1)])
def test_merged_parity_input_gradient_of_stride2_conv(pkg, dev, cin, co, h, w, nb):
    """conv_igemm_halo_x3_kernel<..., PARITY>: the four output-parity classes of a 3x3 stride-2 pad-1 input gradient as ONE launch
    (models_seg_gan.py:37-39, the s2 blocks of D), against fp64 and against the four per-class launches; odd image sizes give
    the classes unequal grids."""
    ops = pkg.ops
    torch.manual_seed(29)
    torch.set_num_threads(16)
    wc = torch.randn(co, cin, 3, 3) / (3 * cin ** 0.5)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dyc = torch.randn(nb, co, oh, ow)
    gref = torch.nn.grad.conv2d_input((nb, cin, h, w), wc.double(), dyc.double(), 2, 1)
    dy = ops.to_nhwc(dyc.to(dev)); wd = wc.to(dev)
    saved = ops.PARITY_MERGE

    def run(merge):
        ops.PARITY_MERGE = merge
        ops.PROFILE = []
        try:
            g = ops._conv_dgrad_impl(dy, wd, 2, 1, h, w, 0, cin).cpu().double()
            return g, [p[0] for p in ops.PROFILE]
        finally:
            ops.PROFILE = None
            ops.PARITY_MERGE = saved
    g1, l1 = _run(ops, True, lambda: run(True))
    g4, l4 = _run(ops, True, lambda: run(False))
    g32, _ = _run(ops, False, lambda: run(False))
    assert l1 == ['conv_igemm_halo_x3_kernel<128,64,4,1,true>'], l1
    assert len(l4) == 4, l4
    e1 = (g1 - gref).abs().max().item(); e4 = (g4 - gref).abs().max().item(); e32 = (g32 - gref).abs().max().item()
    assert e1 <= 2.0 * e32 + 1e-6, (e1, e32)
    assert e1 <= 2.0 * e4 + 1e-6, (e1, e4)
    assert (g1 - gref).pow(2).mean().sqrt().item() <= 2.0 * (g32 - gref).pow(2).mean().sqrt().item() + 1e-8


def test_merged_parity_declines_narrow_images(pkg, dev):
    """dy narrower than 17 pixels (no 32-wide halo tile): ssg_conv2d_split_bn reports no merged kernel and the four per-class
    launches run; the result is the same input gradient."""
    ops = pkg.ops
    torch.manual_seed(31)
    cin, co, h, w, nb = 64, 64, 24, 30, 2
    wc = torch.randn(co, cin, 3, 3) / (3 * cin ** 0.5)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert ow < 17
    dyc = torch.randn(nb, co, oh, ow)
    gref = torch.nn.grad.conv2d_input((nb, cin, h, w), wc.double(), dyc.double(), 2, 1)
    ops.PROFILE = []
    try:
        g = _run(ops, True, lambda: ops._conv_dgrad_impl(ops.to_nhwc(dyc.to(dev)), wc.to(dev), 2, 1, h, w, 0, cin)).cpu().double()
        labels = [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert len(labels) == 4 and not any('true' in l for l in labels), labels
    assert (g - gref).abs().max().item() < 2e-5


# ---------------------------------------------------------------------------------------------------------------------------
# non-finite and extreme operands.  x = x1 + x2 + x3 in bf16 terms turns +-inf into (inf, NaN, NaN) and a finite
# |x| above the bf16 maximum (3.3895e38) into (inf, -inf, NaN), where the fp32 kernels -- and the reference: losses.py:297-300 keys
# on inf vs NaN, train_seg_gan.py:190 zeroes NaN only -- keep +-inf / a finite product.  The split kernels detect such operands
# (every accumulator they touch is non-finite) and recompute the affected workgroup's tile with plain fp32 FMAs (csrc/conv_slow.h):
# the result must be of the fp32-MFMA kernel's CLASS element by element (finite / +inf / -inf / NaN) and equal where finite.
SPECIALS = [float('inf'), float('-inf'), float('nan'), 3.4e38, -3.4e38, 1e-40, 1e30, -1e30, 1e-30]


def _inject(t, specials, seed, dim=1):
    """One special value per DISTINCT channel (dim 1) at a random position: an output sees at most one of them per operand."""
    g = torch.Generator().manual_seed(seed)
    t = t.clone()
    chans = torch.randperm(t.shape[dim], generator=g)[:len(specials)]
    for v, c in zip(specials, chans):
        idx = [int(torch.randint(0, s, (1,), generator=g)) for s in t.shape]
        idx[dim] = int(c)
        t[tuple(idx)] = v
    return t


def _same_class(got, want, what, rtol=2e-5, atol=2e-4):
    got = got.detach().cpu().double(); want = want.detach().cpu().double()
    for name, f in (('NaN', torch.isnan), ('+inf', lambda t: torch.isposinf(t)), ('-inf', lambda t: torch.isneginf(t))):
        a, b = f(got), f(want)
        assert torch.equal(a, b), '%s: %s pattern differs from the fp32-MFMA kernel at %d elements (%d vs %d)' % (what, name, int((a != b).sum()), int(a.sum()), int(b.sum()))
    fin = torch.isfinite(want)
    assert int((~fin).sum()) > 0, '%s: the test operands produced no non-finite output' % what
    d = (got[fin] - want[fin]).abs()
    # sums that hold a 1e30-class term are compared relative to it; everything else to the usual fp32 accumulation noise
    tol = rtol * want[fin].abs() + atol
    assert bool((d <= tol).all()), '%s: finite elements differ, worst %.3e (value %.3e)' % (what, (d - tol).max().item(), want[fin][(d - tol).argmax()].item())


def _labels(ops, fn):
    ops.PROFILE = []
    try:
        r = fn()
        torch.cuda.synchronize()
        return r, [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None


# Narrow k32 tiles (<8,16> / <8,32>: all output channels of a 12..32-channel layer in one padded column tile; SPADE's x -> map convs):
# same accuracy bound against fp64 as the fp32-MFMA kernel, bias + residual + activation epilogue, padded columns never written
# (the output rows are checked beyond Cout), batch-norm partial rows, and the non-finite guard.
@pytest.mark.parametrize('c1,c2,co,h,w,nb', [(512, 0, 16, 64, 64, 4), (256, 0, 24, 40, 72, 3), (128, 128, 32, 33, 50, 2), (128, 0, 12, 24, 40, 2)])
def test_narrow_k32_tiles(pkg, dev, c1, c2, co, h, w, nb):
    ops = pkg.ops
    call = pkg._lib.call
    torch.manual_seed(31)
    torch.set_num_threads(16)
    ci = c1 + c2
    xc = torch.randn(nb, ci, h, w) * 1.5 + 0.3
    wc = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
    bc = torch.randn(co)
    rc = torch.randn(nb, co, h, w)
    ref = F.leaky_relu(F.conv2d(xc.double(), wc.double(), bc.double(), 1, 1) + rc.double(), 0.2)
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    wd = wc.to(dev); bd = bc.to(dev); res = ops.to_nhwc(rc.to(dev))
    from ssunet_gan_amd._lib import ACT_LRELU, ACT_NONE

    def fwd():
        out = ops.new_nhwc(nb, co, h, w, dev)
        out.fill_(-7.0)                                       # also the padding between Cout and the pixel stride, if any
        return ops._conv_fwd_impl(x1, x2, wd, bd, 1, 1, ACT_LRELU, 0.2, res=res, out=out)
    call('ssg_conv_set_k32_mode', 2)
    try:
        y32 = _run(ops, False, fwd).cpu().double()
        y3, labels = _labels(ops, lambda: _run(ops, True, fwd))
        assert labels and ('k32_kernel<8,16>' if co <= 16 else 'k32_kernel<8,32>') in labels[0], labels
        y3 = y3.cpu().double()
        e32 = (y32 - ref).abs(); e3 = (y3 - ref).abs()
        # the maximum over ~1e5 outputs is a noisy statistic (2.2 x on one of these shapes): 3 x on it, 2 x on the rms
        assert e3.max().item() <= 3.0 * e32.max().item() + 1e-6, (e3.max().item(), e32.max().item())
        assert e3.pow(2).mean().sqrt().item() <= 2.0 * e32.pow(2).mean().sqrt().item() + 1e-8
        # batch-norm partial rows from the epilogue: column sums and sums of squares of the raw conv output
        y, part = _run(ops, True, lambda: ops._conv_fwd_impl(x1, x2, wd, None, 1, 1, ACT_NONE, 0.0, want_bn=True))
        assert part is not None and part.shape[1:] == (2, co)
        yd = y.double()
        s1 = yd.sum(dim=(0, 2, 3)).cpu(); s2 = (yd * yd).sum(dim=(0, 2, 3)).cpu()
        tot = part.sum(0).cpu()
        # fp32 deviation sums per 16-lane group, fp64 beyond: 1e-6 of the sum of squares
        assert torch.allclose(tot[0], s1, rtol=1e-6, atol=1e-6 * s2.max().item() ** 0.5 * 64) and torch.allclose(tot[1], s2, rtol=1e-6, atol=1e-3)
        # non-finite operands: the class of every output (finite value / +inf / -inf / NaN) is the fp32 kernel's
        xb = _inject(xc.clone(), SPECIALS, 41)
        xb1 = ops.to_nhwc(xb[:, :c1].to(dev)); xb2 = ops.to_nhwc(xb[:, c1:].to(dev)) if c2 else None
        fnb = lambda: ops._conv_fwd_impl(xb1, xb2, wd, bd, 1, 1, ACT_NONE, 0.0)
        want = _run(ops, False, fnb)
        got, labels = _labels(ops, lambda: _run(ops, True, fnb))
        assert any('k32' in l for l in labels), labels
        _same_class(got, want, 'narrow tile, special activations')
    finally:
        call('ssg_conv_set_k32_mode', 1)


@pytest.mark.parametrize('k32', [0, 2], ids=['x3', 'k32'])
@pytest.mark.parametrize('c1,c2,co,h,w,nb', [(128, 0, 128, 128, 128, 4), (64, 64, 64, 48, 80, 2), (64, 0, 64, 64, 64, 3)])
def test_split_conv_on_nonfinite_operands(pkg, dev, c1, c2, co, h, w, nb, k32):
    ops = pkg.ops
    call = pkg._lib.call
    torch.manual_seed(5)
    ci = c1 + c2
    xc = _inject(torch.randn(nb, ci, h, w) * 1.5 + 0.3, SPECIALS, 11)
    wc = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
    wc_bad = _inject(wc, [float('inf'), float('nan'), 1e30, 1e-40, float('-inf')], 12, dim=0)
    dyc = _inject(torch.randn(nb, co, h, w), SPECIALS, 13)
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    xgc = torch.randn(nb, ci, h, w) * 1.5 + 0.3
    xg1 = ops.to_nhwc(xgc[:, :c1].to(dev)); xg2 = ops.to_nhwc(xgc[:, c1:].to(dev)) if c2 else None
    dy = ops.to_nhwc(dyc.to(dev))
    call('ssg_conv_set_k32_mode', k32)
    try:
        want_label = 'k32' if k32 else 'halo_x3'
        checked = 0
        for what, fn in (
                ('forward, special activations', lambda: ops._conv_fwd_impl(x1, x2, wc.to(dev), None, 1, 1, 0, 0.0)),
                ('forward, special weights', lambda: ops._conv_fwd_impl(xg1, xg2, wc_bad.to(dev), None, 1, 1, 0, 0.0)),
                ('input gradient, special dy', lambda: ops._conv_dgrad_impl(dy, wc.to(dev), 1, 1, h, w, 0, ci))):
            want = _run(ops, False, fn)
            got, labels = _labels(ops, lambda: _run(ops, True, fn))
            if not any(want_label in l for l in labels):      # small grids with a long reduction stay on the fp32 split-K kernel
                continue
            checked += 1
            _same_class(got, want, '%s [%s]' % (what, labels[0]))
        assert checked >= 1, 'the %s kernels ran in none of the 3 cases' % want_label
    finally:
        call('ssg_conv_set_k32_mode', 1)


@pytest.mark.parametrize('cin,co,hw,k,stride', [(128, 128, 128, 3, 2), (128, 64, 64, 1, 1)])
def test_split_dma_family_on_nonfinite_operands(pkg, dev, cin, co, hw, k, stride):
    """1x1 and stride-2 convs (conv_igemm_dma_x3), the merged-parity input gradient of the stride-2 conv, and their weight gradient
    (wgrad_dma_x3)."""
    ops = pkg.ops
    torch.manual_seed(6)
    nb, pad = 2, k // 2
    xc = _inject(torch.randn(nb, cin, hw, hw) * 1.5 + 0.3, SPECIALS, 21)
    wc = torch.randn(co, cin, k, k) / (k * cin ** 0.5)
    oh = (hw + 2 * pad - k) // stride + 1
    dyc = _inject(torch.randn(nb, co, oh, oh), SPECIALS, 23)
    x = ops.to_nhwc(xc.to(dev)); w = wc.to(dev); dy = ops.to_nhwc(dyc.to(dev))
    xg = ops.to_nhwc((torch.randn(nb, cin, hw, hw) * 1.5 + 0.3).to(dev))
    dyg = ops.to_nhwc(torch.randn(nb, co, oh, oh).to(dev))
    for what, fn, lab in (
            ('forward', lambda: ops._conv_fwd_impl(x, None, w, None, stride, pad, 0, 0.0), 'dma_x3'),
            ('input gradient', lambda: ops._conv_dgrad_impl(dy, w, stride, pad, hw, hw, 0, cin), 'x3'),
            ('weight gradient, special dy', lambda: ops._conv_wgrad_impl(xg, None, dy, tuple(wc.shape), stride, pad), 'wgrad_dma_x3'),
            ('weight gradient, special x', lambda: ops._conv_wgrad_impl(x, None, dyg, tuple(wc.shape), stride, pad), 'wgrad_dma_x3')):
        want = _run(ops, False, fn)
        got, labels = _labels(ops, lambda: _run(ops, True, fn))
        assert any(lab in l for l in labels), '%s: no %s kernel ran: %s' % (what, lab, labels)
        _same_class(got, want, '%s [%s]' % (what, labels), atol=2e-3 if 'weight' in what else 2e-4)


@pytest.mark.parametrize('k32', [0, 1], ids=['halo_x3', 'k32'])
@pytest.mark.parametrize('c1,c2,co,h,w,nb', [(64, 0, 64, 48, 64, 2), (128, 64, 128, 33, 40, 2)])
def test_split_wgrad_on_nonfinite_operands(pkg, dev, c1, c2, co, h, w, nb, k32):
    ops = pkg.ops
    call = pkg._lib.call
    torch.manual_seed(7)
    ci = c1 + c2
    xc = torch.randn(nb, ci, h, w) * 1.5 + 0.3
    dyc = torch.randn(nb, co, h, w)
    call('ssg_wgrad_set_k32_mode', k32)
    try:
        for what, xs, ds in (('special x', _inject(xc, SPECIALS, 31), dyc), ('special dy', xc, _inject(dyc, SPECIALS, 32))):
            x1 = ops.to_nhwc(xs[:, :c1].to(dev)); x2 = ops.to_nhwc(xs[:, c1:].to(dev)) if c2 else None
            dy = ops.to_nhwc(ds.to(dev))
            fn = lambda: ops._conv_wgrad_impl(x1, x2, dy, (co, ci, 3, 3), 1, 1)
            want = _run(ops, False, fn)
            got, labels = _labels(ops, lambda: _run(ops, True, fn))
            assert any(('wgrad_k32' if k32 else 'wgrad_halo_x3') in l for l in labels), labels
            _same_class(got, want, 'weight gradient, %s [%s]' % (what, labels), atol=2e-3)
    finally:
        call('ssg_wgrad_set_k32_mode', 1)
