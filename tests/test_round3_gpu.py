"""Round-3 additions: numerics of the conv-epilogue batch-norm statistics under a large mean/std ratio,
split-K launches that keep the statistics epilogue, and the bench sub-metric plumbing."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('k,min_ratio', [(3, 8.0), (1, 30.0)])
def test_bn_epilogue_statistics_with_large_mean(pkg, dev, k, min_ratio):
    """var = E[x^2] - mean^2 cancels; per-lane fp32 partials (relative error ~1e-7 * mean^2 / var in the variance) stop
    matching the stand-alone fp64 statistics pass once |mean| >> std.  The halo (3x3; the zero-padded border rows bound its
    mean/std ratio at ~10) and the LDS-DMA (1x1: ratio > 30) epilogues accumulate per lane in fp64 and must agree with it."""
    ops = pkg.ops
    torch.manual_seed(11)
    n, c, hw = 8, 128, 64
    x = ops.to_nhwc((torch.randn(n, c, hw, hw) * 0.02 + 1.0).to(dev))
    w = (torch.full((c, c, k, k), 1.0 / (k * k * c)) + torch.randn(c, c, k, k) * 2e-5).to(dev)
    y, part = ops._conv_fwd_impl(x, None, w, None, 1, k // 2, 0, 0.0, want_bn=True)
    assert part is not None and part.numel() > 0, 'this shape must take the statistics epilogue'
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    _, st_epi, _ = ops._bn_fwd_impl(y, g, b, None, None, None, 1e-5, 0.1, 0, 0.0, 0, None, part=part)
    _, st_own, _ = ops._bn_fwd_impl(y, g, b, None, None, None, 1e-5, 0.1, 0, 0.0, 0, None, part=None)
    yd = y.double()
    mean = yd.mean((0, 2, 3)); var = yd.var((0, 2, 3), unbiased=False)
    ratio = (mean.abs() / var.sqrt()).min().item()
    assert ratio > min_ratio, ratio
    inv = (var + 1e-5).rsqrt()
    for name, st in (('epilogue', st_epi), ('own pass', st_own)):
        assert (st[0].double() - mean).abs().max().item() < 2e-7 * mean.abs().max().item(), name
        rel = ((st[1].double() - inv) / inv).abs().max().item()
        assert rel < 5e-6, '%s: invstd relative error %.3e at mean/std %.0f' % (name, rel, ratio)
    assert ((st_epi[1] - st_own[1]) / st_own[1]).abs().max().item() < 2e-6


@pytest.mark.parametrize('n,c,s', [(4, 144, 6), (4, 2688, 112), (16, 96, 4), (3, 40, 10), (16, 1152, 48)])
def test_se_gate_matches_the_two_convs(pkg, dev, n, c, s):
    """csrc/se_gate.hip: sigmoid(_se_expand(swish(_se_reduce(sq)))) forward and all five gradients against plain torch
    (efficientnet_pytorch/model.py:84-86; utils.py:37-48 for the swish derivative)."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(5 + c)
    red = nn.Conv2d(c, s, 1); exp = nn.Conv2d(s, c, 1)
    sq = torch.randn(n, c, 1, 1) * 0.7
    g = torch.randn(n, c, 1, 1)
    sqr = sq.double().requires_grad_(True)
    w1 = red.weight.double().detach().requires_grad_(True); b1 = red.bias.double().detach().requires_grad_(True)
    w2 = exp.weight.double().detach().requires_grad_(True); b2 = exp.bias.double().detach().requires_grad_(True)
    h = F.conv2d(sqr, w1, b1)
    ref = torch.sigmoid(F.conv2d(h * torch.sigmoid(h), w2, b2))
    ref.backward(g.double())
    red = red.to(dev); exp = exp.to(dev)
    sqd = sq.to(dev).requires_grad_(True)
    out = pkg.ops.se_gate(sqd, red, exp)
    assert out is not None, 'shape inside the fused range'
    out.backward(g.to(dev))

    def close(a, b, what):
        a = a.detach().cpu().double().reshape(b.shape); err = (a - b).abs().max().item(); ref_ = b.abs().max().item()
        assert err <= 2e-6 * max(ref_, 1.0) + 1e-5 * ref_, '%s: %.3e vs max %.3e' % (what, err, ref_)
    close(out, ref.detach(), 'gate')
    close(sqd.grad, sqr.grad, 'dsq')
    close(red.weight.grad, w1.grad, 'dw1'); close(red.bias.grad, b1.grad, 'db1')
    close(exp.weight.grad, w2.grad, 'dw2'); close(exp.bias.grad, b2.grad, 'db2')
    # outside the range the caller keeps the conv path
    assert pkg.ops.se_gate(torch.randn(17, c, 1, 1, device=dev), red, exp) is None


def test_fused_statistics_finalize_gives_the_bits_of_the_two_launch_route(pkg, dev):
    """ssg_bn_stats_finalize_f32 / ssg_bn_stats_from_partials_finalize_f32 (local batch norm: the second reduce stage finishes the
    channel) against ssg_bn_stats_* + ssg_bn_finalize_f32: outputs, constants and running estimates bit for bit."""
    ops = pkg.ops
    torch.manual_seed(0)
    saved = ops.BN_FUSED_FINALIZE
    try:
        for (n, c, hw, var_mode) in [(6, 64, 32, 1), (2, 128, 64, 0), (3, 40, 17, 0)]:
            x = ops.to_nhwc((torch.randn(n, c, hw, hw) * 0.7 + 0.3).to(dev))
            w = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev)
            outs = []
            for fused in (True, False):
                ops.BN_FUSED_FINALIZE = fused
                rm = torch.randn(c, device=dev, generator=torch.Generator(device=dev).manual_seed(3)); rv = torch.rand(c, device=dev, generator=torch.Generator(device=dev).manual_seed(4)) + 0.5
                y, stats, cnt = ops._bn_fwd_body(x, w, b, rm, rv, None, 1e-5, 0.1, 0, 0.0, var_mode, None)
                assert cnt is None
                outs.append((y.clone(), stats.clone(), rm.clone(), rv.clone()))
            for a, bb, name in zip(outs[0], outs[1], ('y', 'stats', 'running_mean', 'running_var')):
                assert torch.equal(a, bb), '%s differs between the fused and the two-launch route (C=%d)' % (name, c)
    finally:
        ops.BN_FUSED_FINALIZE = saved
