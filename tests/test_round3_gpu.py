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
