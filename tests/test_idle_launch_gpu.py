"""Regression test for the LDS-DMA ring write-after-read race found in round 3.

The ring kernels (conv_igemm_halo / conv_igemm_dma / wgrad_halo / wgrad_dma / gemm_bf16) hand LDS stage (s+2) % 3 to the next
DMA right after the workgroup barrier of step s.  s_barrier does not wait for lgkmcnt, so a wave could reach it with its last
fragment reads of step s-1 (the same stage) still queued; launched on an idle chip, roughly one launch in ten of the round-2
kernels came back with a wrong 16-channel x 64-pixel patch (back-to-back launches never did, which is why the parity suite
and the bench never saw it).  Fixed by draining the wave's LDS reads before the barrier (csrc/lds_dma.h: wait_lds_reads).

Each case launches the kernel repeatedly with an idle gap before every launch into a NaN-prefilled output and requires all
results to be bitwise identical; the first result is also checked against torch's CPU convolution on a few images."""
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LAUNCHES, GAP = 16, 0.12


def _idle():
    torch.cuda.synchronize()
    time.sleep(GAP)


def _same(results, what):
    bad = [i for i, r in enumerate(results[1:], 1) if not torch.equal(r, results[0])]
    assert not bad, '%s: launches %s differ from launch 0 (max |d| %.3e)' % (
        what, bad, max((results[i] - results[0]).abs().max().item() for i in bad))


@pytest.mark.parametrize('split', [True, False], ids=['split', 'fp32mfma'])
@pytest.mark.parametrize('c1,c2,co,hw,stride', [(128, 0, 128, 128, 1), (64, 128, 128, 128, 1), (128, 0, 128, 128, 2), (64, 0, 64, 256, 1)])
def test_conv_fwd_dgrad_wgrad_after_idle(c1, c2, co, hw, stride, split, monkeypatch):
    import ssunet_gan_amd as S
    from ssunet_gan_amd import ops
    if split and stride != 1:
        pytest.skip('stride 2 has no split-operand kernel')
    monkeypatch.setattr(ops, 'MFMA_SPLIT', split)      # restored on every exit path
    from ssunet_gan_amd._lib import ACT_NONE
    dev = torch.device('cuda')
    torch.manual_seed(3)
    nb = 8
    xc = torch.randn(nb, c1 + c2, hw, hw)
    x1 = ops.to_nhwc(xc[:, :c1].to(dev)); x2 = ops.to_nhwc(xc[:, c1:].to(dev)) if c2 else None
    w = (torch.randn(co, c1 + c2, 3, 3) / (3 * (c1 + c2) ** 0.5)).to(dev)
    oh = (hw + 2 - 3) // stride + 1
    dy = ops.to_nhwc(torch.randn(nb, co, oh, oh).to(dev))
    outs, dxs, dws = [], [], []
    for _ in range(LAUNCHES):
        out = ops.new_nhwc(nb, co, oh, oh, dev); out.fill_(float('nan'))
        _idle()
        outs.append(ops._conv_fwd_impl(x1, x2, w, None, stride, 1, ACT_NONE, 0.0, out=out))
        _idle()
        dxs.append(ops._conv_dgrad_impl(dy, w, stride, 1, hw, hw, 0, c1))
        _idle()
        dws.append(ops._conv_wgrad_impl(x1, x2, dy, tuple(w.shape), stride, 1))
    torch.cuda.synchronize()
    _same(outs, 'conv forward'); _same(dxs, 'input gradient'); _same(dws, 'weight gradient')
    ref = F.conv2d(xc[:2], w.cpu(), None, stride, 1)
    err = (outs[0][:2].cpu() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


def test_gemm_bf16_after_idle():
    import ssunet_gan_amd as S
    from ssunet_gan_amd import bf16
    dev = torch.device('cuda')
    torch.manual_seed(4)
    x = torch.randn(4, 96, 128, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(192, 96, 1, 1, device=dev) / 10).requires_grad_()
    dy = torch.randn(4, 192, 128, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ys, dxs, dws = [], [], []
    for _ in range(LAUNCHES):
        xi = x.clone().requires_grad_()
        _idle()
        y = bf16.conv1x1(xi, w)
        _idle()
        dx, dw = torch.autograd.grad(y, (xi, w), dy)
        ys.append(y.detach().float()); dxs.append(dx.float()); dws.append(dw)
    torch.cuda.synchronize()
    _same(ys, 'bf16 GEMM'); _same(dxs, 'bf16 input gradient'); _same(dws, 'bf16 weight gradient')
