"""bf16 path (BASELINE config 4): the pointwise-conv GEMMs on v_mfma_f32_32x32x16_bf16 against fp32 torch on the SAME
bf16-rounded operands (so the only differences are fp32 accumulation order and the final bf16 rounding of the output)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pack(pkg, w, transpose, dev):
    from ssunet_gan_amd._lib import call, ptr, stream_ptr
    o, i = w.shape
    rows, cols = (i, o) if transpose else (o, i)
    rows_pad = (rows + 127) // 128 * 128
    kp = (cols + 31) // 32 * 32
    out = torch.empty((rows_pad, kp), dtype=torch.bfloat16, device=dev)
    call('ssg_pack_weights_bf16', ptr(w), o, i, int(transpose), rows_pad, kp, ptr(out), stream_ptr())
    return out, kp


@pytest.mark.parametrize('P,K,N', [(1000, 24, 144), (4096 + 37, 144, 24), (300, 672, 112), (513, 2688, 448), (130, 8, 8), (256, 48, 1792)])
def test_gemm_bf16_fwd(pkg, dev, P, K, N):
    from ssunet_gan_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(P + K + N)
    x = torch.randn(P, K, generator=g).bfloat16()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    res = torch.randn(P, N, generator=g).bfloat16()
    xd, wd, rd = x.to(dev), w.to(dev), res.to(dev)
    wp, kp = _pack(pkg, wd, 0, dev)
    assert torch.equal(wp[:N, :K].cpu(), w.bfloat16()) and wp[N:].abs().sum().item() == 0 and wp[:, K:].abs().sum().item() == 0
    ref = x.float() @ w.bfloat16().float().t()
    for use_res in (False, True):
        out = torch.full((P, N), float('nan'), dtype=torch.bfloat16, device=dev)
        call('ssg_gemm_bf16', ptr(xd), P, K, K, ptr(wp), kp, N, ptr(rd) if use_res else None, N, ptr(out), N, stream_ptr())
        torch.cuda.synchronize()
        r = ref + (res.float() if use_res else 0)
        err = (out.float().cpu() - r).abs()
        tol = 2 ** -8 * r.abs() + 1e-3          # one bf16 rounding of the result (8 significant bits) + accumulation noise
        assert not torch.isnan(out).any() and (err <= tol).all(), 'max err %.4f at |ref| %.3f' % (err.max().item(), r.abs().max().item())


@pytest.mark.parametrize('P,M,N', [(1000, 24, 144), (4096 + 37, 144, 24), (20000, 112, 672), (700, 448, 2688), (64, 8, 8), (33, 1792, 48)])
def test_gemm_bf16_wgrad(pkg, dev, P, M, N):
    from ssunet_gan_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(P + M + N)
    dy = torch.randn(P, M, generator=g).bfloat16()
    x = torch.randn(P, N, generator=g).bfloat16()
    # A = I check with an asymmetric partner is implied by random data; add a structured case on the first shape
    ref = dy.float().t().double() @ x.float().double()
    dyd, xd = dy.to(dev), x.to(dev)
    nbytes = call('ssg_gemm_wgrad_bf16_workspace_bytes', P, M, N)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=dev)
    dw = torch.full((M, N), float('nan'), dtype=torch.float32, device=dev)
    call('ssg_gemm_wgrad_bf16', ptr(dyd), M, ptr(xd), N, P, M, N, ptr(dw), ptr(ws), nbytes, stream_ptr())
    torch.cuda.synchronize()
    err = (dw.double().cpu() - ref).abs()
    scale = (dy.float().abs().t().double() @ x.float().abs().double())
    assert not torch.isnan(dw).any() and (err <= 2e-6 * scale + 1e-5).all(), 'max err %.3e' % err.max().item()
    dw2 = torch.empty_like(dw)
    call('ssg_gemm_wgrad_bf16', ptr(dyd), M, ptr(xd), N, P, M, N, ptr(dw2), ptr(ws), nbytes, stream_ptr())
    assert torch.equal(dw, dw2), 'weight gradient is not run-to-run reproducible'


# ----------------------------------------------------------------------------- the MBConv path in bf16
import os

from conftest import GOLDEN

# bf16 keeps 8 significant bits: one rounding is 2^-9 relative (3.9e-3 worst case).  Through the ~50 rounded tensors of a
# B0 forward (or ~100 of B4) errors add like a random walk: the stated tolerance for features and input gradients is
# 6e-2 of the tensor's max for the worst element and 1.5e-2 of its RMS for the median element; parameter-gradient norms
# (fp32 sums over >= 1e4 bf16 products) 5e-2 relative for the median tensor.  fp32 runs of the same code meet 2e-5 / 1e-3.
BF16_MAX_REL, BF16_MED_REL, BF16_GRAD_MED = 6e-2, 1.5e-2, 5e-2


def _digest(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


def test_efficientnet_b0_bf16_vs_reference_fp32(pkg, dev):
    gold = np.load(os.path.join(GOLDEN, 'unwired.npz'))
    E = pkg.efficientnet_pytorch
    torch.manual_seed(36)
    net = E.EfficientNet.from_name('efficientnet-b0', override_params=dict(drop_connect_rate=0.0))
    net.to(dev).train().set_compute_dtype(torch.bfloat16)
    x = torch.from_numpy(gold['eff_x']).to(dev).requires_grad_(True)
    f = net.extract_features(x)
    assert f.dtype == torch.float32 and tuple(f.shape) == tuple(gold['eff_feat'].shape)
    ref = gold['eff_feat']
    e = np.abs(f.detach().cpu().numpy() - ref)
    rms = np.sqrt((ref ** 2).mean())
    print('B0 bf16 features: max err %.3e (max |ref| %.3e)  median err %.3e (rms %.3e)' % (e.max(), np.abs(ref).max(), np.median(e), rms))
    assert e.max() < BF16_MAX_REL * np.abs(ref).max() and np.median(e) < BF16_MED_REL * rms
    f.backward(torch.from_numpy(gold['eff_dy']).to(dev))
    dxr = gold['eff_dx']
    dxe = np.abs(x.grad.cpu().numpy() - dxr)
    print('B0 bf16 dx: max err %.3e (max |ref| %.3e) median %.3e' % (dxe.max(), np.abs(dxr).max(), np.median(dxe)))
    assert dxe.max() < 2 * BF16_MAX_REL * np.abs(dxr).max() and np.median(dxe) < 2 * BF16_MED_REL * np.sqrt((dxr ** 2).mean())
    params = [p for n, p in net.named_parameters() if not n.startswith('_fc')]
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in params)
    gd = np.stack([_digest(p.grad) for p in params])
    big = gold['eff_gd'][:, 2] > 2e-3
    rel = np.abs(gd[:, 2] - gold['eff_gd'][:, 2]) / (gold['eff_gd'][:, 2] + 1e-12)
    print('B0 bf16 parameter-gradient norms: median rel err %.3e, worst (of the non-degenerate) %.3e' % (np.median(rel[big]), rel[big].max()))
    assert np.median(rel[big]) < BF16_GRAD_MED and rel[big].max() < 0.5
    # running statistics were updated from bf16-rounded activations: close to the fp32 run's
    net.eval()
    with torch.no_grad():
        fe = net.extract_features(x.detach())
    ee = np.abs(fe.cpu().numpy() - gold['eff_feat_eval'])
    assert ee.max() < BF16_MAX_REL * np.abs(gold['eff_feat_eval']).max() + 1e-3


def test_efficientnet_b4_1024_bf16_config4(pkg, dev):
    """BASELINE config 4 at full size: B4 extract_features fwd+bwd on 4 x 3 x 1024 x 1024 in bf16 against the reference's fp32
    run (fixture: oracle/gen_golden.py --only effb4), and the fp32 path of the same code against the same fixture."""
    gold = np.load(os.path.join(GOLDEN, 'effb4_n4_1024.npz'))
    E = pkg.efficientnet_pytorch
    n, _, h, w = [int(v) for v in gold['shape']]
    x0 = torch.randn(n, 3, h, w, generator=torch.Generator().manual_seed(int(gold['seed_x'])))
    dy = torch.randn(*[int(v) for v in gold['feat_shape']], generator=torch.Generator().manual_seed(int(gold['seed_dy'])))
    for dtype, fmax, fmed, gmed in ((torch.float32, 2e-3, 1e-4, 2e-3), (torch.bfloat16, BF16_MAX_REL, BF16_MED_REL, BF16_GRAD_MED)):
        torch.manual_seed(int(gold['seed_model']))
        net = E.EfficientNet.from_name('efficientnet-b4', override_params=dict(drop_connect_rate=0.0))
        init = np.stack([_digest(p) for p in net.parameters()])
        assert np.allclose(init, gold['init'], rtol=1e-9, atol=1e-12), 'B4 init differs from the reference'
        net.to(dev).train().set_compute_dtype(dtype)
        x = x0.to(dev).requires_grad_(True)
        f = net.extract_features(x)
        fd = _digest(f)
        sub = f.detach()[:, ::16, ::4, ::4].cpu().numpy()
        ref = gold['feat_sub']
        e = np.abs(sub - ref)
        rms = np.sqrt((ref ** 2).mean())
        print('B4 %s features: max err %.3e (max |ref| %.3e) median %.3e (rms %.3e); l2 %.6e vs %.6e' %
              (dtype, e.max(), np.abs(ref).max(), np.median(e), rms, fd[2], gold['feat_digest'][2]))
        assert np.isfinite(fd).all()
        assert e.max() < fmax * np.abs(ref).max() and np.median(e) < fmed * rms + 1e-7
        assert abs(fd[2] - gold['feat_digest'][2]) < 2 * fmed * gold['feat_digest'][2] + 1e-6
        f.backward(dy.to(dev))
        dsub = x.grad[:, :, ::32, ::32].cpu().numpy()
        de = np.abs(dsub - gold['dx_sub'])
        print('B4 %s dx: max err %.3e (max |ref| %.3e) median %.3e' % (dtype, de.max(), np.abs(gold['dx_sub']).max(), np.median(de)))
        assert de.max() < 3 * fmax * np.abs(gold['dx_sub']).max() + 1e-9
        names = [k for k, p in net.named_parameters() if not k.startswith('_fc')]
        assert names == [str(k) for k in gold['names']]
        gd = np.stack([_digest(p.grad) for k, p in net.named_parameters() if not k.startswith('_fc')])
        big = gold['gd'][:, 2] > 1e-2 * np.median(gold['gd'][:, 2])
        rel = np.abs(gd[:, 2] - gold['gd'][:, 2]) / (gold['gd'][:, 2] + 1e-12)
        print('B4 %s parameter-gradient norms: median rel err %.3e, worst non-degenerate %.3e' % (dtype, np.median(rel[big]), rel[big].max()))
        assert np.median(rel[big]) < gmed and rel[big].max() < (0.05 if dtype == torch.float32 else 0.6)
        del net, f, x
        torch.cuda.empty_cache()
