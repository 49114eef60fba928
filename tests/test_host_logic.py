"""CPU: host-side logic of the product package (no kernel launches)."""
import itertools
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT


def _digests(params):
    rows = []
    for p in params:
        t = p.detach().double()
        rows.append([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])
    return np.array(rows)


def test_constructors_reproduce_reference_init_and_keys(pkg):
    g = np.load(os.path.join(GOLDEN, 'step_n2_64.npz'))
    torch.manual_seed(41)
    G = pkg.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False))
    D = pkg.models_seg_gan.Discriminator(3, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024)
    assert list(G.state_dict().keys()) == [str(k) for k in g['state_keys_G']]
    assert list(D.state_dict().keys()) == [str(k) for k in g['state_keys_D']]
    assert [k for k, _ in G.named_parameters()] == [str(k) for k in g['param_names_G']]
    assert np.allclose(_digests(G.parameters()), g['init_G'], rtol=1e-10, atol=1e-12)
    assert np.allclose(_digests(D.parameters()), g['init_D'], rtol=1e-10, atol=1e-12)
    assert sum(p.numel() for p in G.parameters()) == 34226564 and sum(p.numel() for p in D.parameters()) == 23565505
    # checkpoint round trip in the reference's format: DataParallel prefix 'module.' (train_seg_gan.py:529)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'stage1'))
        torch.save({'module.' + k: v for k, v in G.net.state_dict().items()}, os.path.join(d, 'stage1', 'model.pth'))
        G2 = pkg.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False))
        G2.initialize_with_srresnet(d, dict(name='stage1'))
        assert all(torch.equal(a, b) for a, b in zip(G.state_dict().values(), G2.state_dict().values()))
        bad = {'module.' + k: v for k, v in list(G.net.state_dict().items())[:-1]}
        torch.save(bad, os.path.join(d, 'stage1', 'model.pth'))
        with pytest.raises(RuntimeError):
            G2.initialize_with_srresnet(d, dict(name='stage1'))


def test_no_cpu_fallback(pkg):
    G = pkg.archs.BasicBlock(4, 8)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        G(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError):
        pkg.losses.BCEDiceLoss()(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4))
    with pytest.raises(NotImplementedError):
        pkg.models_seg_gan.Generator(dict(arch='NoSuchArch', num_classes=3, input_channels=3, deep_supervision=False))


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, '_lib', None)
    monkeypatch.setattr(pkg._lib, 'LIB_PATH', '/nonexistent/libssunet_hip.so')
    with pytest.raises(RuntimeError, match='HIP extension not built'):
        pkg._lib.load()


def test_nhwc_layout_helpers(pkg):
    ops = pkg.ops
    x = ops.new_nhwc(2, 3, 5, 7, 'cpu')
    assert tuple(x.shape) == (2, 3, 5, 7) and x.stride() == (5 * 7 * 4, 1, 7 * 4, 4) and ops.nhwc_ld(x) == 4
    assert x._base is None                                   # a base tensor: in-place ops by callers are legal
    assert x.untyped_storage().nbytes() == 2 * 5 * 7 * 4 * 4
    y = torch.zeros(2, 8, 5, 7).contiguous(memory_format=torch.channels_last)
    assert ops.nhwc_ld(y) == 8
    assert ops.nhwc_ld(torch.zeros(2, 8, 5, 7)) is None      # NCHW-contiguous is converted, not reinterpreted
    assert ops.nhwc_ld(y[:, :4]) == 8 and ops.nhwc_ld(y[:, 4:]) == 8      # channel slices stay usable (16-B aligned)
    assert ops.nhwc_ld(y[:, :3]) is None                     # a 3-channel slice of foreign data has dirty pad lanes
    assert ops.nhwc_ld(torch.zeros(2, 6, 5, 7).contiguous(memory_format=torch.channels_last)) is None   # ld % 4
    assert ops.pad4(3) == 4 and ops.pad4(64) == 64


def _emulate(desc_taps, x, wt, transpose, gh, gw, oh, ow, in_s, out_s, oy, ox, cout):
    """numpy restatement of the ABI contract of ssg_conv2d_igemm_f32 (include/ssunet_hip.h)."""
    n, c, h, w = x.shape
    out = np.zeros((n, cout, oh, ow))
    for (ky, kx, dy, dx) in desc_taps:
        wk = wt[:, :, ky, kx].T if transpose else wt[:, :, ky, kx]        # [cout, cred]
        for gy in range(gh):
            iy = gy * in_s + dy
            if not 0 <= iy < h:
                continue
            for gx in range(gw):
                ix = gx * in_s + dx
                if 0 <= ix < w:
                    out[:, :, gy * out_s + oy, gx * out_s + ox] += x[:, :, iy, ix] @ wk.T
    return out


@pytest.mark.parametrize('k,s,p,h,w', [(3, 1, 1, 6, 7), (3, 2, 1, 8, 8), (3, 2, 1, 7, 9), (1, 1, 0, 5, 5), (1, 2, 0, 6, 6)])
def test_tap_geometry_forward_and_dgrad(pkg, k, s, p, h, w):
    """The tap lists the host builds (forward, stride-1 dgrad, strided dgrad parity classes) give
    F.conv2d and its input gradient when interpreted by the ABI contract."""
    g = torch.Generator().manual_seed(0)
    cin, cout, n = 3, 4, 2
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64).requires_grad_(True)
    wt = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64)
    y = F.conv2d(x, wt, None, s, p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    oh, ow = y.shape[2:]
    assert (oh, ow) == (pkg.ops._out_size(h, k, s, p), pkg.ops._out_size(w, k, s, p))
    fwd = _emulate(pkg.ops._taps_fwd(k, k, p), x.detach().numpy(), wt.numpy(), False, oh, ow, oh, ow, s, 1, 0, 0, cout)
    assert np.allclose(fwd, y.detach().numpy(), atol=1e-12)
    dx = np.zeros((n, cin, h, w))
    if s == 1:
        taps = [(ky, kx, p - ky, p - kx) for ky in range(k) for kx in range(k)]
        dx = _emulate(taps, dy.numpy(), wt.numpy(), True, h, w, h, w, 1, 1, 0, 0, cin)
    else:
        for py, px in itertools.product(range(s), range(s)):
            taps = [(ky, kx, (py + p - ky) // s, (px + p - kx) // s) for ky in range(k) for kx in range(k)
                    if (py + p - ky) % s == 0 and (px + p - kx) % s == 0]
            gh, gw = (h - py + s - 1) // s, (w - px + s - 1) // s
            if taps and gh > 0 and gw > 0:
                dx += _emulate(taps, dy.numpy(), wt.numpy(), True, gh, gw, h, w, 1, s, py, px, cin)
    assert np.allclose(dx, x.grad.numpy(), atol=1e-12)


def test_average_meter_and_optimizer_gate(pkg):
    m = pkg.utils.AverageMeter()
    m.update(torch.tensor(2.0), 3); m.update(4.0, 1)
    assert float(m.avg) == pytest.approx(2.5)
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert pkg.optim._supported(torch.optim.Adam(p))
    assert not pkg.optim._supported(torch.optim.Adam(p, amsgrad=True))
    assert not pkg.optim._supported(torch.optim.SGD(p, lr=0.1))
    assert pkg.train_seg_gan.ALPA == 1e-4 and pkg.train_seg_gan.BETA == 1e-3 and pkg.train_seg_gan.GRAD_CLIP == 0.8


def test_layout_rules():
    """oracle/ is test infrastructure: nothing in the product package, bench's GPU leg or the header
    may import it; only tests/, __graft_entry__.smoke() and bench.cpu_baseline() do."""
    import re
    pkg_dir = os.path.join(ROOT, 'ssunet-gan_amd')
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dirpath, f)
                assert '/root/reference' not in src, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    assert len(re.findall(r'from oracle', bench)) == 1 and 'def cpu_baseline' in bench
    assert bench.index('from oracle') > bench.index('def cpu_baseline') and bench.index('from oracle') < bench.index('def main')
    for f in ('tests/golden/blocks.npz', 'tests/golden/step_n2_64.npz', 'tests/golden/step_n4_256.npz', 'oracle/gen_golden.py'):
        assert os.path.exists(os.path.join(ROOT, f))
