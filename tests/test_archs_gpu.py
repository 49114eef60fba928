"""The other exported generators (UNet, NestedUNet +- deep supervision, SSUNet,
UNet_ori, UNet_B_SS, UNet_R_SS, AttUNet) on the HIP path vs golden vectors from the reference's archs.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = [('UNet', False), ('NestedUNet', False), ('NestedUNet', True), ('SSUNet', False), ('UNet_ori', False),
         ('UNet_B_SS', False), ('UNet_R_SS', False), ('AttUNet', False)]


@pytest.mark.parametrize('name,ds', CASES)
def test_arch_forward_backward(pkg, dev, name, ds):
    gold = np.load(os.path.join(GOLDEN, 'archs.npz'))
    tag = name + ('_ds' if ds else '')
    torch.manual_seed(52)
    m = pkg.archs.__dict__[name](3, 3, ds).to(dev).train()
    x = torch.from_numpy(gold['x']).to(dev).requires_grad_(True)
    out = m(x)
    outs = out if isinstance(out, list) else [out]
    ref = gold[tag + '_y']
    assert len(outs) == ref.shape[0]
    tot = 0
    for i, o in enumerate(outs):
        e = np.abs(o.detach().cpu().numpy()[..., ::2, ::2] - ref[i])
        assert e.max() < 2e-4 * max(1.0, np.abs(ref[i]).max()), '%s out %d err %.3e' % (tag, i, e.max())
        dy = torch.randn(o.shape, generator=torch.Generator().manual_seed(99 + i)).to(dev)
        tot = tot + (o * dy).sum()
    tot.backward()
    dxr = gold[tag + '_dx']
    e = np.abs(x.grad.cpu().numpy()[..., ::2, ::2] - dxr)
    # batch norms at the deepest levels see few samples (2x2 pixels x 4 images): their backward amplifies
    # fp32 noise, so the input gradient is bounded on its typical element, not its worst
    assert np.median(e) < 2e-3 * np.abs(dxr).max(), '%s dx: median %.3e max %.3e (scale %.3e)' % (tag, np.median(e), e.max(), np.abs(dxr).max())
    gd = np.array([[0, 0, (p.grad.double() ** 2).sum().sqrt().item()] for p in m.parameters() if p.grad is not None])
    refg = gold[tag + '_gd']
    refg = refg[refg[:, 2] > 0] if len(refg) != len(gd) else refg
    rel = np.abs(gd[:, 2] - refg[:, 2]) / (refg[:, 2] + 1e-9)
    big = refg[:, 2] > 1e-2 * np.median(refg[:, 2])
    assert np.median(rel) < 5e-3 and rel[big].max() < 0.25, '%s grads: median %.3e max %.3e' % (tag, np.median(rel), rel[big].max())


def test_generator_accepts_every_built_arch(pkg):
    for name in pkg.archs.__all__:
        g = pkg.models_seg_gan.Generator(dict(arch=name, num_classes=3, input_channels=3, deep_supervision=False))
        assert g.net.__class__.__name__ == name
