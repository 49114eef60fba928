"""Stage-1 trainer (train.py:68-137) on the HIP path vs golden vectors from the
reference's modules (tests/golden/stage1_n2_64.npz): Adam with weight decay, weight clamp +-0.7
between forward and backward."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _digests(params, grads=False):
    rows = []
    for p in params:
        t = (p.grad if grads else p).detach().double().cpu()
        rows.append([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])
    return np.array(rows)


def test_stage1_two_steps(pkg, dev):
    gold = np.load(os.path.join(GOLDEN, 'stage1_n2_64.npz'))
    torch.manual_seed(41)
    model = pkg.archs.UNet_R_SS_v2(3, 3, False).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(2, 3, 64, 64, generator=g); tgt = (torch.rand(2, 3, 64, 64, generator=g) > 0.5).float()
    config = dict(clip=0.7, num_classes=3, deep_supervision=False)
    loader = [(None, inp, tgt, None, None)]
    tap = {}
    model.register_forward_hook(lambda m, i, o: tap.__setitem__('logits', o.detach().clone()))
    for s in range(2):
        r = pkg.train.train(s, config, loader, model, pkg.losses.BCEDiceLoss(), opt, None)
        ref = gold['s%d_scalars' % s]
        tol = np.array([2e-5, 1e-4, 1e-4]) if s == 0 else np.array([5e-3, 1e-2, 5e-3])
        got = np.array([r['loss'], r['iou'], r['dice']])
        assert (np.abs(got - ref) < tol).all(), 'step %d: %s vs %s' % (s, got, ref)
        e = np.abs(tap['logits'].cpu().numpy() - gold['s%d_logits' % s])
        assert (e.max() < 2e-4) if s == 0 else (np.median(e) < 2e-2), 'step %d logits max %.3e median %.3e' % (s, e.max(), np.median(e))
        if s == 0:
            gd = _digests(model.parameters(), True)
            rel = np.abs(gd[:, 2] - gold['s0_grads'][:, 2]) / (gold['s0_grads'][:, 2] + 1e-12)
            assert np.median(rel) < 2e-3, 'grad digests median %.3e' % np.median(rel)
            # every weight is inside [-0.7, 0.7] + one Adam step (BN weights start at 1.0 and are clamped)
            assert max(p.abs().max().item() for p in model.parameters()) <= 0.7 + 2e-4
            pd = _digests(model.parameters())
            numel = np.array([p.numel() for p in model.parameters()])
            assert (np.abs(pd[:, 1] - gold['s0_params'][:, 1]) <= 1e-5 * gold['s0_params'][:, 1] + 0.5 * 1e-4 * numel).all()
    rv = pkg.train.validate(config, loader, model, pkg.losses.BCEDiceLoss())
    assert all(np.isfinite(v) for v in rv.values())
