"""Sliding-window inference (BASELINE config 5): batched eval-mode generator with
folded batch norms vs the reference's one-patch-per-forward loop (tests/golden/infer.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_patch_order_matches_reference(pkg):
    gold = np.load(os.path.join(GOLDEN, 'infer.npz'))
    h, w = [int(v) for v in gold['img_hw']]
    org = pkg.aerial_image_segmentation_api.patch_origins(h, w, int(gold['p_size']), 0.5)
    assert np.array_equal(np.array(org), gold['origins'])
    img = np.arange(h * w).reshape(h, w, 1)
    ip, mp = pkg.aerial_image_segmentation_api.patch_gen(img, img, 16, 0.5)
    assert len(ip) == len(org) and ip[3].shape == (16, 16, 1)
    merged = pkg.aerial_image_segmentation_api.patch_merge_mean(h, w, [np.ones((2, 16, 16))] * len(org), 16, 0.5, 2)
    assert merged.shape == (2, h, w) and np.allclose(merged, 1.0)          # full coverage, averages of ones


@pytest.mark.gpu
def test_batched_eval_inference_matches_per_patch_reference(pkg, dev):
    gold = np.load(os.path.join(GOLDEN, 'infer.npz'))
    torch.manual_seed(41)
    model = pkg.archs.UNet_R_SS_v2(3, 3, False).to(dev)
    model.train()
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        model(inp.to(dev))                                             # same running-stat update as the fixture
    probs = pkg.aerial_image_segmentation_api.infer_patches(model, gold['patches'], batch_size=4).numpy()
    e = np.abs(probs - gold['probs'])
    assert e.max() < 5e-5, 'batched eval inference: max err %.3e' % e.max()
    one = pkg.aerial_image_segmentation_api.infer_patches(model, gold['patches'][:1], batch_size=1).numpy()
    assert np.abs(one - probs[:1]).max() < 1e-5                        # batching does not change a patch's result
    cfg = dict(patch_size=64, patch_overlap=0.5, num_classes=3)
    full = np.zeros((64, 64, 3))
    m, _ = pkg.aerial_image_segmentation_api.segmentation_inference(model, full, gold['patches'][:4], None, cfg)
    assert m.shape == (3, 64, 64) and np.isfinite(m).all()
