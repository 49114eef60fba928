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


# ----------------------------------------------------------------------------- host half of the API (api.py:302-373, 119-234)
def _write_png(path, arr_bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(arr_bgr[:, :, ::-1])).save(path)


def test_get_patched_input_double_normalisation_vs_numpy_restatement_unpinned(pkg, tmp_path):
    """api.py:336-373 on a 2048^2 image: 36 patches of 1024^2 (overlap 0.5) -> 512^2, albumentations Normalize() on the BGR
    image and THEN /255 again (:364-367).  UNPINNED: cv2 / albumentations exist nowhere in this environment, so the expected
    values are an independent numpy evaluation of the documented arithmetic, not reference-generated vectors."""
    A = pkg.aerial_image_segmentation_api
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    p = str(tmp_path / 'image_0.png')
    _write_png(p, img)
    assert np.array_equal(A.imread_bgr(p), img)
    cfg = dict(patch_size=1024, input_w=512, input_h=512, patch_overlap=0.5, num_classes=3)
    full, patches, masks = A.get_patched_input(p, cfg, False)
    assert np.array_equal(full, img) and patches.shape == (36, 3, 512, 512) and patches.dtype == np.float32
    assert masks.shape == (36, 1024, 1024, 3)
    org = A.patch_origins(2048, 2048, 1024, 0.5)
    assert org[0] == (0, 0) and org[9] == (1024, 1024) and len(org) == 36
    mean = np.array((0.485, 0.456, 0.406)); std = np.array((0.229, 0.224, 0.225))
    for k in (0, 7, 20, 35):
        h, w = org[k]
        blk = img[h:h + 2, w:w + 2].astype(np.float64).reshape(4, 3)               # top-left output pixel of patch k
        px = np.floor(blk.mean(0) + 0.5)                                             # 2x2 box mean, rounded half up (uint8)
        want = ((px / 255.0 - mean) / std) / 255.0                                   # channel 0 is BLUE: BGR through RGB statistics
        assert np.allclose(patches[k, :, 0, 0], want, rtol=2e-6, atol=1e-8), (k, patches[k, :, 0, 0], want)
    with pytest.raises(NotImplementedError):
        A.resize_u8(img[:100, :100], 33, 33)


def test_patch_merge_thresholds_and_resize_vs_numpy_restatement_unpinned(pkg):
    A = pkg.aerial_image_segmentation_api
    cfg = dict(num_classes=2)
    img = np.zeros((64, 64, 3), np.uint8)
    org = A.patch_origins(64, 64, 32, 0.5)
    rng = np.random.default_rng(4)
    probs = [rng.random((2, 16, 16)).astype(np.float32) for _ in org]               # inference at 16^2, patches of 32^2
    out = A.patch_merge(img, probs, 32, cfg, 0.5)
    assert len(out) == 2 and out[0].dtype == np.uint8 and out[0].shape == (64, 64)
    assert set(np.unique(out[0])) <= {0, 255}
    # independent evaluation of one pixel: patches covering (5, 40), each (p*255 -> uint8 -> 2x bilinear -> >127), averaged, >127
    def up_val(m, y, x):
        u8 = (m * 255).astype('uint8').astype(np.float64)
        sy, sx = (y + 0.5) / 2 - 0.5, (x + 0.5) / 2 - 0.5
        y0, x0 = int(np.floor(sy)), int(np.floor(sx)); fy, fx = sy - y0, sx - x0
        cl = lambda v, n: min(max(v, 0), n - 1)
        v = ((1 - fy) * ((1 - fx) * u8[cl(y0, 16), cl(x0, 16)] + fx * u8[cl(y0, 16), cl(x0 + 1, 16)]) +
             fy * ((1 - fx) * u8[cl(y0 + 1, 16), cl(x0, 16)] + fx * u8[cl(y0 + 1, 16), cl(x0 + 1, 16)]))
        v = np.floor(v + 0.5)
        return 255.0 if v > 127 else 0.0
    acc, cnt = 0.0, 0
    for (h, w), m in zip(org, probs):
        if h <= 5 < h + 32 and w <= 40 < w + 32:
            acc += up_val(m[1], 5 - h, 40 - w) / 255.0; cnt += 1
    v = int(acc / cnt * 255)
    want = 255 if (127 < v < 255 or v == 255) else 0
    assert out[1][5, 40] == want
    assert np.array_equal(A.post_process_resized_mask(np.array([0, 1, 127, 128, 254, 255], np.uint8)), [0, 0, 0, 255, 255, 255])
    lab = np.zeros((16, 16, 3), np.uint8); lab[:8] = (255, 0, 0); lab[8:] = (0, 0, 255)
    assert A.mask_convert(lab, 1, 32)[:14].min() == 255 and A.mask_convert(lab, 2, 32)[:14].max() == 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_full_size_2048_image_36_patches(pkg, dev, tmp_path):
    """BASELINE config 5 at full size: one 2048 x 2048 image -> 36 overlapping 1024^2 patches -> 512^2 inference -> merged
    class masks, through load_segmentation_models / get_patched_input / segmentation_inference_full; batch 12, the reference's
    batch 1 (hipGraph replay) and batch 1 kernel by kernel agree, and two patches are checked against the CPU oracle."""
    import json
    import yaml
    from oracle import seg_gan_cpu as O
    A = pkg.aerial_image_segmentation_api
    torch.manual_seed(41)
    src = pkg.archs.UNet_R_SS_v2(3, 3, False)
    g = torch.Generator().manual_seed(7)
    src.to(dev).train()
    with torch.no_grad():
        src(torch.randn(2, 3, 64, 64, generator=g).to(dev))                          # non-trivial running statistics
    root = tmp_path / 'models' / 'gen512'
    root.mkdir(parents=True)
    torch.save(src.state_dict(), str(root / 'model.pth'))
    yaml.safe_dump(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False, input_w=512, input_h=512),
                   open(str(root / 'config.yml'), 'w'))
    cfg_file = tmp_path / 'config.json'
    cfg_file.write_text(json.dumps(dict(file_path=dict(model_path=str(tmp_path / 'models')), val_config=dict(name='gen512', patch_overlap=0.5))))
    model, config = A.load_segmentation_models(str(cfg_file))
    assert config['patch_size'] == 1024 and config['patch_overlap'] == 0.5 and not model.training
    for (k, a), (_, b) in zip(model.state_dict().items(), src.state_dict().items()):
        assert torch.equal(a, b), k
    rng = np.random.default_rng(5)
    # per-pixel noise: exactly flat regions would put exact ties into every 2x2 max-pool window, where two implementations
    # may legitimately pick different argmax positions (the indices feed max-unpool)
    img = rng.integers(0, 256, (2048, 2048, 3), dtype=np.uint8)
    p = str(tmp_path / 'image_1.png')
    _write_png(p, img)
    full, patches, masks = A.get_patched_input(p, config, False)
    assert patches.shape == (36, 3, 512, 512)
    probs12 = A.infer_patches(model, patches, batch_size=12).numpy()
    probs1g = A.infer_patches(model, patches, batch_size=1, graph=True).numpy()      # hipGraph replay per patch
    probs1 = A.infer_patches(model, patches[:3], batch_size=1, graph=False).numpy()
    # batch 1 splits the reduction of the deep levels over more workgroups than batch 12 does (split-K): same arithmetic,
    # another fp32 summation order; batch 1 with and without the hipGraph is the same launch sequence -> bit-identical
    # (round 4: batch 12 fills the chip and takes the 32-channel-chunk kernels where batch 1 does not; among 54 M pooled windows a
    # near-tie argmax may then fall differently, which moves single pixels by more than rounding -- so robust statistics here too)
    d12 = np.abs(probs12 - probs1g)
    assert np.median(d12) < 1e-6 and (d12 > 5e-5).mean() < 1e-3 and d12.max() < 1e-2, 'batch 12 vs batch 1: max %.3e' % d12.max()
    assert np.abs(probs1 - probs1g[:3]).max() == 0.0
    # CPU oracle on two patches (eval mode, same weights)
    Go = O.UNetRSSv2CPU(3, 3, False)
    Go.load_state_dict({k: v.cpu() for k, v in src.state_dict().items()})
    Go.eval()
    with torch.no_grad():
        ref = torch.sigmoid(Go(torch.from_numpy(patches[[0, 17]]))).numpy()
    err = np.abs(ref - probs12[[0, 17]])
    # 1.5 M pooled windows per patch: a handful of near-tie argmax choices may differ between the two implementations (they feed
    # max-unpool), so the bound is on robust statistics, as in the 256^2 step test
    assert np.median(err) < 2e-6 and (err > 5e-5).mean() < 1e-3 and err.max() < 1e-2, 'vs CPU oracle: max %.3e median %.3e' % (err.max(), np.median(err))
    all_mask, gt_mask = A.segmentation_inference_full(model, full, patches, masks, config, False, batch_size=12)
    assert len(all_mask) == 3 and all_mask[1].shape == (2048, 2048) and all_mask[1].dtype == np.uint8
    assert set(np.unique(all_mask[1])) <= {0, 255} and gt_mask is all_mask
    want = A.patch_merge(full, list(probs12), 1024, config, 0.5)
    assert all(np.array_equal(a, b) for a, b in zip(all_mask, want))


@pytest.mark.gpu
def test_graph_replay_follows_load_state_dict(pkg, dev):
    """a captured hipGraph bakes in the packed / BN-folded operands of the weights it was captured with.  After
    load_state_dict() with other weights (tensor versions change, the weight epoch does not) `graph=True` must capture anew --
    not replay the old weights, not read operands an eager forward in between has freed."""
    A = pkg.aerial_image_segmentation_api
    torch.manual_seed(5)
    model = pkg.archs.UNet_R_SS_v2(3, 3, False).to(dev).eval()
    torch.manual_seed(6)
    other = pkg.archs.UNet_R_SS_v2(3, 3, False).to(dev).eval()
    g = torch.Generator().manual_seed(8)
    patches = torch.randn(3, 3, 64, 64, generator=g).numpy()
    a_graph = A.infer_patches(model, patches, batch_size=1, graph=True).numpy()
    a_eager = A.infer_patches(model, patches, batch_size=1, graph=False).numpy()
    assert np.abs(a_graph - a_eager).max() < 1e-6
    model.load_state_dict(other.state_dict())
    b_eager_first = A.infer_patches(model, patches, batch_size=1, graph=False).numpy()   # eager in between: refills the caches
    b_graph = A.infer_patches(model, patches, batch_size=1, graph=True).numpy()
    b_ref = A.infer_patches(other, patches, batch_size=1, graph=False).numpy()
    assert np.abs(b_eager_first - b_ref).max() < 1e-6
    assert np.abs(b_graph - b_ref).max() < 1e-6, 'graph replay after load_state_dict returns the old weights\' output'
    assert np.abs(b_graph - a_graph).max() > 1e-3                                         # the two models do differ
