"""Sliding-window full-image inference, MI355X-native (mirrors the GPU side of the reference's
scripts/aerial_image_segmentation_api.py -- SURVEY.md 8f row N2 / BASELINE config 5).

In scope (GPU): the eval-mode generator forward over all patches of an image.  The reference runs one
patch per forward (`api.py:385-390`); here patches are batched and every BasicBlock runs as three MFMA
launches with its batch norms folded in, then one sigmoid kernel -- same values, no per-patch sync.
Host side kept: the patch order of `patch_gen` (4 corner-anchored sweeps, `api.py:45-116`) and the
overlap-averaging of `patch_merge` (`api.py:119-217`) without its cv2-dependent resize/hysteresis steps
(cv2 and albumentations are host-only dependencies that this package does not take on).
"""
import math

import numpy as np
import torch

from . import ops


def patch_origins(img_h, img_w, p_size, overlap=0.5):
    """(h1, w1) of every patch in the reference's order: four sweeps anchored at the top-left,
    bottom-right, bottom-left and top-right corners (api.py:45-116)."""
    step = int(math.ceil((1 - overlap) * p_size))
    i_w = int(math.floor((img_w - p_size) / step)) + 1
    i_h = int(math.floor((img_h - p_size) / step)) + 1
    sweeps = []
    for anchor_bottom, anchor_right in ((False, False), (True, True), (True, False), (False, True)):
        for i in range(i_w):
            for j in range(i_h):
                w1 = img_w - i * step - p_size if anchor_right else i * step
                h1 = img_h - j * step - p_size if anchor_bottom else j * step
                if h1 < 0 or w1 < 0 or h1 + p_size > img_h or w1 + p_size > img_w:
                    raise ValueError('patch outside the image (img %dx%d, patch %d)' % (img_h, img_w, p_size))
                sweeps.append((h1, w1))
    return sweeps


def patch_gen(img, mask, p_size, overlap=0.5):
    org = patch_origins(img.shape[0], img.shape[1], p_size, overlap)
    return ([img[h:h + p_size, w:w + p_size, :] for h, w in org], [mask[h:h + p_size, w:w + p_size, :] for h, w in org])


def infer_patches(model, img_patch_set, batch_size=16):
    """sigmoid(model(patch)) for every patch, batched.  img_patch_set: [P, C, H, W] float32 (numpy or
    tensor, as `get_patched_input` builds it); returns a float32 tensor [P, num_classes, H, W] on the host."""
    x = torch.as_tensor(img_patch_set, dtype=torch.float32)
    model.eval()
    outs = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch_size):
            xb = x[i:i + batch_size].cuda(non_blocking=True)
            y = ops.sigmoid(model(xb))
            outs.append(y)
    out = torch.cat([o.contiguous() for o in outs], 0) if len(outs) > 1 else outs[0].contiguous()
    return out.cpu()


def patch_merge_mean(img_h, img_w, masks, p_size, p_overlap, num_classes):
    """Overlap-average of per-patch probability maps at patch resolution (the arithmetic core of
    patch_merge, api.py:119-217, without the uint8 round trip, cv2.resize and hysteresis)."""
    org = patch_origins(img_h, img_w, p_size, p_overlap)
    if len(org) != len(masks):
        raise ValueError('expected %d patches, got %d' % (len(org), len(masks)))
    merged = np.zeros((num_classes, img_h, img_w), dtype=np.float64)
    div = np.zeros((img_h, img_w), dtype=np.float64)
    for (h, w), m in zip(org, masks):
        m = np.asarray(m)
        if m.shape[-2:] != (p_size, p_size):
            raise NotImplementedError('inference size != patch size needs the cv2.resize step of the reference')
        merged[:, h:h + p_size, w:w + p_size] += m
        div[h:h + p_size, w:w + p_size] += 1
    div[div == 0] = 1.0
    return merged / div


def segmentation_inference(model, img_input, img_patch_set, mask_patch_set, config, gt_mask_flag=False, batch_size=16):
    """api.py:376-410 for the prediction branch: batched GPU inference + overlap averaging."""
    probs = infer_patches(model, img_patch_set, batch_size).numpy()
    all_class_mask = patch_merge_mean(img_input.shape[0], img_input.shape[1], list(probs), config['patch_size'],
                                      config['patch_overlap'], config['num_classes'])
    return all_class_mask, all_class_mask
