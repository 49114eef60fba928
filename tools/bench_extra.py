"""Throughput of the rows beside the headline path (per-op A11/A12), one JSON
object per line.  Not the graded metric (that is bench.py); these are the measurements DESIGN.md quotes for them.
Usage (GPU box): python tools/bench_extra.py"""
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S  # noqa: E402

dev = torch.device('cuda')


def timed(fn, warm, n):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    g = torch.Generator().manual_seed(7)
    out = []
    # C5: eval-mode generator over the 36 patches of a 2048^2 image (patches resized to 512^2), batched
    torch.manual_seed(41)
    G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev)
    patches = torch.randn(36, 3, 512, 512, generator=g)
    for bs in (1, 12):
        dt = timed(lambda: S.aerial_image_segmentation_api.infer_patches(G, patches, batch_size=bs), 1, 3)
        out.append({'what': 'C5 sliding-window inference, 36 x 3x512x512 patches, eval-mode G (BN folded), batch %d, incl. H2D/D2H' % bs,
                    'patches_per_s': round(36 / dt, 2), 's_per_image': round(dt, 3)})
    # batch 1 on resident inputs: kernel-by-kernel launches vs hipGraph replay (what the launch path costs at batch 1)
    xb = patches[:1].to(dev)
    G.eval()
    with torch.no_grad():
        dt_eager = timed(lambda: S.ops.sigmoid(G(xb)), 3, 20)
        gr, sin, sout = S.aerial_image_segmentation_api._graph_for(G, xb.shape, dev)
        sin.copy_(xb)
        dt_graph = timed(lambda: gr.replay(), 3, 20)
    out.append({'what': 'C5 one 3x512x512 patch, eval-mode G, input resident: eager launches vs hipGraph replay',
                'ms_eager': round(dt_eager * 1e3, 3), 'ms_graph': round(dt_graph * 1e3, 3)})
    # N1: stage-1 trainer step (G only, BCEDice, weight clamp, Adam with weight decay)
    cfg = {'clip': 0.7, 'num_classes': 3, 'deep_supervision': False}
    model = S.archs.UNet_R_SS_v2(3, 3, False).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    crit = S.losses.BCEDiceLoss()
    x = torch.randn(16, 3, 512, 512, generator=g); t = (torch.rand(16, 3, 512, 512, generator=g) > 0.5).float()
    loader = [(None, x, t, None, None)]
    import contextlib, io
    def stage1():
        with contextlib.redirect_stdout(io.StringIO()):
            S.train.train(2, cfg, loader, model, crit, opt, None)
    dt = timed(stage1, 2, 4)
    out.append({'what': 'N1 stage-1 trainer step (train.py:68-137), 16 x 3x512x512, fp32, incl. H2D of the batch',
                'images_per_s': round(16 / dt, 2), 'ms_per_step': round(dt * 1e3, 1)})
    del model, opt, G
    torch.cuda.empty_cache()
    # C4: EfficientNet-B4 encoder extract_features fwd+bwd, N=4 @ 1024^2 (fp32 here; the reference defines no B4 U-Net)
    for name, n, hw in (('efficientnet-b0', 16, 512), ('efficientnet-b4', 4, 1024)):
        enc = S.efficientnet_pytorch.EfficientNet.from_name(name).to(dev).train()
        xin = torch.randn(n, 3, hw, hw, generator=g).to(dev)
        def step():
            enc.zero_grad(set_to_none=True)
            f = enc.extract_features(xin)
            f.sum().backward()
        dt = timed(step, 2, 5)
        out.append({'what': 'C4/A10 %s extract_features fwd+bwd, %d x 3x%dx%d, fp32, train mode' % (name, n, hw, hw),
                    'images_per_s': round(n / dt, 2), 'ms_per_step': round(dt * 1e3, 1)})
        del enc
        torch.cuda.empty_cache()
    # A11: xResidualBlock fwd+bwd
    blk = S.xresidualblock.xResidualBlock(64, 64).to(dev).train()
    xin = torch.randn(16, 64, 256, 256, generator=g).to(dev).requires_grad_(True)
    def xres():
        blk.zero_grad(set_to_none=True)
        blk(xin).sum().backward()
    dt = timed(xres, 2, 5)
    out.append({'what': 'A11 xResidualBlock(64,64) fwd+bwd, 16 x 64x256x256', 'ms_per_step': round(dt * 1e3, 2),
                'gb_per_s_of_input_tensor': round(16 * 64 * 256 * 256 * 4 / dt / 1e9, 1)})
    # A12: spectral norm on the discriminator's 8 conv weights (one power iteration each, as a training forward does)
    D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
    convs = [m for m in D.modules() if isinstance(m, nn.Conv2d)]
    for m in convs:
        S.spectral_norm.spectral_norm(m)
    hooks = [(next(h for h in m._forward_pre_hooks.values() if isinstance(h, S.spectral_norm.SpectralNorm)), m) for m in convs]

    def sn():
        with torch.no_grad():
            for h, m in hooks:
                h(m, None)
    dt = timed(sn, 2, 10)
    wbytes = sum(m.weight_orig.numel() * 4 for m in convs)
    out.append({'what': 'A12 spectral_norm power iteration + W/sigma on the 8 discriminator conv weights (%.1f MB)' % (wbytes / 1e6),
                'us_total': round(dt * 1e6, 1)})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == '__main__':
    main()
