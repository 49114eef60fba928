"""GPU, world_size 2 on ONE device (gloo backend moving CUDA tensors; the multi-GPU job uses the same
code over RCCL): data-parallel G+D step with sync-BN and bucketed gradient averaging == one process
over the concatenated batch with the sync-BN formula."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _build(S, dev):
    torch.manual_seed(41)
    G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev).train()
    D = S.models_seg_gan.Discriminator(3, 3, 64, 8, 1024).to(dev).train()
    return G, D


def _batch():
    g = torch.Generator().manual_seed(7)
    return torch.randn(4, 3, 64, 64, generator=g), (torch.rand(4, 3, 64, 64, generator=g) > 0.5).float()


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
        import torch.nn as nn
        import ssunet_gan_amd as S
        S.dp.init_from_env(backend='gloo')
        dev = torch.device('cuda', 0)
        G, D = _build(S, dev)
        S.dp.broadcast_parameters(G); S.dp.broadcast_parameters(D)
        S.dp.convert_sync_batchnorm(G); S.dp.convert_sync_batchnorm(D)
        og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
        inp, tgt = _batch()
        sl = slice(rank * 2, rank * 2 + 2)
        sg, sd = S.dp.grad_syncs(G, D)
        assert sg is not None and len(sg.buckets) >= 1
        loss, iou, dice, closs, adv_g, adv_d = S.train_seg_gan.gan_step(inp[sl].to(dev), tgt[sl].to(dev), G, D, S.losses.BCEDiceLoss(),
                                                                       nn.BCEWithLogitsLoss(), nn.MSELoss(), og, od, 3, sg, sd)
        torch.cuda.synchronize()
        out = dict(loss=float(loss), iou=float(iou), dice=float(dice),
                   g_digest=[float(p.detach().double().abs().sum()) for p in G.parameters()],
                   d_grad=[float(p.grad.detach().double().norm()) for p in D.parameters()],
                   rm=float(G.net.conv0_0.bn1.running_mean.double().abs().sum()))
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
        q.put((rank, 'ok', out))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


@pytest.mark.timeout(600)
def test_dp2_syncbn_step_equals_single_process_full_batch(pkg, dev):
    import torch.nn as nn
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), [r[1] for r in res]
    r0, r1 = res[0][2], res[1][2]
    # replicas stay identical: same parameters after the step on both ranks, same reduced metrics
    assert np.allclose(r0['g_digest'], r1['g_digest'], rtol=0, atol=0)
    assert r0['loss'] == pytest.approx(r1['loss'], abs=1e-9) and r0['iou'] == pytest.approx(r1['iou'], abs=1e-12)
    # single process, batch of 4, sync-BN formula (clamp(var, eps)^-1/2, batchnorm.py:127)
    S = pkg
    G, D = _build(S, dev)
    for m in list(G.modules()) + list(D.modules()):
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m._ssg_var_mode = 1
    og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
    inp, tgt = _batch()
    loss, iou, dice, closs, adv_g, adv_d = S.train_seg_gan.gan_step(inp.to(dev), tgt.to(dev), G, D, S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(),
                                                                   nn.MSELoss(), og, od, 3)
    assert abs(float(loss) - r0['loss']) < 2e-5, (float(loss), r0['loss'])
    assert abs(float(iou) - r0['iou']) < 1e-4 and abs(float(dice) - r0['dice']) < 1e-4
    gd = np.array([float(p.detach().double().abs().sum()) for p in G.parameters()])
    numel = np.array([p.numel() for p in G.parameters()])
    assert (np.abs(gd - np.array(r0['g_digest'])) <= 1e-5 * gd + (0.5 * numel + 2) * 2e-5).all()
    rm = float(G.net.conv0_0.bn1.running_mean.double().abs().sum())
    assert abs(rm - r0['rm']) < 1e-5 * max(1.0, rm)
    dg = np.array([float(p.grad.detach().double().norm()) for p in D.parameters()])
    rel = np.abs(dg - np.array(r0['d_grad'])) / (dg + 1e-12)
    assert np.median(rel) < 5e-2
