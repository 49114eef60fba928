"""CPU, world_size 2 over gloo: the data-parallel plumbing (dp.GradSync bucketing/averaging,
metric-sum reduction).  The GPU path uses the same code over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        import ssunet_gan_amd as S
        r, w, _ = S.dp.init_from_env(backend='gloo')
        assert (r, w) == (rank, world) and S.dp.is_dist()
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3), nn.Tanh(), nn.Linear(3, 1))
        S.dp.broadcast_parameters(model)
        x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1)); y = torch.randn(8, 1, generator=torch.Generator().manual_seed(2))
        # reference: full batch on one process (mean loss)
        ref = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3), nn.Tanh(), nn.Linear(3, 1))
        ref.load_state_dict(model.state_dict())
        nn.functional.mse_loss(ref(x), y).backward()
        sync = S.dp.GradSync(model, bucket_bytes=16)            # tiny buckets: several all-reduces
        assert len(sync.buckets) >= 3
        xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
        for it in range(2):                                     # twice: buffers are re-bound each step
            for p in model.parameters():
                p.grad = None
            sync.begin()
            nn.functional.mse_loss(model(xs), ys).backward()
            sync.finish()
            for p, q_ in zip(model.parameters(), ref.parameters()):
                assert torch.allclose(p.grad, q_.grad, atol=1e-6), 'averaged grads != full-batch grads'
                assert p.grad.data_ptr() % 16 == 0
        # hooks are inert outside begin()/finish(): "stale" grads of the other network are untouched
        for p in model.parameters():
            p.grad = None
        nn.functional.mse_loss(model(xs), ys).backward()
        g_local = [p.grad.clone() for p in model.parameters()]
        assert not all(torch.allclose(a, b.grad, atol=1e-6) for a, b in zip(g_local, ref.parameters()))
        # a bucket whose parameters never receive a gradient is flushed by finish()
        sync.begin()
        model[4](torch.randn(2, 3)).sum().backward()
        sync.finish()
        # metric sums: ratios of global sums, not means of ratios
        sums = torch.tensor([3.0, 10.0, 2.5, 6.0, 7.0], dtype=torch.float64) * (rank + 1)
        iou, dice = S.dp.reduce_metric_sums(sums)
        assert abs(iou.item() - (9 + 1e-5) / (30 + 1e-5)) < 1e-12 and abs(dice.item() - (15 + 1e-5) / (39 + 1e-5)) < 1e-12
        assert abs(S.dp.reduce_mean(torch.tensor(float(rank))).item() - 0.5) < 1e-12
        bn = nn.Sequential(nn.BatchNorm2d(4))
        S.dp.convert_sync_batchnorm(bn)
        assert bn[0]._ssg_sync_group is not None
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:          # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(180)
def test_gradsync_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == 'ok' for r in res), res
