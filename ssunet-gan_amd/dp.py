"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI.

Replaces the reference's nn.DataParallel wrapping (train_seg_gan.py:480-481) and its unwired
thread-based SyncBN machinery (batchnorm.py + comm.py + replicate.py):
  * GradSync      -- gradients live in flat per-bucket buffers (p.grad are views); as soon as a
                     bucket's last gradient has been accumulated during backward, its all-reduce
                     is issued asynchronously (RCCL runs it on its own stream, overlapping the
                     rest of backward); finish() fences and leaves the AVERAGED gradient in place.
  * sync batch norm -- ops.batch_norm_act all-reduces the per-channel [sum, sumsq] (forward) and
                     [sum g, sum g*xhat] (backward) fp64 vectors between its two kernel stages
                     when a BatchNorm carries a `_ssg_sync_group` (convert_sync_batchnorm).
  * metrics       -- IoU/Dice are ratios of whole-batch sums (metrics.py:19-21,32-35): the five
                     partial sums are all-reduced, not the ratios.
No per-forward parameter broadcast: replicas stay identical by identical init + identical
(all-reduced) gradients + identical optimizer arithmetic; broadcast_parameters() is run once.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

BUCKET_BYTES = 64 << 20       # xGMI is point-to-point: few, large messages keep every link busy


# SSG_DIST_FORCE=1: treat a world of ONE rank as distributed too, so that every collective of the data-parallel path
# (bucketed all-reduce, sync-BN sums, metric sums) really goes through the backend.  A 1-GPU box can run RCCL only at
# world size 1 (two ranks on one device are refused), so this is how the RCCL calls are rehearsed there.
FORCE = os.environ.get('SSG_DIST_FORCE', '0') == '1'


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # read by the HSA runtime when it initialises (first HIP call below): dmabuf IPC, which RCCL needs on this driver.
    # A launcher normally exports it already; this covers a bare `python script.py` rank.
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if (world > 1 or FORCE) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # RCCL ('nccl') on GPUs; SSG_DIST_BACKEND=gloo lets several ranks share one GPU in rehearsals
            backend = os.environ.get('SSG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def convert_sync_batchnorm(module, group=None):
    """Mark every BatchNorm in `module` as synchronized over `group` (default: WORLD).
    Arithmetic follows the reference's sync branch (batchnorm.py:115-127, clamp(var, eps)^-1/2)."""
    if not is_dist():
        return module
    group = group if group is not None else dist.group.WORLD
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m._ssg_sync_group = group
    return module


def broadcast_parameters(module, src=0, group=None):
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    from . import ops
    ops.bump_weight_epoch()          # written through .data: version counters did not move, packed/folded caches must


class _Bucket(object):
    __slots__ = ('flat', 'params', 'views', 'pending', 'work')


class GradSync(object):
    """Bucketed, backward-overlapped gradient averaging for one module."""

    def __init__(self, module, group=None, bucket_bytes=BUCKET_BYTES):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.reduce = is_dist()
        self.active = False
        self.buckets = []
        self._where = {}
        params = [p for p in module.parameters() if p.requires_grad]
        cur, cur_bytes = [], 0
        for p in reversed(params):                      # ~ the order backward produces gradients
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        for p in params:
            p.register_post_accumulate_grad_hook(self._hook)
        backend = dist.get_backend(group) if self.reduce else None
        self._avg_native = backend == 'nccl'

    def _make_bucket(self, params):
        b = _Bucket()
        b.params = list(params)
        n = sum((p.numel() + 3) // 4 * 4 for p in params)          # keep every view 16-byte aligned
        b.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        b.views, off = [], 0
        for p in params:
            b.views.append(b.flat[off:off + p.numel()].view_as(p))
            off += (p.numel() + 3) // 4 * 4
        b.pending, b.work = 0, None
        for p in params:
            self._where[p] = len(self.buckets)
        self.buckets.append(b)

    def begin(self):
        """Call after optimizer.zero_grad() and before backward()."""
        for b in self.buckets:
            b.flat.zero_()
            for p, v in zip(b.params, b.views):
                p.grad = v
            b.pending, b.work = len(b.params), None
        self.active = True

    def _launch(self, b):
        if not self.reduce:
            return
        if self._avg_native:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p):
        if not self.active:
            return
        b = self.buckets[self._where[p]]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def finish(self):
        """Call after backward(): flush incomplete buckets, fence, leave averaged grads in p.grad."""
        self.active = False
        for b in self.buckets:
            if b.pending > 0 and b.work is None:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if not self._avg_native:
                    b.flat.div_(self.world)
                b.work = None


_SYNCS = {}


def grad_syncs(generator, discriminator):
    """(GradSync for G, GradSync for D) when running distributed, else (None, None)."""
    if not is_dist():
        return None, None
    key = (id(generator), id(discriminator))
    if key not in _SYNCS:
        _SYNCS[key] = (GradSync(generator), GradSync(discriminator))
    return _SYNCS[key]


def grad_sync(model):
    """One cached GradSync per module (stage-1 trainer: train() is called once per epoch; a fresh GradSync per call would
    leak a model-sized bucket set and one dead hook per parameter each epoch)."""
    if not is_dist():
        return None
    key = id(model)
    hit = _SYNCS.get(key)
    if hit is None or hit[0]() is not model:
        import weakref
        hit = (weakref.ref(model), GradSync(model))
        _SYNCS[key] = hit
    return hit[1]


def reduce_metric_sums(sums, group=None):
    """sums = fp64[5]: (|pred&tgt|, |pred|tgt|, sum p*t, sum p, sum t) over this rank's batch.
    Returns (iou, dice) over the GLOBAL batch."""
    sums = sums.clone()
    if is_dist():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    iou = (sums[0] + 1e-5) / (sums[1] + 1e-5)
    dice = (2.0 * sums[2] + 1e-5) / (sums[3] + sums[4] + 1e-5)
    return iou, dice


def reduce_mean(value, group=None):
    """Average a scalar tensor over ranks (per-rank mean losses over equal local batches)."""
    if not is_dist():
        return value.detach()
    v = value.detach().clone().double()
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v / dist.get_world_size(group)
