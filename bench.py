#!/usr/bin/env python3
"""Headline benchmark: train images/sec of the ssUnet-GAN G+D step on synthetic 3x512x512 tiles.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: RANK/WORLD_SIZE are in the
environment) or, when they are not there, this process starts them itself: the parent never touches HIP, runs
torch.distributed.run as a CHILD process, relays rank 0's JSON line and exits with the child's status.  A rank whose
WORLD_SIZE differs from --gpus, or whose backend is not RCCL ("nccl") without SSG_DIST_BACKEND saying so, exits non-zero.

One "step" = one pass of the hot path (train_seg_gan.py:182-233: G fwd+bwd, 3x D fwd+bwd, two
clip+Adam updates) over one batch of 16 tiles per GPU already resident in HBM.  Prints ONE JSON
line (rank 0) with the whole-job images/sec, the MFMA roofline of the dominant kernel measured
with HIP events on the launch stream during the timed steps, and (N=1 only) a bounded CPU
baseline: the oracle's plain-torch step timed on this box's host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist
import torch.nn as nn

# 2*(3*208.625 + 9*24.631) GFLOP per image per step.  With the opt-in SSG_ELIDE_DEAD_D_GRADS=1
# (ssunet-gan_amd/train_seg_gan.py: the discriminator parameter gradients of the G-step backward, which the reference
# zeroes unread at train_seg_gan.py:225, are not computed) 8 instead of 9 discriminator passes are EXECUTED.
D_PASSES = 8 if os.environ.get('SSG_ELIDE_DEAD_D_GRADS', '0') == '1' else 9
FLOP_PER_IMG_512 = 2 * (3 * 208.625 + D_PASSES * 24.631) * 1e9
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6      # dense bf16: 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz (the guide's "~2.5 PF dense")
SPLIT_TERMS = 6                     # bf16 MFMAs a split-operand kernel executes per algorithmic fp32 MFMA step (conv_igemm_halo_x3.hip)


def is_split_kernel(label):
    """Kernels that multiply fp32 operands as three bf16 terms on the bf16 matrix pipe: the 16-channel-chunk x3 family
    (v_mfma_f32_32x32x16_bf16) and the round-4 k32 family (conv_igemm_halo_k32.hip / conv_wgrad_k32.hip, v_mfma_f32_16x16x32_bf16)."""
    return '_x3_' in label or 'k32' in label



def _norm_symbol(name):
    """'void (anonymous namespace)::conv_igemm_halo_kernel<128, 128, 2, 2>(ConvArgs)' -> 'conv_igemm_halo_kernel<128,128,2,2>'."""
    name = name.split('::')[-1].split('(')[0]
    return name.replace(' ', '')


def pmc_traffic_table():
    """HBM bytes per launch per kernel symbol from the newest tracked PMC summary (profiles/rNN_*pmc_hbm_traffic_per_kernel.csv:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950; aggregated by tools/pmc_traffic.py).  bench.py cannot run the profiler on itself, so the table is read from the
    tracked file by kernel symbol; tests/test_host_logic.py fails when a dispatchable MFMA kernel has no row."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*pmc_hbm_traffic_per_kernel.csv')))
    if not files:
        return None, {}
    table = {}
    for row in csv.DictReader(open(files[-1])):
        table[_norm_symbol(row['kernel'])] = int(row['hbm_bytes_per_launch_corrected'])
    return os.path.relpath(files[-1], ROOT), table


def csrc_sha1():
    """sha1 over the kernel sources (csrc/*.hip, *.h, sorted): what a PMC summary was taken on vs what runs now."""
    import glob
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'ssunet-gan_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:12]


def pmc_traffic_meta(src):
    """Sidecar written by tools/pmc_traffic.py next to the summary: {'csrc_sha1': ..., 'git_commit': ...} (None for old files)."""
    if not src:
        return {}
    f = os.path.join(ROOT, src[:-4] + '.meta.json')
    try:
        return json.load(open(f))
    except Exception:
        return {}


# the 12 forward 3x3 convs of the six encoder BasicBlocks, 54.7 GMAC per 512^2 image
ENCODER_3X3_GMAC_PER_IMG_512 = 54.65


def pmc_traffic_for(label, table):
    """`label` is ops._CONV_LABELS / _WGRAD_LABELS style ('conv_igemm_halo_kernel<128,128>'): match the symbol whose template
    argument list starts with the label's."""
    if label in table:
        return table[label]
    stem = label.rstrip('>')
    for k, v in table.items():
        if k.startswith(stem + ',') or k.startswith(stem + '>'):
            return v
    return None


def measured_ceilings(S, dev):
    """fp32-MFMA issue peak and HBM copy rate of THIS device (csrc/tools.hip), to read the fractions against."""
    from ssunet_gan_amd._lib import call, ptr, stream_ptr
    scratch = torch.empty(768 * 256, device=dev)
    def timeit(fn, n):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    it = 2000
    ms = timeit(lambda: call('ssg_tool_mfma_peak_f32', ptr(scratch), 768, it, stream_ptr()), 3)
    mfma = 768 * 4 * it * 16 * 4096 / ms / 1e9
    ms = timeit(lambda: call('ssg_tool_mfma_peak_bf16', ptr(scratch), 768, 2 * it, stream_ptr()), 3)
    mfma16 = 768 * 4 * 2 * it * 16 * 32768 / ms / 1e9
    # ... and on random operands: constants cost the multipliers little power, real data pulls the clock down --
    # long enough (~0.25 s per launch) for the clock to settle
    data = torch.randn(4096 * 8, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: call('ssg_tool_mfma_peak_bf16_data', ptr(scratch), 768, 100 * it, ptr(data), stream_ptr()), 2)
    mfma16r = 768 * 4 * 100 * it * 16 * 32768 / ms / 1e9
    ms = timeit(lambda: call('ssg_tool_mfma_peak_bf16_data16', ptr(scratch), 768, 100 * it, ptr(data), stream_ptr()), 2)
    mfma16r16 = 768 * 4 * 100 * it * 16 * 32768 / ms / 1e9
    a = torch.empty(1 << 28, device=dev); b = torch.empty(1 << 28, device=dev)      # 1 GiB each: far beyond the 256-MB Infinity Cache
    ms = timeit(lambda: call('ssg_tool_copy_f32', ptr(a), ptr(b), a.numel(), stream_ptr()), 3)
    return {'mfma_f32_tflops': round(mfma, 1), 'mfma_bf16_tflops': round(mfma16, 1), 'mfma_bf16_random_operand_tflops': round(mfma16r, 1),
            'mfma_bf16_16x16x32_random_operand_tflops': round(mfma16r16, 1), 'hbm_copy_tbps': round(2 * a.numel() * 4 / ms / 1e9, 2),
            'hbm_copy_form': 'one float4 per thread, non-temporal, 1 GiB -> 1 GiB (the grid-stride probe of rounds 1-3 read 4.6-4.8)'}


def _under_profiler():
    preload = ' '.join(os.environ.get(k, '') for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'))
    return 'rocprof' in preload.lower() or any(k.startswith('ROCPROFILER_') or k.startswith('ROCPROF_') for k in os.environ)


def side_configs(S, dev, note):
    """Short legs for the other BASELINE configs, so that the driver's own run carries them; ~20 s in all.
    config 4: EfficientNet-B4 extract_features fwd+bwd, 4 x 3x1024x1024, bf16 -- eager and as a hipGraph replay (tools/bench_b4.py is
    the long form with the per-kernel table).  config 5: eval-mode generator over the 36 patches of a 2048^2 image, batch 12 and the
    reference's batch 1 (tools/bench_extra.py).  dp_costs: a CHILD run of this script with SSG_DIST_FORCE=1 (world 1, every
    collective of the data-parallel path through RCCL) -- what the sync-BN and gradient-bucket all-reduces cost on this box."""
    import types
    out = {}
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import bench_b4
        a4 = types.SimpleNamespace(batch=4, size=1024, steps=4, warmup=2)
        dte, _ = bench_b4.run(torch.bfloat16, a4, dev, False)
        try:
            dtg, ok = bench_b4.run_graph(torch.bfloat16, a4, dev)
            graph = {'images_per_s': round(4 / dtg, 2), 'ms_per_step': round(dtg * 1e3, 2), 'finite_grads': bool(ok)}
        except Exception as e:
            graph = {'error': type(e).__name__}
        out['config4_b4_bf16'] = {'workload': 'EfficientNet-B4 extract_features fwd+bwd, 4 x 3x1024x1024, bf16 tensors, train mode', 'steps': 4,
                                  'eager': {'images_per_s': round(4 / dte, 2), 'ms_per_step': round(dte * 1e3, 2)}, 'hipgraph_replay': graph}
        note('config 4: bf16 B4 %.1f img/s eager' % (4 / dte))
    except Exception as e:                       # a side leg must never cost the headline line
        out['config4_b4_bf16'] = {'error': '%s: %s' % (type(e).__name__, e)}
    try:
        torch.manual_seed(41)
        G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False)).to(dev)
        patches = torch.randn(36, 3, 512, 512, generator=torch.Generator().manual_seed(7))
        c5 = {'workload': 'eval-mode G (BN folded) over the 36 patches (3x512x512) of a 2048^2 image, incl. H2D / D2H'}
        for bs in (12, 1):
            S.aerial_image_segmentation_api.infer_patches(G, patches, batch_size=bs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                S.aerial_image_segmentation_api.infer_patches(G, patches, batch_size=bs)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
            c5['batch_%d' % bs] = {'patches_per_s': round(36 / dt, 1), 's_per_image': round(dt, 3)}
        out['config5_infer'] = c5
        del G
        torch.cuda.empty_cache()
        note('config 5: %.0f patches/s batched by 12' % c5['batch_12']['patches_per_s'])
    except Exception as e:
        out['config5_infer'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if os.environ.get('SSG_DIST_FORCE') != '1' and not _under_profiler():
        try:
            env = dict(os.environ, SSG_DIST_FORCE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-fp32-reference',
                                '--no-side-configs'], env=env, capture_output=True, text=True, timeout=240)
            line = next((json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')), None)
            if line is None:
                out['dp_costs'] = {'error': 'child exited %d: %s' % (r.returncode, r.stderr[-300:])}
            else:
                out['dp_costs'] = {'how': 'child run with SSG_DIST_FORCE=1: world 1, sync-BN statistics, gradient buckets and metric sums through RCCL',
                                   'backend': line.get('backend'), 'ms_per_step': line.get('ms_per_step'),
                                   'sync_bn_allreduce': line.get('sync_bn_allreduce'), 'sync_bn_allreduce_ms': line.get('sync_bn_allreduce_ms'),
                                   'grad_bucket_allreduce': line.get('grad_bucket_allreduce')}
                note('dp costs (world 1 over RCCL): sync-BN all-reduces %.2f ms/step' % (line.get('sync_bn_allreduce_ms') or float('nan')))
        except Exception as e:
            out['dp_costs'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def host_cores():
    """Threads this process may really use: the cgroup CPU quota if there is one (a GPU box gives a
    1-GPU job a share of its host, although affinity still lists every core), else the affinity
    mask; capped at 32 -- beyond that a 1-image step only loses time to oversubscription."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get('SSG_CPU_BASELINE_THREADS')
    if env:
        return max(1, int(env))
    return max(1, min(n, 16))          # the pool's per-GPU CPU share is 16 cores


def cpu_model():
    """CPU model string of the host (BASELINE.md 3 asks for it beside the core count)."""
    try:
        for l in open('/proc/cpuinfo'):
            if l.lower().startswith('model name'):
                return l.split(':', 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(seconds_budget=30.0):
    """Oracle (plain torch CPU restatement, pinned to the reference's golden vectors) timed on a
    bounded sample: 1 x 3 x 512 x 512 steps (the GPU workload is 16 such images per step)."""
    from oracle import seg_gan_cpu as O
    cores = host_cores()
    torch.set_num_threads(cores)
    G, D, og, od = O.make_models()
    inp, tgt = O.synthetic_batch(1, 512, 512)
    print('[bench] cpu baseline: oracle step on %d host threads ...' % cores, file=sys.stderr, flush=True)
    t0 = time.time()
    O.gan_step(G, D, og, od, inp, tgt)                      # warm-up (also sizes the sample)
    warm = time.time() - t0
    print('[bench] cpu baseline warm-up step: %.1f s' % warm, file=sys.stderr, flush=True)
    cpu = cpu_model()
    if warm > seconds_budget / 2:                           # slow host: the warm-up step IS the sample
        return {'value': round(1.0 / warm, 4), 'unit': 'images/sec', 'cores': cores, 'cpu_model': cpu, 'kind': 'port',
                'sample': '1 G+D step (first call, no warm-up) on 1x3x512x512, fp32, torch %d threads' % cores}
    n = max(1, min(3, int((seconds_budget - warm) / max(warm, 1e-3))))
    t0 = time.time()
    for _ in range(n):
        O.gan_step(G, D, og, od, inp, tgt)
    dt = (time.time() - t0) / n
    return {'value': round(1.0 / dt, 4), 'unit': 'images/sec', 'cores': cores, 'cpu_model': cpu, 'kind': 'port',
            'sample': '%d timed G+D steps (after 1 warm-up) on 1x3x512x512, fp32, torch %d threads' % (n, cores)}


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(n, argv):
    """Parent of an N-rank run: start `python -m torch.distributed.run` as a CHILD (this process has not touched HIP and
    never will), let the ranks inherit stdout (rank 0 prints the JSON line) and return the child's exit status."""
    # Under rocprofv3 the profiler's preloaded library has ALREADY initialised the GPU in this process, so starting the launcher
    # from here is the fork/exec hop of a GPU-initialised process that this pool forbids (it can take the machine down).
    # Multi-rank profiling wraps each rank's own python process (rocprofv3 ... -- python3 bench.py under torchrun), never this parent.
    preload = ' '.join(os.environ.get(k, '') for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'))
    if 'rocprof' in preload.lower() or any(k.startswith('ROCPROFILER_') or k.startswith('ROCPROF_') for k in os.environ):
        print('[bench] refusing to self-launch %d ranks under a profiler preload (%s): start the ranks with torch.distributed.run '
              'and wrap each rank, e.g. torchrun ... --no-python rocprofv3 ... -- python3 bench.py --gpus %d' % (n, preload.strip() or 'ROCPROF*', n),
              file=sys.stderr)
        return 2
    env = os.environ.copy()
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # read when HSA initialises in the ranks (dmabuf IPC for RCCL)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % n,
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    print('[bench] starting %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def bucket_allreduce_ms(syncs, dev, reps=5):
    """Stand-alone timing of one all-reduce per gradient bucket (the same flat buffers, same op, same communicator) after the
    timed steps: what each bucket costs when nothing overlaps it."""
    out = []
    for name, sync in syncs:
        if sync is None:
            continue
        for i, b in enumerate(sync.buckets):
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM)
            e1.record(); torch.cuda.synchronize()
            out.append({'bucket': '%s%d' % (name, i), 'mbytes': round(b.flat.numel() * 4 / 1e6, 1), 'ms': round(e0.elapsed_time(e1) / reps, 3)})
    return out


def launch_check(args):
    """--launch-check: the launcher / rendezvous / rank bookkeeping of an N-rank run WITHOUT the GPU step (CPU rehearsal over
    gloo: tests/test_bench_launch.py).  Prints the same JSON skeleton with value null."""
    import ssunet_gan_amd as S
    rank, world, local = S.dp.init_from_env()
    if world != args.gpus:
        print('[bench] WORLD_SIZE=%d but --gpus %d' % (world, args.gpus), file=sys.stderr)
        return 2
    backend = dist.get_backend() if world > 1 else None
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'train images/sec (512^2 tiles)', 'value': None, 'unit': 'images/sec', 'n_gpus': world,
                          'launch_check': True, 'backend': backend, 'rank_sum': float(t.item())}))
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='tiles per GPU')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-reference', action='store_true', help='skip the extra steps with every conv on the fp32 MFMA')
    ap.add_argument('--launch-check', action='store_true', help='rendezvous only, no GPU step (CPU rehearsal of the N-rank launch)')
    ap.add_argument('--no-side-configs', action='store_true', help='skip the short legs for BASELINE configs 4 / 5 and the world-1 RCCL cost run')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # Not started by a launcher: start the ranks ourselves.  Nothing above this line has touched HIP
        # (importing torch does not), and the child is a new process, not an exec of this one.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env != args.gpus:
        print('[bench] refusing to run: WORLD_SIZE=%d but --gpus %d (start the ranks with --nproc-per-node %d, or let '
              'bench.py start them by running it without a launcher)' % (world_env, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        sys.exit(launch_check(args))

    import ssunet_gan_amd as S
    rank, world, local = S.dp.init_from_env()
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    dev = torch.device('cuda', torch.cuda.current_device())
    backend = None
    if world > 1:
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus:
            print('[bench] process group has %d ranks, --gpus %d' % (dist.get_world_size(), args.gpus), file=sys.stderr)
            sys.exit(2)
        if backend != 'nccl' and os.environ.get('SSG_DIST_BACKEND') != backend:
            print('[bench] backend is %r, not RCCL ("nccl")' % backend, file=sys.stderr)
            sys.exit(2)
        if backend == 'nccl' and torch.cuda.device_count() < world:
            print('[bench] %d ranks but %d visible GPUs' % (world, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(2)

    torch.manual_seed(41)                                          # train_seg_gan.py:35-36
    G = S.models_seg_gan.Generator(dict(arch='UNet_R_SS_v2', num_classes=3, input_channels=3, deep_supervision=False))
    D = S.models_seg_gan.Discriminator(3, kernel_size=3, n_channels=64, n_blocks=8, fc_size=1024)
    G.to(dev).train(); D.to(dev).train()
    if S.dp.is_dist():                                             # world > 1, or the 1-rank RCCL rehearsal (SSG_DIST_FORCE=1)
        S.dp.broadcast_parameters(G); S.dp.broadcast_parameters(D)
        S.dp.convert_sync_batchnorm(G); S.dp.convert_sync_batchnorm(D)
    og = torch.optim.Adam(params=filter(lambda p: p.requires_grad, G.parameters()), lr=2e-5)
    od = torch.optim.Adam(params=filter(lambda p: p.requires_grad, D.parameters()), lr=2e-5)
    crit, adv, con = S.losses.BCEDiceLoss(), nn.BCEWithLogitsLoss(), nn.MSELoss()
    sync_g, sync_d = S.dp.grad_syncs(G, D)

    gen = torch.Generator().manual_seed(7 + rank)
    inp = torch.randn(args.batch, 3, args.size, args.size, generator=gen).to(dev)
    tgt = (torch.rand(args.batch, 3, args.size, args.size, generator=gen) > 0.5).float().to(dev)

    def step():
        return S.train_seg_gan.gan_step(inp, tgt, G, D, crit, adv, con, og, od, 3, sync_g, sync_d)

    def note(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    note('models on device, starting %d warm-up steps' % args.warmup)
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        note('warm-up %d: %.1f ms' % (i, (time.perf_counter() - t_w) * 1e3))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    S.ops.PROFILE = []; S.ops.PROFILE_HBM = []; S.ops.PROFILE_COMM = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, S.ops.PROFILE = S.ops.PROFILE, None
    prof_hbm, S.ops.PROFILE_HBM = S.ops.PROFILE_HBM, None
    prof_comm, S.ops.PROFILE_COMM = S.ops.PROFILE_COMM, None
    note('timed %d steps: %.1f ms/step' % (args.steps, dt / args.steps * 1e3))
    # the same step with every conv on the fp32 MFMA (SSG_MFMA_SPLIT=0), measured in THIS run: 1 warm-up + a few timed steps
    ref32 = None
    if S.ops.MFMA_SPLIT and not args.no_fp32_reference:
        S.ops.MFMA_SPLIT = False
        try:
            step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            k32 = max(1, min(args.steps, 4))
            t1 = time.perf_counter()
            for _ in range(k32):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([d32], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d32 = float(t.item())
            ref32 = {'ms_per_step': round(d32 / k32 * 1e3, 2), 'value': round(args.batch * world * k32 / d32, 3), 'steps': k32,
                     'step_frac_of_conv_roofline': round(args.batch * k32 / d32 * FLOP_PER_IMG_512 * (args.size / 512.0) ** 2 / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        finally:
            S.ops.MFMA_SPLIT = True
        note('fp32-MFMA reference: %.1f ms/step' % ref32['ms_per_step'])
    devices = [torch.cuda.current_device()]
    buckets = None
    if world == 1 and sync_g is not None:                  # SSG_DIST_FORCE=1: one rank, every collective through the backend
        buckets = bucket_allreduce_ms((('G', sync_g), ('D', sync_d)), dev)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ids = [None] * world
        dist.all_gather_object(ids, (rank, torch.cuda.current_device(), os.environ.get('LOCAL_RANK')))
        devices = [d for _, d, _ in sorted(ids)]
        buckets = bucket_allreduce_ms((('G', sync_g), ('D', sync_d)), dev)

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        # dominant MFMA kernel: aggregate HIP-event durations per kernel symbol
        agg = {}
        enc = [0.0, 0.0, 0]                               # FLOPs, seconds, launches of the forward 3x3 convs of the encoder blocks
        for label, flops, e0, e1, tag in prof:
            a = agg.setdefault(label, [0.0, 0.0, 0])
            sec = e0.elapsed_time(e1) * 1e-3
            a[0] += flops; a[1] += sec; a[2] += 1
            if tag == 'encoder_3x3':
                enc[0] += flops; enc[1] += sec; enc[2] += 1
        hbm = {}
        for stage, nbytes, e0, e1 in prof_hbm:
            a = hbm.setdefault(stage, [0.0, 0.0, 0])
            a[0] += nbytes; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        dom = max(agg.items(), key=lambda kv: kv[1][1]) if agg else None
        conv_t = sum(v[1] for v in agg.values()); conv_f = sum(v[0] for v in agg.values())
        roof = None
        if dom:
            label, (fl, tt, cnt) = dom
            ach = fl / tt / 1e12
            src, table = pmc_traffic_table()
            traffic = pmc_traffic_for(label, table)
            if traffic is None:
                note('WARNING: no PMC traffic row for the dominant kernel %s in %s' % (label, src))
            meta = pmc_traffic_meta(src)
            now = csrc_sha1()
            def kernel_entry(k, v):
                e = {'tflops': round(v[0] / v[1] / 1e12, 2), 'time_frac_of_step': round(v[1] / dt, 4), 'launches': v[2]}
                if is_split_kernel(k):         # fp32 operands split into three bf16 terms: 6 bf16 MFMAs per algorithmic fp32 MFMA step
                    e['pipe'] = 'bf16 (3-term split, fp32 accumulate)'
                    e['executed_mfma_tflops'] = round(SPLIT_TERMS * v[0] / v[1] / 1e12, 1)
                    e['executed_frac_of_bf16_peak'] = round(SPLIT_TERMS * v[0] / v[1] / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
                return e
            # `achieved` = ALGORITHMIC fp32 conv FLOPs / HIP-event time of the dominant kernel.  `peak` = the ceiling of the pipe that
            # kernel multiplies on, in the same algorithmic unit: the fp32 MFMA peak for a v_mfma_f32_32x32x2_f32 kernel; for a
            # split-operand kernel the dense bf16 MFMA peak / 6 (six bf16 MFMAs of the same shape stand for one fp32 product:
            # DESIGN.md 3.9) -- `executed` restates that in executed bf16 FLOP/s, `frac_of_fp32_mfma_peak` against the fp32 pipe.
            split_dom = is_split_kernel(label)
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS if split_dom else PEAK_FP32_MFMA_TFLOPS
            roof = {'bound': 'mfma', 'kernel': label, 'achieved': round(ach, 2), 'peak': round(peak, 1),
                    'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'peak_is': ('dense bf16 MFMA peak %.1f / %d products per fp32 multiply (split-operand kernel)' % (PEAK_BF16_MFMA_TFLOPS, SPLIT_TERMS))
                               if split_dom else 'dense fp32 MFMA peak',
                    'fp32_mfma_peak': PEAK_FP32_MFMA_TFLOPS, 'frac_of_fp32_mfma_peak': round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    'traffic': traffic, 'traffic_source': src, 'traffic_git_commit': meta.get('git_commit'),
                    'traffic_csrc_sha1': meta.get('csrc_sha1'), 'csrc_sha1': now,
                    'traffic_stale': (meta.get('csrc_sha1') != now) if meta.get('csrc_sha1') else None,
                    'launches': cnt, 'avg_launch_ms': round(tt / cnt * 1e3, 4),
                    'all_mfma_kernels': {k: kernel_entry(k, v) for k, v in sorted(agg.items())},
                    'mfma_time_frac_of_step': round(conv_t / dt, 4),
                    'step_frac_of_fp32_mfma_roofline': round(value * FLOP_PER_IMG_512 * (args.size / 512.0) ** 2 / 1e12 / PEAK_FP32_MFMA_TFLOPS / world, 4)}
            roof['step_frac_of_conv_roofline'] = roof['step_frac_of_fp32_mfma_roofline']       # the name rounds 1-2 reported
            if split_dom:
                roof['executed'] = {'pipe': 'bf16 MFMA (%s), fp32 operands split into 3 bf16 terms, 6 of the 9 products, fp32 accumulate' % ('v_mfma_f32_16x16x32_bf16' if 'k32' in label else 'v_mfma_f32_32x32x16_bf16'),
                                    'achieved': round(SPLIT_TERMS * ach, 1), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': round(SPLIT_TERMS * ach / PEAK_BF16_MFMA_TFLOPS, 4)}
        line = {
            'metric': 'train images/sec (512^2 tiles)', 'value': round(value, 3), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'matrix_pipe': ('fp32 tensors and fp32 accumulation everywhere; the dense 3x3 unit-stride convs and input gradients multiply on the bf16 pipe with each '
                            'fp32 operand split into 3 bf16 terms (error vs fp64 = that of the fp32-MFMA kernel: tests/test_split_gpu.py; SSG_MFMA_SPLIT=0 '
                            'runs them on v_mfma_f32_32x32x2_f32: `fp32_mfma_reference`)') if S.ops.MFMA_SPLIT else 'v_mfma_f32_32x32x2_f32 (fp32 MFMA) for every conv',
            'config': {'workload': 'UNet_R_SS_v2 generator + SRGAN-style discriminator, one G+D step (train_seg_gan.py:182-233), '
                                   '%d x 3x%dx%d tiles per GPU, fp32, Adam lr 2e-5' % (args.batch, args.size, args.size),
                       'global_batch': args.batch * world, 'tile': args.size,
                       'parallelism': 'dp%d%s' % (world, ' (RCCL grad all-reduce + sync-BN)' if S.dp.is_dist() else ''),
                       'dead_d_param_grads_of_g_step': 'computed' if D_PASSES == 9 else 'not computed (zeroed unread in the reference, train_seg_gan.py:225)'},
            'backend': backend if backend else (dist.get_backend() if S.dp.is_dist() else None), 'devices': devices,
            'loss': round(float(out[0]), 6), 'iou': round(float(out[1]), 6), 'dice': round(float(out[2]), 6),
            'roofline': roof,
        }
        # north-star sub-metrics (BASELINE.json): MFMA roofline on the 3x3 encoder convs; achieved HBM rate on the memory-bound
        # stages -- algorithmic bytes of SURVEY.md 8(d) / HIP-event time on the launch stream, against the 8 TB/s spec
        if enc[2]:
            line['encoder_3x3'] = {'flops_per_step': round(enc[0] / args.steps), 'launches_per_step': enc[2] // args.steps,
                                   'ms_per_step': round(enc[1] / args.steps * 1e3, 3), 'tflops': round(enc[0] / enc[1] / 1e12, 2),
                                   'frac': round(enc[0] / enc[1] / 1e12 / (PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS if S.ops.MFMA_SPLIT else PEAK_FP32_MFMA_TFLOPS), 4),
                                   'frac_of_fp32_mfma_peak': round(enc[0] / enc[1] / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   'algorithmic_gmac_per_img': ENCODER_3X3_GMAC_PER_IMG_512 * (args.size / 512.0) ** 2}
        if hbm:
            line['hbm_stages'] = {k: {'gbytes_per_step': round(v[0] / args.steps / 1e9, 3), 'ms_per_step': round(v[1] / args.steps * 1e3, 3),
                                      'launches_per_step': v[2] // args.steps, 'tbps': round(v[0] / v[1] / 1e12, 3),
                                      'frac_of_8tbps': round(v[0] / v[1] / 8e12, 4)} for k, v in sorted(hbm.items())}
        if ref32 is not None:
            line['fp32_mfma_reference'] = ref32
        if prof_comm:
            # the sync-BN statistics all-reduces sit on the compute stream between the two stages of every batch norm: their
            # event time is stream time the step cannot overlap (measured, not estimated)
            comm = {}
            for kind, nbytes, e0, e1 in prof_comm:
                a = comm.setdefault(kind, [0, 0.0, 0])
                a[0] += nbytes; a[1] += e0.elapsed_time(e1); a[2] += 1
            line['sync_bn_allreduce'] = {k: {'per_step': v[2] // args.steps, 'ms_per_step': round(v[1] / args.steps, 3),
                                             'avg_us': round(v[1] / v[2] * 1e3, 1), 'avg_bytes': v[0] // v[2]} for k, v in sorted(comm.items())}
            line['sync_bn_allreduce_ms'] = round(sum(v[1] for v in comm.values()) / args.steps, 3)
        if buckets is not None:
            line['grad_bucket_allreduce'] = buckets
        if world == 1:
            line['measured_ceilings'] = measured_ceilings(S, dev)
            # the split kernels against what the bf16 pipe of THIS device sustains on random operands (power-limited: DESIGN.md 3.9)
            if roof and roof.get('executed') and line['measured_ceilings'].get('mfma_bf16_random_operand_tflops'):
                key = 'mfma_bf16_16x16x32_random_operand_tflops' if 'k32' in roof['kernel'] else 'mfma_bf16_random_operand_tflops'
                roof['executed']['frac_of_measured_random_operand_rate'] = round(roof['executed']['achieved'] / line['measured_ceilings'][key], 4)
        if world == 1 and not args.no_side_configs and os.environ.get('SSG_DIST_FORCE') != '1':
            line.update(side_configs(S, dev, note))
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_available() and dist.is_initialized():          # also the world-1 rehearsal under SSG_DIST_FORCE=1
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
