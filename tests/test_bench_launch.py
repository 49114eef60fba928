"""CPU rehearsal of `bench.py --gpus N`: the parent must start N ranks itself (torch.distributed.run as a child process),
the ranks must form a world of N, and a WORLD_SIZE / --gpus mismatch must fail loudly instead of silently measuring one GPU
(bench.py:111).  --launch-check runs the launcher, rendezvous and rank bookkeeping without the GPU step."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra, timeout=170):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(180)
def test_bench_gpus2_starts_two_ranks_over_gloo():
    r = _run(['--gpus', '2', '--launch-check'], {'SSG_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['backend'] == 'gloo' and line['rank_sum'] == 3.0


@pytest.mark.timeout(240)
def test_bench_gpus8_starts_eight_ranks_over_gloo():
    """The N = 8 shape of the driver's scaling run (BASELINE config 3), rehearsed on CPU: eight ranks, one process group,
    one JSON line from rank 0, a destroyed group on every rank (no 'destroy_process_group() was not called' warning)."""
    r = _run(['--gpus', '8', '--launch-check'], {'SSG_DIST_BACKEND': 'gloo', 'OMP_NUM_THREADS': '1'}, timeout=230)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['backend'] == 'gloo' and line['rank_sum'] == 36.0
    assert 'destroy_process_group() was not called' not in r.stderr


@pytest.mark.timeout(60)
def test_bench_refuses_to_self_launch_under_a_profiler_preload():
    """under rocprofv3 the parent has already initialised the GPU; starting the launcher from it is the forbidden
    exec hop.  bench.py must refuse (exit 2) instead."""
    r = _run(['--gpus', '2', '--launch-check'], {'SSG_DIST_BACKEND': 'gloo', 'ROCPROF_KERNEL_TRACE': '1'}      )
    assert r.returncode == 2 and 'refusing to self-launch' in r.stderr


@pytest.mark.timeout(60)
def test_bench_refuses_world_size_mismatch():
    r = _run(['--gpus', '2', '--launch-check'], {'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode == 2 and 'WORLD_SIZE=1 but --gpus 2' in r.stderr


def test_pmc_traffic_rows_cover_the_mfma_kernels_of_the_tracked_trace():
    """bench.py reads roofline.traffic from the newest tracked PMC summary by kernel symbol.  Every MFMA conv / weight-gradient
    kernel that the newest tracked kernel trace of the bench shows must have a row there: a renamed or new kernel fails
    here instead of silently printing a stale (or no) number."""
    import csv
    import glob
    sys.path.insert(0, ROOT)
    import bench
    src, table = bench.pmc_traffic_table()
    assert src and table, 'no profiles/rNN_*pmc_hbm_traffic_per_kernel.csv'
    traces = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*kernel_stats_bench_bs16_512.csv')))     # traces of bench.py itself
    assert traces
    assert os.path.basename(traces[-1])[:3] == os.path.basename(src)[:3], 'PMC summary %s is older than the kernel trace %s' % (src, traces[-1])
    names = [bench._norm_symbol(r['Name']) for r in csv.DictReader(open(traces[-1]))]
    mfma = [n for n in names if n.startswith(('conv_igemm', 'wgrad_halo', 'wgrad_dma', 'wgrad_kernel', 'gemm_bf16'))]
    assert mfma
    missing = [n for n in mfma if n not in table]
    assert not missing, 'no PMC traffic row in %s for %s' % (src, missing)
