"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) into HBM bytes per launch per kernel.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  python tools/pmc_traffic.py out/fetch out/write > profiles/rNN_pmc_hbm_traffic_per_kernel.csv

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is doubled.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def read(directory, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under %s' % directory)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            a = acc[row['Kernel_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
    return acc


def write_meta(path):
    """Sidecar <summary>.meta.json: sha1 of the kernel sources the counters were taken on (bench.py compares it with the
    sources it runs: `roofline.traffic_stale`) and, when a git checkout is at hand, the commit."""
    import hashlib
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    d = os.path.join(root, 'ssunet-gan_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    meta = {'csrc_sha1': h.hexdigest()[:12], 'git_commit': None}
    try:
        meta['git_commit'] = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    json.dump(meta, open(path, 'w'))


def main():
    if len(sys.argv) > 3:                 # python tools/pmc_traffic.py fetch_dir write_dir summary.csv  -> also writes summary.meta.json
        sys.stdout = open(sys.argv[3], 'w')
        write_meta(sys.argv[3][:-4] + '.meta.json')
    fetch = read(sys.argv[1], 'FETCH_SIZE')
    write = read(sys.argv[2], 'WRITE_SIZE')
    out = csv.writer(sys.stdout)
    out.writerow(['kernel', 'dispatches', 'FETCH_SIZE_avg_KB_raw', 'WRITE_SIZE_avg_KB', 'hbm_bytes_per_launch_corrected'])
    rows = []
    for k, (n, tot) in fetch.items():
        f = tot / n
        wn, wt = write.get(k, (0, 0.0))
        w = wt / wn if wn else 0.0
        rows.append((n * (2 * f + w), k, n, f, w))
    for _, k, n, f, w in sorted(rows, reverse=True):
        out.writerow([k, n, '%.1f' % f, '%.1f' % w, int((2 * f + w) * 1024)])


if __name__ == '__main__':
    main()
