"""Aggregate rocprofv3 --pmc passes (one directory per pass) of tools/micro_split_one.py into one CSV: average counter value per
launch for every kernel whose name contains one of the given substrings.
  python tools/sq_counters.py out.csv "label" dir_pass1 dir_pass2 -- k32_kernel wgrad_k32 ...
Derived rows: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); effective clock =
GRBM_GUI_ACTIVE / 8 / kernel duration (from the kernel trace of the same pass)."""
import csv, glob, os, sys
from collections import defaultdict
out, label = sys.argv[1], sys.argv[2]
sep = sys.argv.index('--')
dirs, keys = sys.argv[3:sep], sys.argv[sep + 1:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((key for key in keys if key in row['Kernel_Name']), None)
            if k is None:
                continue
            a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((key for key in keys if key in row['Kernel_Name']), None)
            if k is not None:
                dur[k][0] += 1; dur[k][1] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
counters = sorted({c for k in acc for c in acc[k]})
w = csv.writer(open(out, 'w'))
w.writerow(['counter (average per launch; %s; rocprofv3 --pmc, separate passes)' % label] + keys)
for c in counters:
    w.writerow([c] + [round(acc[k][c][1] / acc[k][c][0]) if c in acc[k] else '' for k in keys])
def get(k, c):
    return acc[k][c][1] / acc[k][c][0] if c in acc[k] else None
row_busy, row_clk, row_us, row_wait, row_valu = ['MFMA pipe busy (derived)'], ['effective clock GHz (derived)'], ['kernel duration us (profiled pass)'], ['waves parked: SQ_WAIT_ANY / SQ_WAVE_CYCLES'], ['non-MFMA vector instructions per MFMA']
for k in keys:
    g, b = get(k, 'GRBM_GUI_ACTIVE'), get(k, 'SQ_VALU_MFMA_BUSY_CYCLES')
    us = dur[k][1] / dur[k][0] / 1e3 if dur[k][0] else None
    row_busy.append(round(b / (g / 8 * 1024), 4) if g and b else '')
    row_clk.append(round(g / 8 / (us * 1e3), 3) if g and us else '')
    row_us.append(round(us, 1) if us else '')
    wa, wc = get(k, 'SQ_WAIT_ANY'), get(k, 'SQ_WAVE_CYCLES')
    row_wait.append(round(wa / wc, 4) if wa and wc else '')
    iv, im = get(k, 'SQ_INSTS_VALU'), get(k, 'SQ_INSTS_MFMA')
    row_valu.append(round((iv - im) / im, 3) if iv and im else '')
for r in (row_busy, row_clk, row_us, row_wait, row_valu):
    w.writerow(r)
print(open(out).read())
