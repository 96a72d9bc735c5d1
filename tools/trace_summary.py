"""Summarise a rocprofv3 --kernel-trace CSV per batch (between k_sketch launches): wall span and per-kernel total durations."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(r['Kernel_Name'].split('(')[0], int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Grid_Size_X']) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if r[0] in ('k_sketch', 'k_sketch32')] + [len(rows)]
for b in range(len(idx) - 1):
    seg = rows[idx[b]:idx[b + 1]]; t0 = seg[0][1]; tot = {}
    for r in seg: tot[r[0]] = tot.get(r[0], 0) + r[2]
    print('batch', b, 'span %.1f' % ((seg[-1][1] + seg[-1][2] - t0) / 1e6), ' '.join('%s=%.1f' % (k[2:], v / 1e6) for k, v in sorted(tot.items(), key=lambda x: -x[1])[:10]))
    if len(sys.argv) > 2 and b == int(sys.argv[2]):
        for r in seg:
            if r[2] > 400000 or r[0] in ('k_ext_phase', 'k_ext_records'): print('   %-22s start %7.2f dur %6.2f ms grid=%s' % (r[0], (r[1] - t0) / 1e6, r[2] / 1e6, r[3]))
# timed batches only (skip the first: warm-up): per-kernel total over them, ms per batch
tb = range(1, len(idx) - 1); tot = {}; span = 0.0
for b in tb:
    seg = rows[idx[b]:idx[b + 1]]; span += (seg[-1][1] + seg[-1][2] - seg[0][1]) / 1e6
    for r in seg: tot[r[0]] = tot.get(r[0], 0) + r[2]
nb = max(1, len(tb))
print('timed batches: %d, mean span %.1f ms' % (nb, span / nb))
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:24]: print('  %-26s %7.2f ms/batch' % (k, v / 1e6 / nb))
