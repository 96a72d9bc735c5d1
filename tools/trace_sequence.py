import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(r['Kernel_Name'].split('(')[0], int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Grid_Size_X']) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if r[0] in ('k_sketch', 'k_sketch32')] + [len(rows)]
b = int(sys.argv[2]); seg = rows[idx[b]:idx[b + 1]]; t0 = seg[0][1]
for r in seg: print('%8.3f %7.3f %-28s %s' % ((r[1] - t0) / 1e6, r[2] / 1e6, r[0].replace('__amd_rocclr_', 'RT_')[:28], r[3]))
