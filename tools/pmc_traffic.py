"""Summarise rocprofv3 PMC passes of `bench.py --steps K --warmup 1 --streams 1` into profiles/<name>.json.

    python tools/pmc_traffic.py OUT.json STEPS fetch_dir write_dir [valu_dir] [workload_id]

fetch_dir / write_dir: output directories of `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and `--pmc WRITE_SIZE ...` (separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes); valu_dir (optional): `--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE`. FETCH_SIZE / WRITE_SIZE
are reported in KiB; the values written are counters x 1024, uncorrected. Per step = total over the run / (STEPS + 1 warm-up batch).
"""
import collections, csv, glob, json, sys


def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    return agg, n


out, steps = sys.argv[1], int(sys.argv[2])
fa, fn = load(sys.argv[3]); wa, wn = load(sys.argv[4])
va = load(sys.argv[5])[0] if len(sys.argv) > 5 and sys.argv[5] != '-' else {}
workload_id = sys.argv[6] if len(sys.argv) > 6 else None
VALU_PEAK_PER_US_PER_SIMD = 574.0      # profiles/r02_valu_calibration.md: packed-int16 / DPP / v_max wave-instructions per microsecond per SIMD (4.18 cycles each at 2.4 GHz)
batches = steps + 1
res = {'workload_id': workload_id, 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps %d --warmup 1 --streams 1 '
                 '(%d batches of 4096 reads), gfx950' % (steps, batches),
       'unit_note': 'rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane) streaming reads '
                    'by 2x, other widths and WRITE_SIZE are uncalibrated. Values below are the raw counters x 1024.',
       'kernels': {}}
for k in sorted(fa, key=lambda k: -(fa[k]['FETCH_SIZE'] + wa.get(k, {}).get('WRITE_SIZE', 0))):
    f = fa[k]['FETCH_SIZE'] * 1024; w = wa.get(k, {}).get('WRITE_SIZE', 0) * 1024; ln = fn[k]['FETCH_SIZE']
    e = {'launches': ln, 'fetch_size_bytes_per_launch': f / ln, 'write_size_bytes_per_launch': w / ln, 'hbm_bytes_per_launch': (f + w) / ln,
         'hbm_bytes_per_step': (f + w) / batches}
    if k in va and va[k].get('GRBM_GUI_ACTIVE'):
        # SQ_INSTS_VALU counts wave instructions; the packed-int16 / DPP instructions of the DP kernels take 4 cycles per SIMD (calibrated:
        # profiles/r02_valu_calibration.md, where this same formula reads 0.96 on saturated loops); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        e['valu_insts'] = va[k]['SQ_INSTS_VALU']; e['valu_utilisation'] = (va[k]['SQ_INSTS_VALU'] / 1024 * 4) / (va[k]['GRBM_GUI_ACTIVE'] / 8)
        e['valu'] = {'wave_insts_per_step': va[k]['SQ_INSTS_VALU'] / batches, 'busy_cycles_per_step_per_xcd': va[k]['GRBM_GUI_ACTIVE'] / 8 / batches,
                     'frac_of_calibrated_peak': e['valu_utilisation'] / 0.96,
                     'peak_wave_insts_per_s': VALU_PEAK_PER_US_PER_SIMD * 1e6 * 1024, 'cycles_per_wave_inst': 4.18,
                     'note': 'time-weighted over every launch of the kernel incl. the low-occupancy redo launches; calibration in profiles/r02_valu_calibration.md'}
    if f + w > 64e6:
        res['kernels'][k] = e
json.dump(res, open(out, 'w'), indent=1)
print(out, {k: round(v['hbm_bytes_per_step'] / 1e9, 2) for k, v in list(res['kernels'].items())[:8]})
