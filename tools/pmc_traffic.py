"""Summarise rocprofv3 PMC passes of the DEFAULT bench command (`python bench.py`, 3 batches in flight) into profiles/<name>.json.

    python tools/pmc_traffic.py OUT.json BENCH_LINE.json fetch_dir write_dir [valu_dir]

fetch_dir / write_dir: output directories of `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py ...` and
`--pmc WRITE_SIZE ...` (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes); valu_dir (optional):
`--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE`. BENCH_LINE.json: the JSON line bench.py printed under one of the passes (workload id, timed steps,
bases of the warm-up batches and of the timed batches). FETCH_SIZE / WRITE_SIZE are reported in KiB; the values written are counters x 1024,
uncorrected. Counter collection serialises the dispatches, so the bytes are those of the kernels themselves, whatever ran beside them
in the timed run. The run also executes warm-up batches (every context aligns the longest batch once); "per step" = total x
timed bases / (timed + warm-up bases) / steps — traffic and instruction counts scale with the bases processed.
"""
import collections, csv, glob, json, sqlite3, sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            k = r['Kernel_Name'].split('(')[0]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
        return agg, n
    db = glob.glob(d + '/**/*.db', recursive=True)[0]           # rocpd output (ROCm 7 default)
    c = sqlite3.connect(db)
    for name, counter, value in c.execute('select name, counter_name, value from counters_collection'):
        k = name.split('(')[0]
        agg[k][counter] += float(value); n[k][counter] += 1
    return agg, n


out = sys.argv[1]
bench = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
fa, fn = load(sys.argv[3]); wa, wn = load(sys.argv[4])
va = load(sys.argv[5])[0] if len(sys.argv) > 5 and sys.argv[5] != '-' else {}
steps = bench['steps']
timed_bases = bench['timed_bases']; warm_bases = bench['warmup_bases']
scale = timed_bases / float(timed_bases + warm_bases) / steps            # total over the run -> per timed step
VALU_PEAK_PER_US_PER_SIMD = 574.0      # profiles/r02_valu_calibration.md: packed-int16 / DPP / v_max wave-instructions per microsecond per SIMD (4.18 cycles each at 2.4 GHz)
def once(k):      # kernels of the index build and the read upload: they run once per process, not per step
    return k.startswith('k_ref_sketch') or k.startswith('k_idx_') or 'rocprim' in k or k.startswith('k_encode')


tot_valu = sum(v.get('SQ_INSTS_VALU', 0.0) for k, v in va.items() if not once(k)) if va else 0.0
res = {'workload_id': bench['config']['workload_id'], 'steps': steps, 'warmup_batches': bench.get('warmup_batches'), 'per_step_scale': scale,
       'valu_wave_insts_per_step': (tot_valu * scale) if tot_valu else None, 'valu_peak_wave_insts_per_s': VALU_PEAK_PER_US_PER_SIMD * 1e6 * 1024,
       'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE (separate passes, --kernel-trace only) of the default '
                 'bench command (%d timed steps of %d reads, %s), gfx950' % (steps, bench['config']['reads_per_step_per_gpu'], bench['config']['schedule']),
       'unit_note': 'rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane) streaming reads '
                    'by 2x, other widths and WRITE_SIZE are uncalibrated. Values below are the raw counters x 1024.',
       'kernels': {}}
for k in sorted(fa, key=lambda k: -(fa[k]['FETCH_SIZE'] + wa.get(k, {}).get('WRITE_SIZE', 0))):
    f = fa[k]['FETCH_SIZE'] * 1024; w = wa.get(k, {}).get('WRITE_SIZE', 0) * 1024; ln = fn[k]['FETCH_SIZE']
    e = {'launches': ln, 'fetch_size_bytes_per_step': f * scale, 'write_size_bytes_per_step': w * scale, 'hbm_bytes_per_step': (f + w) * scale}
    if k in va and va[k].get('GRBM_GUI_ACTIVE'):
        # SQ_INSTS_VALU counts wave instructions; the packed-int16 / DPP instructions of the DP kernels take 4 cycles per SIMD (calibrated:
        # profiles/r02_valu_calibration.md, where this same formula reads 0.96 on saturated loops); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        util = (va[k]['SQ_INSTS_VALU'] / 1024 * 4) / (va[k]['GRBM_GUI_ACTIVE'] / 8)
        e['valu'] = {'wave_insts_per_step': va[k]['SQ_INSTS_VALU'] * scale, 'busy_cycles_per_step_per_xcd': va[k]['GRBM_GUI_ACTIVE'] / 8 * scale,
                     'utilisation_4cycle_model': util, 'frac_of_calibrated_peak': util / 0.96,
                     'peak_wave_insts_per_s': VALU_PEAK_PER_US_PER_SIMD * 1e6 * 1024, 'cycles_per_wave_inst': 4.18,
                     'note': 'time-weighted over every launch of the kernel incl. the low-occupancy redo launches; calibration in profiles/r02_valu_calibration.md'}
    if once(k):
        res.setdefault('once_per_process', {})[k.split('<')[0][:60]] = {'launches': ln, 'hbm_bytes_total': f + w}
        continue
    if (f + w) * scale > 16e6 or k in ('k_gapfill_fill_ns', 'k_local_seed', 'k_local_seed_band', 'k_cluster_big', 'k_cluster_long', 'k_cluster_gen', 'k_cluster'):
        res['kernels'][k] = e
res['total_hbm_bytes_per_step'] = sum((fa[k]['FETCH_SIZE'] + wa.get(k, {}).get('WRITE_SIZE', 0)) for k in fa if not once(k)) * 1024 * scale
json.dump(res, open(out, 'w'), indent=1)
print(out, {k: round(v['hbm_bytes_per_step'] / 1e9, 2) for k, v in list(res['kernels'].items())[:8]})
