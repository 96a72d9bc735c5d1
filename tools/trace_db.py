"""Kernel-trace analysis of a rocprofv3 run (rocpd sqlite output, ROCm 7): per-kernel statistics CSV (the --stats table), and the
concurrency of the timed region — how much of the wall time 0, 1, 2, 3+ kernels were running, and which kernels the time was spent under.

    python tools/trace_db.py RESULTS.db [--csv OUT.csv] [--skip N]      N = k_sketch launches before the timed region (default 4)
"""
import collections, csv, sqlite3, sys


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end, stream, grid_x, workgroup_x, lds_size from kernels order by start').fetchall()
    return [(n.split('(')[0], s, e, st, g, w, l) for n, s, e, st, g, w, l in rows]


def main():
    db = sys.argv[1]
    rows = load(db)
    if '--csv' in sys.argv:
        out = sys.argv[sys.argv.index('--csv') + 1]
        agg = collections.OrderedDict()
        for n, s, e, *_ in rows:
            a = agg.setdefault(n, [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
        tot = sum(a[1] for a in agg.values())
        with open(out, 'w', newline='') as f:
            w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
            for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
                w.writerow([n, a[0], a[1], '%.1f' % (a[1] / a[0]), '%.2f' % (100.0 * a[1] / tot), a[2], a[3]])
    # timed region: from the k_sketch launch that follows the warm-up batches to the end
    sk = [r for r in rows if r[0] in ('k_sketch', 'k_sketch32')]
    skip = int(sys.argv[sys.argv.index('--skip') + 1]) if '--skip' in sys.argv else 4      # verify batch + 3 warm-ups
    if '--timed' in sys.argv:      # round 6: the number of TIMED batches is what the caller knows (the warm-up runs follow the number of contexts the HBM has room for)
        skip = max(0, len(sk) - int(sys.argv[sys.argv.index('--timed') + 1]))
    t0 = sk[skip][1] if len(sk) > skip else rows[0][1]
    sel = [r for r in rows if r[1] >= t0 and not r[0].startswith('__amd')]
    t1 = max(r[2] for r in sel)
    ev = []
    for i, r in enumerate(sel):
        ev.append((r[1], 1, i)); ev.append((r[2], -1, i))
    ev.sort()
    live = set(); last = t0; hist = collections.Counter(); under = collections.defaultdict(float); solo = collections.Counter()
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            k = len(live); hist[min(k, 4)] += dt
            for j in live:
                under[sel[j][0]] += dt / max(k, 1)
            if k == 1:
                solo[sel[next(iter(live))][0]] += dt
        last = t
        (live.add if d > 0 else live.discard)(i)
    wall = t1 - t0
    nb = len(sk) - skip
    # per kernel name: time at least one launch of it is resident (union over the streams), time two or more are, and the sum of durations
    names = sorted(set(r[0] for r in sel))
    uni = collections.Counter(); multi = collections.Counter(); tot = collections.Counter()
    for nm in names:
        e2 = []
        for r in sel:
            if r[0] == nm: e2.append((r[1], 1)); e2.append((r[2], -1)); tot[nm] += r[2] - r[1]
        e2.sort(); k = 0; lt = t0
        for t, d in e2:
            if k >= 1: uni[nm] += t - lt
            if k >= 2: multi[nm] += t - lt
            lt = t; k += d
    print('timed region %.1f ms, %d batches, %d kernel launches' % (wall / 1e6, nb, len(sel)))
    for k in range(5):
        print('  %s kernels resident: %6.1f ms (%4.1f %%)' % (('%d' % k) if k < 4 else '4+', hist[k] / 1e6, 100.0 * hist[k] / wall))
    print('wall-time share per kernel (an instant is split evenly among the kernels resident in it): ms, %, ms per batch; time alone on the GPU')
    for n, v in sorted(under.items(), key=lambda x: -x[1])[:18]:
        print('  %-24s %7.1f  %4.1f %%  %5.2f   alone %6.1f ms' % (n, v / 1e6, 100.0 * v / wall, v / 1e6 / max(nb, 1), solo[n] / 1e6))
    print('residency per kernel name, ms per batch: sum of launch durations / time one or more launches are resident / time two or more are')
    for n, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
        print('  %-24s sum %6.2f   one+ %6.2f   two+ %6.2f' % (n, v / 1e6 / max(nb, 1), uni[n] / 1e6 / max(nb, 1), multi[n] / 1e6 / max(nb, 1)))


if __name__ == '__main__':
    main()
