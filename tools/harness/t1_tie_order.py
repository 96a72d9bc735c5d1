"""SURVEY T1, quantified: how many reads change when the reference's np.argsort calls break ties differently?

The reference sorts with np.argsort (numba: an unstable quicksort) at mammap_clrnano.py:23103 (guide by reference position), :23572 and :23652
(hit2work_1) and :28585 (local anchors by q + l); oracle, goldens and kernels define STABLE order. This script (build container only: it imports the
reference in place through tools/harness/refload.py) runs the imported reference on the golden reads and on seeded synthetic reads with
np.argsort patched to (i) stable — the baseline, (ii) stable with every run of equal keys reversed, (iii) stable with every run of equal keys
permuted by a seeded generator, and reports the reads whose outputs differ from the baseline, by stage: V2 = decode_hit (paths, score, MAPQ),
V3 = the local chain handed to extend_func, V6 = the records of get_readmap_DP_test.

    python tools/harness/t1_tie_order.py [--synthetic 60] [--len 4000] [--procs 8] [--out profiles/r04_t1_tie_order.json]
"""
import argparse, json, os, sys, time
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _HERE); sys.path.insert(0, os.path.join(_ROOT, 'tests')); sys.path.insert(0, _ROOT)

VARIANT = {'v': 'stable', 'rng': None, 'ties': 0, 'calls': 0}
_ORIG = np.argsort


def _argsort(a, axis=-1, kind=None, order=None, **kw):
    idx = _ORIG(a, axis=axis, kind='stable', order=order, **kw)
    VARIANT['calls'] += 1
    arr = np.asarray(a)
    if VARIANT['v'] == 'stable' or arr.ndim != 1 or len(arr) < 2:
        return idx
    keys = arr[idx]
    brk = np.flatnonzero(keys[1:] != keys[:-1]) + 1
    starts = np.concatenate([[0], brk]); ends = np.concatenate([brk, [len(keys)]])
    out = idx.copy()
    for s, e in zip(starts, ends):
        if e - s > 1:
            VARIANT['ties'] += 1
            out[s:e] = idx[s:e][::-1] if VARIANT['v'] == 'reversed' else VARIANT['rng'].permutation(idx[s:e])
    return out


def _canon(x):
    if isinstance(x, (list, tuple)):
        return [_canon(y) for y in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, np.generic):
        return x.item()
    return x


def run_group(job):
    """one (mode, k, reference, reads) group in a process of its own: the three variants on every read"""
    mode, k, names, contigs, reads, tag = job
    import oracle_lib as O
    O.build(); O.lib()
    import refrun
    ix = O.Index.from_seqs(names, contigs, k=k, w=10)
    al = refrun.Aligner(oracle_index=ix)
    ctx = refrun.RefContext(mode, al)
    np.argsort = _argsort                       # (refload patched np.argsort to stable when it imported the module: replace that patch)
    m = ctx.m
    cap = {}
    o_dec, o_ext = m.decode_hit, m.extend_func

    def dec(*a, **kw):
        r = o_dec(*a, **kw); cap['v2'] = _canon([r[0], r[1], r[4]]); return r

    def ext(*a, **kw):
        cap.setdefault('v3', []).append(_canon(a[0] if a else None)); return o_ext(*a, **kw)
    m.decode_hit, m.extend_func = dec, ext
    res = []
    for ri, (name, seq) in enumerate(reads):
        per = {}
        for v in ('stable', 'reversed', 'random'):
            VARIANT.update(v=v, rng=np.random.default_rng(1000 + ri), ties=0, calls=0)
            cap.clear()
            st, one = ctx.align(name, seq)
            per[v] = {'v2': cap.get('v2'), 'v3': cap.get('v3'), 'v6': [st, _canon(one)], 'ties': VARIANT['ties'], 'calls': VARIANT['calls']}
        d = {'group': tag, 'read': name, 'len': len(seq), 'tied_sorts': per['reversed']['ties'], 'argsort_calls': per['stable']['calls']}
        for v in ('reversed', 'random'):
            d[v] = [s for s in ('v2', 'v3', 'v6') if per[v][s] != per['stable'][s]]
        res.append(d)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=60, help='synthetic reads per mode (H k15, L k19, R k15)')
    ap.add_argument('--len', type=int, default=4000); ap.add_argument('--procs', type=int, default=8); ap.add_argument('--golden', type=int, default=1)
    ap.add_argument('--out', default=os.path.join(_ROOT, 'profiles', 'r04_t1_tie_order.json'))
    args = ap.parse_args()
    from vacmap_amd import synth
    jobs = []
    if args.golden:
        meta = json.load(open(os.path.join(_ROOT, 'tests', 'golden', 'cases.json'))); arr = np.load(os.path.join(_ROOT, 'tests', 'golden', 'cases.npz'))
        for cid, c in sorted(meta.items()):
            if not isinstance(c, dict) or 'reads' not in c:
                continue
            contigs = [arr['%s_contig%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))]
            reads = [(r['name'], arr['%s_r%d_seq' % (cid, ri)].tobytes().decode()) for ri, r in enumerate(c['reads'])]
            per = max(1, -(-len(reads) // 3))
            for a in range(0, len(reads), per):               # (cut into pieces so that the processes stay busy)
                jobs.append((c['mode'], c['k'], c['names'], contigs, reads[a:a + per], 'golden_' + cid))
    for mode, k, err, seed in (('H', 15, 0.10, 71), ('L', 19, 0.005, 72), ('R', 15, 0.10, 73)):
        contigs = synth.make_reference([400000, 150000], seed=seed)
        cs = [c.tobytes().decode() for c in contigs]
        cat, off, _ = synth.sample_reads_concat(contigs, args.synthetic, mean_len=args.len, err=err, seed=seed + 100, min_len=1000, max_len=3 * args.len,
                                               shape='hifi' if mode == 'L' else 'ont', sd=args.len // 8)
        reads = [('s%s%d' % (mode, i), cat[off[i]:off[i + 1]].tobytes().decode()) for i in range(args.synthetic)]
        per = max(1, -(-len(reads) // 4))
        for a in range(0, len(reads), per):
            jobs.append((mode, k, ['a', 'b'], cs, reads[a:a + per], 'synthetic_%s_k%d' % (mode, k)))
    t0 = time.time()
    import multiprocessing as mp
    with mp.get_context('spawn').Pool(args.procs) as pool:
        parts = pool.map(run_group, jobs, chunksize=1)
    rows = [r for p in parts for r in p]
    summ = {}
    for r in rows:
        s = summ.setdefault(r['group'], {'reads': 0, 'reads_with_tied_sorts': 0, 'reversed': {'v2': 0, 'v3': 0, 'v6': 0}, 'random': {'v2': 0, 'v3': 0, 'v6': 0}, 'differing_reads': []})
        s['reads'] += 1; s['reads_with_tied_sorts'] += int(r['tied_sorts'] > 0)
        for v in ('reversed', 'random'):
            for st in r[v]:
                s[v][st] += 1
        if r['reversed'] or r['random']:
            s['differing_reads'].append({'read': r['read'], 'len': r['len'], 'reversed': r['reversed'], 'random': r['random']})
    tot = {'reads': len(rows), 'reads_with_tied_sorts': sum(int(r['tied_sorts'] > 0) for r in rows),
           'reads_v6_differs_reversed': sum('v6' in r['reversed'] for r in rows), 'reads_v6_differs_random': sum('v6' in r['random'] for r in rows),
           'reads_any_stage_differs': sum(bool(r['reversed'] or r['random']) for r in rows), 'seconds': time.time() - t0}
    out = {'what': __doc__.split('\n\n')[0], 'total': tot, 'groups': summ}
    json.dump(out, open(args.out, 'w'), indent=1)
    print(json.dumps(tot)); print('->', args.out)


if __name__ == '__main__':
    main()
