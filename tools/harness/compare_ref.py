#!/usr/bin/env python3
"""Pin the CPU oracle against the reference's own Python on synthetic reads (build container only).

usage: compare_ref.py [--mode H|L|R] [--n 40] [--seed 7] [--reflen 300000] [--sv] [--procs 8]
For every read: reference get_readmap_DP_test (mammap_*.py, imported in place) vs oracle vmo_align_read.
"""
import argparse, os, sys, time
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT)
sys.path.insert(0, _HERE)
import refrun
from refrun import O
from vacmap_amd import synth


def build_case(args):
    contigs = synth.make_reference([args.reflen, args.reflen // 2], seed=args.seed)
    names = ['chrA', 'chrB']
    if args.repeats:
        rng = np.random.default_rng(args.seed + 100)
        c0 = contigs[0].copy()
        elem = synth.make_reference([3000], seed=args.seed + 101)[0]
        for t in range(25):   # interspersed copies, 2 % divergence
            p = int(rng.integers(0, len(c0) - 3000))
            c0[p:p + 3000] = synth.mutate(elem, 0.02, rng, ratio=(1, 0, 0))[:3000]
        unit = synth.make_reference([37], seed=args.seed + 102)[0]
        p = len(c0) // 3
        c0[p:p + 37 * 80] = np.tile(unit, 80)     # tandem repeat
        c1 = contigs[1].copy()
        c1[5000:8000] = elem
        contigs = [c0, c1]
    src = contigs
    if args.dense:
        # repeat-dense case for the *_fast chain variants: a 6 kb element in 64 copies (1 % divergence) and a period-23 tandem array of
        # 9.2 kb; reads are drawn from the copies / the array (+ flanks) only
        rng = np.random.default_rng(args.seed + 400)
        c0 = contigs[0].copy()
        elem = synth.make_reference([6000], seed=args.seed + 401)[0]
        starts = []
        ncopy = 64
        step = (len(c0) - 20000) // (ncopy + 1)
        for t in range(ncopy):
            p = 5000 + t * step
            c0[p:p + 6000] = synth.mutate(elem, 0.01, rng, ratio=(1, 0, 0))[:6000]
            starts.append(p)
        unit = synth.make_reference([23], seed=args.seed + 402)[0]
        tp = 5000 + ncopy * step + 2000
        c0[tp:tp + 23 * 400] = np.tile(unit, 400)
        # a diverged second copy of the array with its flanks on the other contig: reads through the array get a secondary chain,
        # i.e. more than one guide (LC-mm) on top of the dense local anchors
        c1 = contigs[1].copy()
        seg = synth.mutate(c0[tp - 3000:tp + 9200 + 3000], 0.004, rng, ratio=(1, 0, 0))
        c1[40000:40000 + len(seg)] = seg
        contigs = [c0, c1]
        src = [c0[p - 1500:p + 7500] for p in starts[:6]] + [c0[tp - 4000:tp + 9200 + 4000]] * 3
    if args.chimera:
        pass
    if args.sv:
        # donor genome with nested / complex SVs; reads are sampled from the donor, aligned to the original
        L = args.reflen
        ops = [('INV', L // 10, 3000), ('DEL', 2 * L // 10, 1500), ('INS', 3 * L // 10, 800, 5), ('DUP', 4 * L // 10, 2500, 2),
               ('INVDUP', 5 * L // 10, 2000), ('INV', 6 * L // 10, 600), ('DEL', 7 * L // 10, 300), ('DUP', 8 * L // 10, 400, 3)]
        d0 = synth.implant_svs(contigs[0], ops)
        # translocation: move a 5 kb piece of chrB into chrA's donor
        piece = contigs[1][1000:6000]
        cut = 9 * L // 10
        d0 = np.concatenate([d0[:cut], piece, d0[cut:]])
        src = [d0, contigs[1]]
    if args.mode == 'L':
        reads = synth.sample_reads(src, args.n, mean_len=args.mean or 18000, err=0.005, seed=args.seed + 1, shape='hifi', min_len=5000)
        k = 19
    else:
        reads = synth.sample_reads(src, args.n, mean_len=args.mean or 15000, err=(args.err if args.err >= 0 else 0.10), seed=args.seed + 1, shape='ont')
        k = 15
    if args.dense:     # two-locus reads through the tandem array: more than one guide chain + dense local anchors (LC-mm and its _fast twin)
        rng = np.random.default_rng(args.seed + 410)
        err = args.err if args.err >= 0 else (0.005 if args.mode == 'L' else 0.10)
        c0 = contigs[0]
        eq = np.concatenate([[0], np.cumsum(c0[23:] == c0[:-23])])
        tp = int(np.flatnonzero(eq[500:] - eq[:-500] == 500)[0])    # first position of the period-23 array
        for i in range(3):
            a = synth.mutate(c0[max(0, tp - 2500 - 500 * i):tp + 9200 + 2500], err, rng)
            b = synth.mutate(contigs[1][20000 + 9000 * i:26000 + 9000 * i], err, rng)
            reads.append(('twolocus%d' % i, np.concatenate([a, b] if i != 1 else [b, a]), {}))
    if args.chimera:   # join pairs of reads (translocation-like chimeras) and add pure-random reads (unmapped)
        rng = np.random.default_rng(args.seed + 200)
        out = []
        for i in range(0, len(reads) - 1, 2):
            a, b = reads[i][1], reads[i + 1][1]
            out.append(('chim%d' % i, np.concatenate([a[:len(a) // 2], b[len(b) // 3:]]), {}))
        for i in range(4):
            out.append(('rand%d' % i, synth.make_reference([4000 + 1000 * i], seed=args.seed + 300 + i)[0], {}))
        out.append(('short0', reads[0][1][:300], {})); out.append(('short1', reads[1][1][:80], {})); out.append(('tiny', reads[1][1][:12], {}))
        withn = reads[2][1].copy(); withn[1000:1040] = ord('N'); withn[5000] = ord('N'); out.append(('withN', withn, {}))
        reads = out
    return names, contigs, reads, k


def worker(job):
    mode, names, contigs, reads, k, lo, hi = job
    ix = O.Index.from_seqs(names, [synth.tostr(c) for c in contigs], k=k, w=10)
    al = refrun.Aligner(oracle_index=ix)
    ctx = refrun.RefContext(mode, al)
    prm = O.params(mode)
    res = []
    for i in range(lo, hi):
        name, rd, truth = reads[i]
        s = synth.tostr(rd)
        t0 = time.time()
        st, one = ctx.align(name, s)
        t1 = time.time()
        st2, recs = O.align_read(ix, s, prm)
        ref_t = [(t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), t[8]) for t in one]
        ora_t = [(names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]) for t in recs]
        ok = (st == 0) == (st2 == 0) and ref_t == ora_t
        res.append((i, ok, st, st2, ref_t, ora_t, t1 - t0, repr(getattr(ctx, 'last_exc', None)) if st < 0 else ''))
    return res, O.fast_counters()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='H'); ap.add_argument('--n', type=int, default=40); ap.add_argument('--seed', type=int, default=7)
    ap.add_argument('--reflen', type=int, default=300000); ap.add_argument('--sv', action='store_true')
    ap.add_argument('--procs', type=int, default=8); ap.add_argument('--mean', type=int, default=0)
    ap.add_argument('-v', action='store_true'); ap.add_argument('--repeats', action='store_true'); ap.add_argument('--chimera', action='store_true')
    ap.add_argument('--dense', action='store_true'); ap.add_argument('--err', type=float, default=-1.0)
    args = ap.parse_args()
    names, contigs, reads, k = build_case(args)
    import multiprocessing as mp
    chunks = np.linspace(0, len(reads), args.procs + 1).astype(int)
    jobs = [(args.mode, names, contigs, reads, k, int(chunks[i]), int(chunks[i + 1])) for i in range(args.procs) if chunks[i + 1] > chunks[i]]
    with mp.Pool(len(jobs)) as pool:
        got = pool.map(worker, jobs)
    out = [r for rs, _ in got for r in rs]
    fc = np.sum([c for _, c in got], axis=0)
    print('oracle fast-path calls (GC-fast, LC-fast, LC-mm-fast):', fc.tolist())
    nok = sum(1 for r in out if r[1]); nrec = sum(len(r[4]) for r in out); nskip = sum(1 for r in out if r[2] < 0)
    nun = sum(1 for r in out if r[2] == 0 and not r[4])
    print('mode %s sv=%s reads=%d identical=%d ref_records=%d ref_raised=%d unmapped=%d ref_s/read=%.2f' % (
        args.mode, args.sv, len(out), nok, nrec, nskip, nun, np.mean([r[6] for r in out])))
    for r in out:
        if not r[1] or args.v:
            print('read', r[0], 'status ref/oracle', r[2], r[3], r[7])
            for a in r[4]:
                print('   REF', a[:7], a[7][:60], len(a[7]))
            for a in r[5]:
                print('   ORA', a[:7], a[7][:60], len(a[7]))
    sys.exit(0 if nok == len(out) else 1)
