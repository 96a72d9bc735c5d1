#!/usr/bin/env python3
"""Gap-fill schedule alone (vm_k_cigar_batch_banded = what vm_align_batch launches for E5) on ONT-shape problems:
    python tools/ubench/gapfill_bench.py [--n 200000] [--len 268] [--err 0.10] [--reps 3]
prints problems, wall ms per call (incl. upload / download), stats. Run under rocprofv3 --kernel-trace for the kernel times."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=200000); ap.add_argument('--len', type=int, default=268); ap.add_argument('--spread', type=int, default=40)
    ap.add_argument('--real', type=int, default=0, help='replicate the oracle-collected problems of dp_problems_160reads.npz to at least this many')
    ap.add_argument('--err', type=float, default=0.10); ap.add_argument('--reps', type=int, default=3); ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    from vacmap_amd.lib import Context
    rng = np.random.default_rng(a.seed)
    if a.real:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_problems_160reads.npz'))
        to = np.concatenate([[0], np.cumsum(d['tl'])]); qo = np.concatenate([[0], np.cumsum(d['ql'])])
        tb = d['t'].tobytes().decode(); qb = d['q'].tobytes().decode()
        ts = [tb[to[i]:to[i + 1]] for i in range(len(d['tl']))]; qs = [qb[qo[i]:qo[i + 1]] for i in range(len(d['ql']))]
        rep = (a.real + len(ts) - 1) // len(ts)
        ts = ts * rep; qs = qs * rep
        return run(a, ts, qs)
    L = np.clip(rng.normal(a.len, a.spread, a.n).astype(np.int64), 20, 500)
    tot = int(L.sum())
    base = rng.integers(0, 4, tot, dtype=np.uint8)
    lut = np.frombuffer(b'ACGT', dtype=np.uint8)
    ts, qs = [], []
    off = np.concatenate([[0], np.cumsum(L)])
    for i in range(a.n):
        t = base[off[i]:off[i + 1]]
        r = rng.random(len(t))
        q = []
        # substitutions / deletions / insertions in equal parts
        keep = r >= a.err / 3
        sub = (r >= a.err / 3) & (r < 2 * a.err / 3)
        tq = t.copy(); tq[sub] = (tq[sub] + 1 + rng.integers(0, 3, int(sub.sum()))) % 4
        tq = tq[keep]
        ins = np.nonzero(rng.random(len(tq)) < a.err / 3)[0]
        tq = np.insert(tq, ins, rng.integers(0, 4, len(ins), dtype=np.uint8))
        ts.append(lut[t].tobytes().decode()); qs.append(lut[tq].tobytes().decode())
    run(a, ts, qs)


def run(a, ts, qs):
    from vacmap_amd.lib import Context
    ctx = Context(0)
    a.n = len(ts)
    for rep in range(a.reps):
        t0 = time.time(); cg, flag, st = ctx.k_cigar_batch_banded(ts, qs); dt = time.time() - t0
        print('rep', rep, 'problems', a.n, 'wall ms %.1f' % (dt * 1e3), st, 'ns kept', sorted(set(int(f) for f in flag)))


if __name__ == '__main__':
    main()
