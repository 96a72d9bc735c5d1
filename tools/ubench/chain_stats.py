"""How the global chain DP's candidate scans look on synthetic reads (CPU, oracle = test infrastructure): per anchor the number of
candidates the descending-S scan evaluates before its break rule fires, and where a new score lands in the sorted index.
    python tools/ubench/chain_stats.py [ont|hifi] [ref_mb] [reads]
"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as O
from vacmap_amd import synth

shape = sys.argv[1] if len(sys.argv) > 1 else 'ont'
ref_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 30
nreads = int(sys.argv[3]) if len(sys.argv) > 3 else 40
k = 15 if shape == 'ont' else 19
mode = 'H' if shape == 'ont' else 'L'
contigs = synth.make_reference([int(ref_mb * 1e6)], seed=1)
if shape == 'ont':
    reads = synth.sample_reads(contigs, nreads, mean_len=15000, err=0.10, seed=5, min_len=1000)
else:
    reads = synth.sample_reads(contigs, nreads, mean_len=18000, err=0.005, seed=5, min_len=5000, shape='hifi')
oi = O.Index.from_seqs(['chr1'], [synth.tostr(contigs[0])], k=k, w=10)
prm = O.params(mode)
firsts = []; ranks = []; ns = []; groups = []
for r in reads:
    s = synth.tostr(r[1])
    a = oi.map(s, check_num=100)
    if len(a) < 3: continue
    flip, a = O.strand_flip(a, len(s))
    order = np.argsort(a[:, 0], kind='stable'); a = a[order]
    g, S, P, SA = O.chain_global_raw(a, k, skipcost=prm.global_skipcost, maxdiff=prm.global_maxdiff, mode=mode)
    n = len(a); ns.append(n)
    q = a[:, 0]; l = a[:, 3]
    # group prefix: candidates of anchor i = anchors with q_j < q_i
    starts = np.searchsorted(q, q, side='left')
    for i in range(1, n):
        pre = S[:starts[i]]
        if len(pre) == 0: firsts.append(0); continue
        firsts.append(int((pre > S[i] - l[i]).sum()))
        ranks.append(int((S[:i] > S[i]).sum()))
    groups.append(len(np.unique(q)))
f = np.array(firsts); rk = np.array(ranks)
print(shape, 'reads', len(ns), 'anchors/read mean', np.mean(ns), 'positions/read', np.mean(groups))
for nm, v in (('candidates evaluated per anchor', f), ('insertion rank from the top', rk)):
    print(nm, 'mean %.1f' % v.mean(), 'pct<=4 %.3f <=8 %.3f <=15 %.3f <=31 %.3f <=63 %.3f' % tuple((v <= t).mean() for t in (4, 8, 15, 31, 63)), 'max', v.max())
