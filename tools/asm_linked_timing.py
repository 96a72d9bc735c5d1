import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import oracle_lib as O, kernel_cases as KC
from vacmap_amd.lib import Context
ctx = Context(0)
meta, arr = KC.asm_golden()
# timing: the 105 k-anchor first-round batch of the 600 kb contig (no carried state) + 8x noise
c = meta['AS3']['contigs'][1]
e = [x for x in c['linked_calls'] if 'key' in x and x['which'] == 0][0]
kk = e['key']; rows = arr[kk + '_rows']; n_pre = e['n_pre']
rng = np.random.default_rng(1)
new = rows[n_pre:]
for mult in (0, 8, 64):
    m = mult * len(new)
    q = rng.integers(new[:, 0].min(), new[:, 0].max() + 1, m)
    noise = np.stack([q, rng.integers(0, 3_000_000_000, m), rng.choice([-1, 1], m), np.full(m, 15)], axis=1).astype(np.int64)
    allnew = np.concatenate([new, noise]); allnew = allnew[np.argsort(allnew[:, 0], kind='stable')]
    linked = np.ascontiguousarray(np.concatenate([rows[:n_pre], allnew]))
    kw = e['kw']
    args = (int(kw['kmersize']), kw['skipcost'], int(kw['maxdiff']), int(kw['maxgap']), e['g_max_scores'], e['g_max_index'], arr[kk + '_preS'], arr[kk + '_preP'], e['prereadloc'])
    ctx.chain_linked(linked, 0, *args)
    t = time.time(); r = ctx.chain_linked(linked, 0, *args); dt = time.time() - t
    t = time.time(); g, S, P, SA = O.chain_linked_raw(linked, 0, *args); dto = time.time() - t
    print('anchors %d: device call %.3f s (%.2f us per anchor), oracle %.3f s, equal %s, hot %d cold %d' % (len(linked), dt, dt / len(linked) * 1e6, dto, np.array_equal(r['P'], P) and np.array_equal(r['S'], S), r['n_hot'], r['n_cold']), flush=True)
