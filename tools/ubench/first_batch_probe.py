"""How long does a context's FIRST batch take (its grow-only pools are allocated on the way) next to its later ones? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vacmap_amd import synth
from vacmap_amd.lib import Context, Index, ResidentReads
ctx = Context(0)
contigs = synth.make_reference([100_000_000], seed=1)
t = time.time(); index = Index.from_seqs(ctx, ['chr1'], contigs, k=15, w=10); print('index %.2f s' % (time.time() - t))
cat, off, _ = synth.sample_reads_concat(contigs, 4096, mean_len=15000, err=0.10, seed=5)
prm = ctx.lib.params('H')
rr = ResidentReads(ctx, concat=cat, offsets=off)
for name, cx in (('ctx0', ctx), ('ctx1', Context(0, lib=ctx.lib))):
    for i in range(3):
        t = time.time(); st, _, stats = rr.align(index, prm, want_records=False, ctx=cx); dt = time.time() - t
        print('%s batch %d: wall %.3f s, device %.1f ms, host syncs %d' % (name, i, dt, stats['ms_total'], stats['n_host_syncs']))
