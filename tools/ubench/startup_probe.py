"""where the driver's start-up goes: context creation, index build, first / second batch per context"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
t0 = time.time()
from vacmap_amd import synth, pipeline
from vacmap_amd.lib import Context, Index, load, align_batch_raw
print('import %.2f' % (time.time() - t0)); t = time.time()
ctx = Context(0); print('first context %.2f' % (time.time() - t)); t = time.time()
ref = synth.make_reference([100_000_000], seed=1); print('reference gen %.2f' % (time.time() - t)); t = time.time()
idx = Index.from_seqs(ctx, ['chr1'], ref, k=15, w=10); print('index build %.2f' % (time.time() - t)); t = time.time()
cat, off, _ = synth.sample_reads_concat(ref, 4096, mean_len=15000, err=0.10, seed=7); print('reads gen %.2f' % (time.time() - t)); t = time.time()
prm = load().params('H')
pipe = pipeline.Pipeline(idx, prm, inflight=3, first_ctx=ctx); print('2 more contexts %.2f' % (time.time() - t)); t = time.time()
for rep in range(3):
    for i, cx in enumerate(pipe.ctxs):
        t = time.time(); raw = align_batch_raw(cx, idx, prm, cat, off); dt = time.time() - t
        print('rep %d ctx %d align %.3f s (device %.1f ms)' % (rep, i, dt, raw.stats['ms_total'])); raw.close()
