import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
from vacmap_amd import synth
contigs = synth.make_reference([20_000_000], seed=1)
oi = O.Index.from_seqs(['chr1'], [contigs[0].tobytes()], k=15, w=10)
op = O.params('H')
cat, off, _ = synth.sample_reads_concat(contigs, 2048, mean_len=15000, err=0.1, seed=5)
rds = [cat[off[i]:off[i+1]].tobytes() for i in range(2048)]
for nt in (32, 128, 256):
    t = time.time(); st, recs = O.align_batch(oi, rds, op, nthreads=nt); dt = time.time() - t
    print(os.environ.get('VMO_NO_MALLOPT', 'mallopt'), nt, 'threads', round(dt, 2), 's', round(len(rds) / dt, 1), 'reads/s', flush=True)
