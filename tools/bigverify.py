import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib as O
from vacmap_amd import synth
from vacmap_amd.lib import Context, Index, align_batch, load
ctx = Context(0); lib = load()
SO = int(os.environ.get('BIGVERIFY_SEED_OFFSET', '0'))      # a second, independent sample of reads: BIGVERIFY_SEED_OFFSET=1000
print('bigverify: GPU path (vm_align_batch) vs the oracle, whole records per read (read seed offset %d)' % SO, flush=True)
if '--no-hg38' not in sys.argv:
    # the metric's defining size: hg38-size reference (24 contigs, 3.1 Gb), ONT shape in mode H k=15 and HiFi shape in mode L k=19 (config 3)
    names38 = list(synth.HG38_NAMES); c38 = synth.make_reference_fast(synth.hg38_like_lengths(), seed=3)
    for mode, k, n, kw in (('H', 15, 3000, dict(mean_len=15000, err=0.10)), ('L', 19, 1500, dict(mean_len=18000, err=0.005, shape='hifi'))):
        cat, off, _ = synth.sample_reads_concat(c38, n, seed=177 + SO, **kw)
        seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)]
        gi = Index.from_seqs(ctx, names38, c38, k=k, w=10)
        t = time.time(); oi = O.Index.from_seqs(names38, c38, k=k, w=10); ti = time.time() - t
        t = time.time(); status, recs, stats = align_batch(ctx, gi, lib.params(mode), seqs); tg = time.time() - t
        t = time.time(); ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=32); tc = time.time() - t
        a = {}; b = {}
        for t_ in recs: a.setdefault(t_[0], []).append(t_[1:])
        for t_ in orecs: b.setdefault(t_[0], []).append(t_[1:])
        bad = sum(1 for i in range(n) if a.get(i) != b.get(i))
        print('hg38-size ref, mode', mode, 'k', k, 'reads', n, 'records', len(orecs), 'status equal', [(int(x) == 0) for x in status] == [(int(x) == 0) for x in ost],
              'reads with different records', bad, 'gpu %.2fs cpu %.2fs (oracle index %.0fs)' % (tg, tc, ti), 'anchors/read %.0f' % (stats['n_anchors'] / n), flush=True)
        del gi, oi
    del c38
contigs = synth.make_reference([30_000_000], seed=1)
names = ['chr1']
for mode, k, n, kw in (('H', 15, 3000, dict(mean_len=15000, err=0.10)), ('L', 19, 1500, dict(mean_len=15000, err=0.005)), ('R', 15, 1500, dict(mean_len=12000, err=0.10)), ('S', 15, 1000, dict(mean_len=12000, err=0.13)),
                        ('H', 15, 500, dict(mean_len=45000, err=0.10)), ('H', 15, 2000, dict(mean_len=2500, err=0.15))):   # long reads (large LDS buckets, many stripes); short reads (small DP problems)
    cat, off, _ = synth.sample_reads_concat(contigs, n, seed=77 + SO, **kw)
    seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)]
    gi = Index.from_seqs(ctx, names, [contigs[0].tobytes()], k=k, w=10)
    oi = O.Index.from_seqs(names, [contigs[0].tobytes()], k=k, w=10)
    t = time.time(); status, recs, stats = align_batch(ctx, gi, lib.params(mode), seqs); tg = time.time() - t
    t = time.time(); ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=min(os.cpu_count(), 128)); tc = time.time() - t
    same_st = [(int(s) == 0) for s in status] == [(int(s) == 0) for s in ost]
    bad = 0
    if recs != orecs:
        a = {}; b = {}
        for t_ in recs: a.setdefault(t_[0], []).append(t_[1:])
        for t_ in orecs: b.setdefault(t_[0], []).append(t_[1:])
        bad = sum(1 for i in range(n) if a.get(i) != b.get(i))
    print('mode', mode, 'reads', n, 'records', len(orecs), 'status equal', same_st, 'reads with different records', bad, 'gpu %.2fs cpu %.2fs' % (tg, tc), 'failed', int(sum(1 for s in status if s != 0)), flush=True)
    del gi, oi

# reads from a donor with an insertion or deletion of 30-600 bp every ~2.5 kb: gap-fill problems far from square (the banded fill's proof
# has to reject them or keep them for the right reason)
rng = np.random.default_rng(5)
ref = synth.make_reference([6_000_000], seed=3)
ops = []
for p_ in range(20_000, 5_900_000, 2500):
    n_ = int(rng.integers(30, 600))
    ops.append(('DEL', p_, n_) if rng.random() < 0.5 else ('INS', p_, n_, int(rng.integers(1 << 30))))
donor = synth.implant_svs(ref[0], ops)
for mode, n, kw in (('H', 1500, dict(mean_len=12000, err=0.08)), ('R', 800, dict(mean_len=9000, err=0.08))):
    cat, off, _ = synth.sample_reads_concat([donor], n, seed=99 + SO, **kw)
    seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)]
    gi = Index.from_seqs(ctx, names, [ref[0].tobytes()], k=15, w=10)
    oi = O.Index.from_seqs(names, [ref[0].tobytes()], k=15, w=10)
    status, recs, stats = align_batch(ctx, gi, lib.params(mode), seqs)
    ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=min(os.cpu_count(), 128))
    a = {}; b = {}
    for t_ in recs: a.setdefault(t_[0], []).append(t_[1:])
    for t_ in orecs: b.setdefault(t_[0], []).append(t_[1:])
    bad = sum(1 for i in range(n) if a.get(i) != b.get(i))
    print('indel donor, mode', mode, 'reads', n, 'records', len(orecs), 'status equal', [(int(x) == 0) for x in status] == [(int(x) == 0) for x in ost], 'reads with different records', bad, flush=True)
    del gi, oi

# reads across COMPLEX SVs: a donor made from a 30 Mb reference of three contigs by the vacsim grammar (nested INV / DUP / TRA / INS / DEL: bench.py's vacsim_r
# text, 150 per line), reads drawn around the events — split alignments, inversions inside duplications, the cases the non-linear chain exists for
if '--no-vacsim' not in sys.argv:
    from vacmap_amd import vacsim
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench as _bench
    refv = synth.make_reference([12_000_000, 10_000_000, 8_000_000], seed=11); namesv = ['chrA', 'chrB', 'chrC']
    donor, pieces, events = vacsim.implant(refv, _bench.VACSIM_TEXT % {'n': 150}, seed=172 + SO)
    around = vacsim.event_positions(pieces, events)
    for mode, k, n, kw in (('R', 15, 1500, dict(mean_len=18000, err=0.005, shape='hifi', min_len=5000)), ('H', 15, 1500, dict(mean_len=15000, err=0.10)),
                            ('S', 15, 800, dict(mean_len=12000, err=0.12)), ('L', 19, 1000, dict(mean_len=18000, err=0.005, shape='hifi', min_len=5000))):
        cat, off, _ = synth.sample_reads_concat(donor, n, seed=55 + SO, around=around, **kw)
        seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)]
        gi = Index.from_seqs(ctx, namesv, [c_.tobytes() for c_ in refv], k=k, w=10)
        oi = O.Index.from_seqs(namesv, [c_.tobytes() for c_ in refv], k=k, w=10)
        status, recs, stats = align_batch(ctx, gi, lib.params(mode), seqs)
        ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=min(os.cpu_count(), 128))
        a = {}; b = {}
        for t_ in recs: a.setdefault(t_[0], []).append(t_[1:])
        for t_ in orecs: b.setdefault(t_[0], []).append(t_[1:])
        bad = sum(1 for i in range(n) if a.get(i) != b.get(i))
        print('vacsim donor (%d events), mode' % len(events), mode, 'reads', n, 'records', len(orecs), 'records per read %.2f' % (len(orecs) / n), 'status equal',
              [(int(x) == 0) for x in status] == [(int(x) == 0) for x in ost], 'reads with different records', bad, flush=True)
        del gi, oi
