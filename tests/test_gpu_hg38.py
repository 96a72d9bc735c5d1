"""GPU parity at the metric's DEFINING SIZE (BASELINE configs[2]/[3], SURVEY §8(d) config 3/4 reference): the hg38-size synthetic
reference (24 contigs with hg38's proportions, 3.1 Gb, seed 3) is indexed on the MI355X and by the oracle; the index columns, the
`.map()` anchors and the full records of a stratified read sample must be identical — mode H k=15 (ONT shape) and mode L k=19
(HiFi shape, config 3), including reads from the last contigs whose global offsets exceed 2^31 and reads at contig ends.
Reference call sites: /root/reference/src/vacmap/vacmap:344-367 (index + contig table), mammap_clrnano.py:23985 (map)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from vacmap_amd.lib import Context
    return Context(0)


@pytest.fixture(scope='module')
def hg38_ref():
    from vacmap_amd import synth
    return list(synth.HG38_NAMES), synth.make_reference_fast(synth.hg38_like_lengths(), seed=3)


def stratified_reads(contigs, shape, seed):
    """(reads, where): reads from the first / middle / last contigs, at contig starts and ends, both strands, plus one chimera that
    joins the first and the last contig and one read with an inversion"""
    from vacmap_amd import synth
    rng = np.random.default_rng(seed)
    err, mean, sd = (0.10, 15000, 0) if shape == 'ont' else (0.005, 18000, 2000)
    spec = []
    nc = len(contigs)
    for c in (0, 1, nc // 2, nc - 3, nc - 2, nc - 1):            # chrX / chrY live above 2^31 on the global axis
        L = len(contigs[c])
        for st in (0, L // 3, L - 1):
            ln = int(np.clip(rng.gamma(2.0, mean / 2.0), 3000, 40000)) if shape == 'ont' else int(max(rng.normal(mean, sd), 5000))
            st = min(max(st, 0), L - ln)
            spec.append((c, st, ln, int(rng.integers(0, 2))))
    reads, where = [], []
    for c, st, ln, strand in spec:
        frag = contigs[c][st:st + ln]
        if strand:
            frag = synth.revcomp(frag)
        reads.append(synth.mutate(frag, err, rng).tobytes()); where.append((c, st, ln, strand))
    a = contigs[0][5_000_000:5_007_000]; b = contigs[nc - 1][1_000_000:1_008_000]
    reads.append(synth.mutate(np.concatenate([a, synth.revcomp(b)]), err, rng).tobytes()); where.append(('chimera', 0, 15000, 0))
    d = contigs[nc - 2][40_000_000:40_016_000].copy()
    d[6000:9000] = synth.revcomp(d[6000:9000])
    reads.append(synth.mutate(d, err, rng).tobytes()); where.append(('inv', 0, 16000, 0))
    return reads, where


@pytest.mark.parametrize('mode,k,shape', [('H', 15, 'ont'), ('L', 19, 'hifi')])
def test_hg38_size_index_map_and_records(ctx, oracle, hg38_ref, mode, k, shape):
    from vacmap_amd.lib import Index, align_batch
    names, contigs = hg38_ref
    gi = Index.from_seqs(ctx, names, contigs, k=k, w=10)
    oi = oracle.Index.from_seqs(names, contigs, k=k, w=10)
    assert gi.offsets == oi.offsets and gi.offsets[-1] > 2 ** 31 and sum(gi.lens) == 3_100_000_000
    assert gi.n_minimizers() == oi.n_minimizers() > 500_000_000
    assert gi.n_distinct() == oi.n_distinct() and gi.mid_occ == oi.mid_occ
    gh, gp = gi.minimizers()                       # read back from the device: hashes recovered from the table, positions as stored
    assert np.array_equal(gp, oi.positions_view())
    oh = np.ctypeslib.as_array(oracle.lib().vmo_index_hashes(oi.h), shape=(len(gh),))
    assert np.array_equal(gh, oh)
    del gh, gp, oh
    reads, where = stratified_reads(contigs, shape, seed=100 + k)
    anchors = ctx.map_batch(gi, reads, check_num=100, mid_occ=-1)
    n_hi = 0
    for i, rd in enumerate(reads):
        oa = oi.map(rd, 100, -1)
        assert np.array_equal(anchors[i], oa), 'anchors of read %d %r differ' % (i, where[i])
        n_hi += int((oa[:, 1] > 2 ** 31).sum()) if len(oa) else 0
    assert n_hi > 1000                              # anchors beyond 2^31 were exercised
    prm = ctx.lib.params(mode); op = oracle.params(mode)
    status, recs, stats = align_batch(ctx, gi, prm, reads)
    nrec = 0
    for i, rd in enumerate(reads):
        ost, orecs = oracle.align_read(oi, rd, op)
        mine = [t[1:] for t in recs if t[0] == i]
        assert (status[i] == 0) == (ost == 0) and mine == [t[1:] for t in orecs], 'records of read %d %r differ' % (i, where[i])
        nrec += len(mine)
    assert nrec >= len(reads) and stats['n_failed'] == 0
    # the bulk sample (round 6: tools/bigverify.py's hg38-size leg promoted to this test): 2000 (ONT) / 1000 (HiFi) more
    # reads of the configuration's own read model, whole path, records vs the oracle — and the row chain kernels' rare paths (scan past the 16-entry window,
    # insertion through HBM) must have been reached by this sample with the PRODUCT's window, with the records still equal
    import os
    from vacmap_amd import synth
    from vacmap_amd.lib import chain_counters
    chain_counters(ctx.lib, 1); chain_counters(ctx.lib, -1)
    n_bulk = 2000 if shape == 'ont' else 1000
    cat, off, _ = synth.sample_reads_concat(contigs, n_bulk, mean_len=15000 if shape == 'ont' else 18000, err=0.10 if shape == 'ont' else 0.005, seed=900 + k,
                                            shape=shape, **({'min_len': 5000} if shape == 'hifi' else {}))
    bulk = [cat[off[i]:off[i + 1]].tobytes() for i in range(n_bulk)]
    bst, brecs, bstats = align_batch(ctx, gi, prm, bulk)
    ost, orecs = oracle.align_batch(oi, bulk, op, nthreads=min(os.cpu_count() or 1, 32))
    assert [(int(x) == 0) for x in bst] == [(int(x) == 0) for x in ost]
    if brecs != orecs:
        a, b = {}, {}
        for t_ in brecs: a.setdefault(t_[0], []).append(t_[1:])
        for t_ in orecs: b.setdefault(t_[0], []).append(t_[1:])
        bad = [i for i in range(n_bulk) if a.get(i) != b.get(i)]
        assert not bad, 'bulk sample: %d of %d reads differ from the oracle (first: read %d)' % (len(bad), n_bulk, bad[0])
    assert len(orecs) >= n_bulk - 8 and bstats['n_failed'] == 0
    cc = chain_counters(ctx.lib, -1)
    assert cc['global_anchors'] > 50 * n_bulk and cc['local_anchors'] > 50 * n_bulk, cc
    assert cc['global_scans_past_window'] + cc['local_scans_past_window'] > 0 and cc['global_insertions_through_hbm'] + cc['local_insertions_through_hbm'] > 0, cc
    if mode == 'H':
        # -mode asm at this size: assembly contigs with SVs against the same index — one below 500 kb (the fork's per-read
        # function), one above (the batch-linked path; noise hits outnumber the true ones nine to one here), one from the last contig (> 2^31)
        rng = np.random.default_rng(77)
        nc = len(contigs)
        asm = []
        for c, st, ln, ops, rev in ((2, 30_000_000, 180_000, [('INV', 60_000, 2500), ('DEL', 120_000, 900)], 0),
                                    (nc // 2, 12_000_000, 750_000, [('DEL', 100_000, 3000), ('INV', 330_000, 1800), ('DUP', 560_000, 1500, 2)], 1),
                                    (nc - 1, 20_000_000, 560_000, [('INS', 200_000, 700, 5), ('INV', 410_000, 3000)], 0)):
            seq = synth.mutate(synth.implant_svs(contigs[c][st:st + ln], ops), 0.003, rng)
            asm.append(synth.tostr(synth.revcomp(seq) if rev else seq))
        aprm = ctx.lib.params('asm'); aop = oracle.params('asm')
        ast, arecs, _ = align_batch(ctx, gi, aprm, asm)
        for x, cseq in enumerate(asm):
            ost, orecs = oracle.align_asm(oi, cseq, aop)
            assert ost == 0 and ast[x] == 0 and len(orecs) >= 2
            assert [t[1:] for t in arecs if t[0] == x] == [t[1:] for t in orecs], 'asm contig %d: records differ from the oracle' % x
    # a replica allocated from the metadata and filled from the builder's HBM pieces maps identically (the broadcast's receive side)
    from vacmap_amd.dist import index_blobs
    rep = Index.from_meta(ctx, gi.meta())
    for dst, src in zip(index_blobs(rep, 'cuda:0'), index_blobs(gi, 'cuda:0')):
        dst.copy_(src)
    import torch
    torch.cuda.synchronize()
    ra = ctx.map_batch(rep, reads[:6], check_num=100, mid_occ=-1)
    assert all(np.array_equal(x, y) for x, y in zip(ra, anchors[:6]))
    assert rep.seq(len(names) - 1, 1000, 1060) == contigs[-1][1000:1060].tobytes().decode()
    rep.close(); gi.close()


def test_batches_in_flight_do_not_change_a_record(ctx, hg38_ref):
    """The schedule the bench and the driver run (vacmap_amd/pipeline.py: plan_batches -> upload_batches -> Pipeline.run_resident with several contexts
    in flight, twice over so that every context meets batches of both sizes) against the same batches run one after the other on ONE context:
    every status and every record identical, batch by batch — contexts share the index and the device, nothing else"""
    import os
    from vacmap_amd import pipeline, synth
    from vacmap_amd.lib import Index
    names, contigs = hg38_ref
    gi = Index.from_seqs(ctx, names, contigs, k=15, w=10)
    prm = ctx.lib.params('H')
    n_reads, per_batch = 12288, 1024
    cat, off, _ = synth.sample_reads_concat(contigs, n_reads, mean_len=15000, err=0.10, seed=4242)
    plan = pipeline.plan_batches(np.diff(off), per_batch, 16)
    assert len(plan) == 12
    resident = pipeline.upload_batches(ctx, cat, off, plan)
    alone = [r.align(gi, prm, want_records=True, ctx=ctx) for r in resident]
    pipe = pipeline.Pipeline(gi, prm, device=0, inflight=6, first_ctx=ctx)
    try:
        pipe.warm(resident[0])
        for rep in range(2):
            got = {}
            pipe.run_resident(resident, want_records=True, on_result=lambda i, res: got.__setitem__(i, res))
            assert sorted(got) == list(range(len(plan)))
            for i in range(len(plan)):
                assert np.array_equal(got[i][0], alone[i][0]), 'pass %d batch %d: status differs with batches in flight' % (rep, i)
                assert got[i][1] == alone[i][1], 'pass %d batch %d: records differ with batches in flight' % (rep, i)
    finally:
        pipe.close()
    assert sum(len(a[1]) for a in alone) >= n_reads - 24          # (the records themselves are pinned to the oracle by the test above: same index, same read model, one context)
    for r in resident:
        r.close()
