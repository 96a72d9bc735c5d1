#!/usr/bin/env python3
"""Golden fixtures for -mode asm, from the REFERENCE's own Python (src/vacmap/mammap_asm.py imported in place; build container only).

tests/golden/asm.npz + asm.json. Per case: the reference contigs, the assembly contigs ("reads" of this mode) and, per assembly contig,
  V6a  the 9-tuples assembly_get_readmap_DP_test returns (:23204) — CIGARs kept whole up to 4 kb, otherwise (length, crc32)
  V2a  decode_hit (:21280) on the contigs that take the per-read function: MAPQ, signed score, primary path
  VL   every call of the LINKED chain DPs the contig makes (:21686 GC-exact, :21871 GC-fast, :21504 LC): for the first calls that carry a state
       the whole input (rows, pre_S, pre_P, prereadloc, running maximum) and output (g_max_index, S, P, S_arg); for all calls a checksum
Cases: AS1 contigs below 500 kb (the module's get_readmap_DP_test, check_num = -1); AS2 a 600 kb contig with the reference's constants (one
first-round batch, six linked second-round batches); AS3 the same function with its three size constants shrunk IN MEMORY (split 100000, batch
6000 anchors, window 20000: many linked first-round batches, the final-flush duplicate of yield_mapinfo :22439-22443); AS4 a repeat-dense
contig (GC-fast of the fork :20738) and a direct call of the linked GC-fast (:21871) on a carried state built the way :23254-23272 builds it;
AS5 contigs that fit two near-identical copies of a segment equally well (MAPQ 0: decode_hit's edlib tie-break :21302-21326).
"""
import inspect, json, os, re, shutil, sys, tempfile, zlib
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, _HERE)
import refrun
from refrun import O
from vacmap_amd import synth

GOLD = os.path.join(_ROOT, 'tests', 'golden')
LINKED = {'linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all': 0,
          'linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all': 1,
          'linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_all': 2}


def shrink(m, split_len, batch_anchors, window):
    """the three size constants are literals inside two functions: re-execute those two definitions, IN MEMORY and inside the module's own
    namespace, with the literals replaced (nothing is written anywhere)"""
    src = inspect.getsource(m.yield_mapinfo)
    assert src.count('500000') == 1
    exec(compile(src.replace('500000', str(batch_anchors)), '<yield_mapinfo/shrunk>', 'exec'), m.__dict__)
    src = inspect.getsource(m.assembly_get_readmap_DP_test)
    assert src.count('< 500000') == 1 and src.count('batch = 100000') == 2
    src = src.replace('< 500000', '< %d' % split_len).replace('batch = 100000', 'batch = %d' % window)
    exec(compile(src, '<assembly_get_readmap_DP_test/shrunk>', 'exec'), m.__dict__)


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def run_case(cid, names, contigs, asm_contigs, arrays, meta, sizes=None, full_linked=2):
    ix = O.Index.from_seqs(names, contigs, k=15, w=10)
    al = refrun.Aligner(oracle_index=ix)
    opt = refrun.make_option('asm', eqx=True)
    opt['maxdivergence'] = np.float64(1.)                      # mammap_asm.py:23483
    ctx = refrun.RefContext('asm', al, opt)
    m = ctx.m
    if sizes:
        shrink(m, *sizes)
    from Bio.Seq import Seq
    case = {'names': names, 'k': 15, 'w': 10, 'sizes': list(sizes) if sizes else [500000, 500000, 100000], 'contigs': []}
    for ci, c in enumerate(contigs):
        arrays['%s_ref%d' % (cid, ci)] = np.frombuffer(c.encode(), dtype=np.uint8)
    work = tempfile.mkdtemp(prefix='asmwd_')
    for ri, (rname, seq) in enumerate(asm_contigs):
        key = '%s_c%d' % (cid, ri)
        arrays[key + '_seq'] = np.frombuffer(seq.encode(), dtype=np.uint8)
        rec = {'name': rname, 'len': len(seq)}
        calls = []
        orig = {n: getattr(m, n) for n in LINKED}

        def mk(n):
            def f(g_max_scores, g_max_index, pre_S, pre_P, prereadloc, one_mapinfo, **kw):
                out = orig[n](g_max_scores, g_max_index, pre_S, pre_P, prereadloc, one_mapinfo, **kw)
                g, S, P, SA = out[0], out[1], out[2], out[3]
                e = {'which': LINKED[n], 'n': int(len(one_mapinfo)), 'n_pre': int(len(pre_S)), 'g': int(g), 'kw': {a: float(b) for a, b in kw.items()},
                     'prereadloc': int(prereadloc), 'g_max_scores': float(g_max_scores), 'g_max_index': int(g_max_index),
                     'crc_rows': crc(np.asarray(one_mapinfo, np.int64)), 'crc_S': crc(np.asarray(S, np.float64)), 'crc_P': crc(np.asarray(P, np.int64)),
                     'crc_Sarg': crc(np.asarray(SA, np.int64))}
                nfull = sum(1 for x in calls if x['which'] == LINKED[n] and 'key' in x)
                if len(pre_S) > 0 and nfull < full_linked and len(one_mapinfo) <= 40000:
                    kk = '%s_vl%d' % (key, len(calls))
                    arrays[kk + '_rows'] = np.asarray(one_mapinfo, np.int64).reshape(-1, 4)
                    arrays[kk + '_preS'] = np.asarray(pre_S, np.float64); arrays[kk + '_preP'] = np.asarray(pre_P, np.int64)
                    arrays[kk + '_S'] = np.asarray(S, np.float64); arrays[kk + '_P'] = np.asarray(P, np.int64); arrays[kk + '_Sarg'] = np.asarray(SA, np.int64)
                    e['key'] = kk
                calls.append(e)
                return out
            return f
        for n in LINKED:
            setattr(m, n, mk(n))
        wd = os.path.join(work, rname) + '/'
        try:
            try:
                one = m.assembly_get_readmap_DP_test(wd, rname, seq, str(Seq(seq).reverse_complement()), len(seq), al, m.pos2contig,
                                                     ctx.contig2start, ctx.contig2seq, ctx.index2contig, opt)
                st = 0
            except Exception as e:                              # the worker logs and skips the contig (:23493-23498)
                one, st = [], -1
                rec['raised'] = repr(e)[:200]
        finally:
            for n, f in orig.items():
                setattr(m, n, f)
            shutil.rmtree(wd, ignore_errors=True)
        rec['status'] = st
        rec['records'] = []
        for t in one:
            cg = t[8]
            r = [t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), len(cg), int(zlib.crc32(cg.encode()))]
            if len(cg) <= 4096:
                r.append(cg)
            rec['records'].append(r)
        rec['linked_calls'] = calls
        if len(seq) < case['sizes'][0]:                         # V2a
            mapq, scores, path, factor = m.decode_hit(al, ctx.index2contig, seq, len(seq), ctx.contig2start, 15, ctx.contig2seq,
                                                      skipcost=(opt['golbal_skipcost'],) * 2, maxdiff=(opt['golbal_maxdiff'],) * 2, maxgap=200,
                                                      check_num=-1, c_bias=5000, bin_size=100, overlapprecentage=0.5, hastra=False, H=False, mid_occ=-1)
            rec['v2_mapq'] = int(mapq); rec['v2_score'] = float(scores)
            arrays[key + '_v2_path'] = np.array([[int(v) for v in a] for a in path], dtype=np.int64).reshape(-1, 4)
        case['contigs'].append(rec)
        print(cid, rname, len(seq), 'status', st, 'records', len(one), 'linked calls', [(c['which'], c['n'], c['n_pre']) for c in calls][:12], flush=True)
    shutil.rmtree(work, ignore_errors=True)
    meta[cid] = case
    return ctx


def main():
    arrays, meta = {}, {}
    # ---- AS1: the per-read function of the fork
    rc_ = synth.make_reference([300000, 100000], seed=201)
    ops = [('INV', 30000, 2500), ('DEL', 70000, 900), ('INS', 110000, 500, 5), ('DUP', 150000, 1800, 2), ('INVDUP', 200000, 1500), ('DEL', 250000, 300)]
    d0 = synth.implant_svs(rc_[0], ops)
    rng = np.random.default_rng(202)
    lst = []
    for i, (a, ln, e) in enumerate([(5000, 60000, 0.003), (60000, 45000, 0.01), (100000, 80000, 0.005), (140000, 30000, 0.02), (190000, 50000, 0.002),
                                    (230000, 40000, 0.008)]):
        s = synth.mutate(d0[a:a + ln], e, rng)
        if i % 2:
            s = synth.revcomp(s)
        lst.append(('ctg%d' % i, synth.tostr(s)))
    lst.append(('ctgB', synth.tostr(synth.mutate(rc_[1][10000:70000], 0.004, rng))))
    lst.append(('chim', synth.tostr(np.concatenate([synth.mutate(d0[20000:45000], 0.004, rng), synth.revcomp(synth.mutate(rc_[1][30000:52000], 0.004, rng))]))))
    lst.append(('rand', synth.tostr(synth.make_reference([8000], seed=203)[0])))
    lst.append(('short', synth.tostr(d0[120000:120900])))
    names = ['chrA', 'chrB']
    run_case('AS1', names, [synth.tostr(c) for c in rc_], lst, arrays, meta)
    # ---- AS2: the reference's own constants: a 600 kb contig (forward) over a 1.3 Mb reference
    big = synth.make_reference([1300000], seed=211)
    bd = synth.implant_svs(big[0], [('INV', 150000, 4000), ('DEL', 260000, 1500), ('DUP', 380000, 3000, 2), ('INS', 500000, 800, 9), ('INV', 610000, 600)])
    rng = np.random.default_rng(212)
    run_case('AS2', ['chrBig'], [synth.tostr(big[0])], [('ctg600k', synth.tostr(synth.mutate(bd[60000:660000], 0.004, rng)))], arrays, meta)
    # ---- AS3: shrunk constants -> many linked first-round batches
    rng = np.random.default_rng(222)
    l3 = [('ctg180k', synth.tostr(synth.mutate(d0[40000:220000], 0.006, rng))),
          ('ctg250k_rc', synth.tostr(synth.revcomp(synth.mutate(d0[30000:280000], 0.004, rng)))),
          ('ctg120k_hi', synth.tostr(synth.mutate(d0[100000:220000], 0.03, rng)))]
    run_case('AS3', names, [synth.tostr(c) for c in rc_], l3, arrays, meta, sizes=(100000, 6000, 20000))
    # ---- AS4: GC-fast of the fork. A 2.5 kb element in 64 copies: more than 5 anchors per contig base with check_num = -1
    rng = np.random.default_rng(230)
    e0 = synth.make_reference([420000], seed=231)[0]
    elem = synth.make_reference([2500], seed=232)[0]
    step = (len(e0) - 40000) // 65
    starts = []
    for t in range(64):
        p = 3000 + t * step
        e0[p:p + 2500] = synth.mutate(elem, 0.01, rng, ratio=(1, 0, 0))[:2500]
        starts.append(p)
    l4 = [('elem%d' % i, synth.tostr(synth.mutate(e0[starts[9 * i + 2] - 300 + 100 * i:starts[9 * i + 2] + 2900], 0.004, rng))) for i in range(3)]
    ctx4 = run_case('AS4', ['chrE'], [synth.tostr(e0)], l4, arrays, meta)
    # direct calls: GC-fast of the fork on the dense anchors of elem0, then the LINKED GC-fast on a state carried the way :23254-23272 carries it
    m = ctx4.m
    al = ctx4.al
    seq = l4[0][1]
    A = np.array(al.map(seq, check_num=-1, mid_occ=-1), dtype=np.int64).reshape(-1, 4)
    flag, A = m.get_reversed_chain_numpy_rough(A.copy(), len(seq))
    A = np.ascontiguousarray(A)[np.argsort(np.ascontiguousarray(A)[:, 0])]
    half = len(A) // 2
    first, second = A[:half], A[half:]
    sk, md = np.float64(30.), 50
    g1, S1, P1, SA1 = m.get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all(first, kmersize=15, skipcost=sk, maxdiff=md, maxgap=1000)
    arrays['AS4_direct_first'] = first; arrays['AS4_direct_S1'] = np.asarray(S1, np.float64); arrays['AS4_direct_P1'] = np.asarray(P1, np.int64)
    arrays['AS4_direct_SA1'] = np.asarray(SA1, np.int64)
    gms = S1[SA1[-1]]
    low = gms - sk - 36 - 20
    sl = len(S1) - 1
    while low < S1[SA1[sl]]:
        sl -= 1
        if sl == 0:
            break
    pre_S = S1[SA1[sl:]] - S1[SA1[sl]] + 1000
    pre_P = -np.asarray(P1)[SA1[sl:]]
    pre_rows = first[SA1[sl:]]
    linked = np.concatenate((pre_rows, second))
    prl = int(pre_rows[:, 0].max())
    g2, S2, P2, SA2 = m.linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all(
        pre_S[-1], len(pre_S) - 1, pre_S, pre_P, prl, linked, kmersize=15, skipcost=sk, maxdiff=md, maxgap=1000)
    arrays['AS4_direct_linked'] = linked; arrays['AS4_direct_preS'] = np.asarray(pre_S, np.float64); arrays['AS4_direct_preP'] = np.asarray(pre_P, np.int64)
    arrays['AS4_direct_S2'] = np.asarray(S2, np.float64); arrays['AS4_direct_P2'] = np.asarray(P2, np.int64); arrays['AS4_direct_SA2'] = np.asarray(SA2, np.int64)
    meta['AS4']['direct'] = {'g1': int(g1), 'g2': int(g2), 'prereadloc': prl, 'n_first': int(len(first)), 'n_linked': int(len(linked)), 'n_pre': int(len(pre_S))}
    # ---- AS5: decode_hit's edlib tie-break (:21302-21326): a 40 kb segment and a copy of it with 1 - 3 substitutions; contigs cut from inside
    # either copy (0.3 % errors, both strands) score within 0.1 % on both placements -> MAPQ 0 -> the least divergent placement wins
    l5 = []
    r5 = synth.make_reference([400000], seed=400)[0]
    seg = r5[50000:90000].copy()
    rng = np.random.default_rng(401)
    cp = seg.copy()
    for p_ in rng.integers(1000, 39000, 3):
        cp[p_] = ord('ACGT'[(b'ACGT'.index(bytes([cp[p_]])) + 1) % 4])
    r5[250000:290000] = cp
    for i, (a_, b_, rcflag) in enumerate([(50500, 89500, 0), (250800, 289000, 0), (52000, 88000, 1), (251500, 289500, 1)]):
        sq = synth.mutate(r5[a_:b_], 0.003, rng)
        if rcflag:
            sq = synth.revcomp(sq)
        l5.append(('dup%d' % i, synth.tostr(sq)))
    run_case('AS5', ['chrD'], [synth.tostr(r5)], l5, arrays, meta)
    np.savez_compressed(os.path.join(GOLD, 'asm.npz'), **arrays)
    json.dump(meta, open(os.path.join(GOLD, 'asm.json'), 'w'), indent=0, sort_keys=True)
    print('asm goldens:', {k: len(v['contigs']) for k, v in meta.items()}, 'npz bytes', os.path.getsize(os.path.join(GOLD, 'asm.npz')),
          'json bytes', os.path.getsize(os.path.join(GOLD, 'asm.json')))


if __name__ == '__main__':
    main()
