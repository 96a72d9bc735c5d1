"""CPU tests: the oracle (oracle/liboracle.so) against the golden vectors captured from the reference's own Python
(tools/harness/gen_golden.py). No GPU, no /root/reference needed."""
import hashlib, json, os, zlib
import numpy as np
import pytest

CASES = ['A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'J', 'K', 'M', 'M2', 'N', 'O', 'P']     # J-P (round 3): mode S, more L / H / R reads, reads of 40-52 kb, the last fix_simple_inv branch; E, F: repeat-dense reads that take the *_fast chain variants (G3, L5); G: mode R;
                                                       # H: mode R on a donor made by the vacsim-grammar implanter (nested INV / DUP / TRA, BASELINE configs[4])
                                                       # I: inputs that drive the rare branches of the segment surgery (V4)


def _index(O, meta, arrays, cid):
    c = meta[cid]
    contigs = [arrays['%s_contig%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))]
    return O.Index.from_seqs(c['names'], contigs, k=c['k'], w=c['w'])


def _seq(arrays, cid, ri):
    return arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode()


def test_tables_bit_identical(oracle):
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'tables.json')))
    which = {'extra': 0, 'readgap_h': 1, 'readgap_r': 2, 'large_readgap': 3, 'log2cache': 4, 'log2int': 5}
    for name, w in which.items():
        t = oracle.table(w)
        g = gold[name]
        assert str(t.dtype) == g['dtype'] and len(t) == g['len']
        assert hashlib.sha256(t.tobytes()).hexdigest() == g['sha256'], name


@pytest.mark.parametrize('cid', CASES)
def test_map_is_stable(oracle, golden, cid):
    """own-spec map(): the anchors stored in the fixture (inputs of the reference run) are reproduced"""
    meta, arrays = golden
    ix = _index(oracle, meta, arrays, cid)
    for ri, r in enumerate(meta[cid]['reads']):
        a = ix.map(_seq(arrays, cid, ri))
        assert np.array_equal(a, arrays['%s_r%d_anchors' % (cid, ri)].reshape(-1, 4))


@pytest.mark.parametrize('cid', CASES)
def test_v1_strand_flip(oracle, golden, cid):
    meta, arrays = golden
    for ri, r in enumerate(meta[cid]['reads']):
        a = arrays['%s_r%d_anchors' % (cid, ri)].reshape(-1, 4)
        flag, out = oracle.strand_flip(a, r['len'])
        assert flag == r['v1_flag']
        assert np.array_equal(out, arrays['%s_r%d_v1' % (cid, ri)].reshape(-1, 4))


@pytest.mark.parametrize('cid', CASES)
def test_v2_global_chain(oracle, golden, cid):
    meta, arrays = golden
    c = meta[cid]
    prm = oracle.params(c['mode'])
    for ri, r in enumerate(c['reads']):
        key = '%s_r%d' % (cid, ri)
        a = arrays[key + '_anchors'].reshape(-1, 4)
        if 'v2_gmax' in r:
            fl = arrays[key + '_v1'].reshape(-1, 4)
            srt = fl[np.argsort(fl[:, 0], kind='stable')]
            g, S, P, SA = oracle.chain_global_raw(srt, c['k'], prm.global_skipcost, prm.global_maxdiff, 1000, 0, c['mode'])
            assert g == r['v2_gmax']
            assert np.array_equal(S.view(np.uint64), arrays[key + '_v2_S'].view(np.uint64)), 'S not bit-identical'
            assert np.array_equal(P, arrays[key + '_v2_P'])
            assert np.array_equal(SA, arrays[key + '_v2_Sarg'])
        if 'v2f_gmax' in r:      # GC-fast raw arrays (:25033-25339): S_arg ordered by (int(S), diagonal key)
            fl = arrays[key + '_v1'].reshape(-1, 4)
            srt = fl[np.argsort(fl[:, 0], kind='stable')]
            g, S, P, SA = oracle.chain_global_raw(srt, c['k'], prm.global_skipcost, prm.global_maxdiff, 1000, 1, c['mode'])
            assert g == r['v2f_gmax']
            assert np.array_equal(S.view(np.uint64), arrays[key + '_v2f_S'].view(np.uint64)), 'GC-fast S not bit-identical'
            assert np.array_equal(P, arrays[key + '_v2f_P'])
            assert np.array_equal(SA, arrays[key + '_v2f_Sarg'])
        res = oracle.decode_hit(a, r['len'], c['k'], prm)
        if r.get('v2_raised'):       # mode R: the reference raises (unbound `factor`) when <= 2 anchors survive
            assert res['rc'] < 0
            continue
        assert res['rc'] == 0
        assert res['mapq'] == r['v2_mapq']
        assert res['score'] == r['v2_score']
        assert [p.tolist() for p in res['paths']] == r['v2_paths']


@pytest.mark.parametrize('cid', CASES)
def test_v3_local_chain(oracle, golden, cid):
    meta, arrays = golden
    c = meta[cid]
    prm = oracle.params(c['mode'])
    ix = _index(oracle, meta, arrays, cid)
    comp = bytes.maketrans(b'ACGTN', b'TGCAN')
    for ri, r in enumerate(c['reads']):
        key = '%s_r%d' % (cid, ri)
        if 'v3_score' not in r:
            continue
        seq = _seq(arrays, cid, ri)
        rd = seq if r['v2_score'] > 0 else seq.encode().translate(comp)[::-1].decode()
        res = oracle.local_chain(ix, rd, [np.array(p, dtype=np.int64) for p in r['v2_paths']], prm)
        assert res['rc'] == 0
        raw = res['raw']
        raw = raw[np.argsort(raw[:, 0] + (0 if c['mode'] == 'R' else raw[:, 3]), kind='stable')]     # the DP's input order (:28585; mode R: by read start)
        assert len(raw) == r.get('v3_raw_n', len(raw))
        if key + '_v3_raw' in arrays:
            assert np.array_equal(raw, arrays[key + '_v3_raw'].reshape(-1, 4)), 'local anchors differ'
        else:                    # dense cases carry a checksum of the raw local anchors instead of the rows
            assert zlib.crc32(np.ascontiguousarray(raw.astype(np.int64)).tobytes()) == r['v3_raw_crc'], 'local anchors differ'
        assert res['variant'] == r['v3_variant']
        assert res['score'] == r['v3_score']
        assert np.array_equal(res['chain'], arrays[key + '_v3_path'].reshape(-1, 4))


@pytest.mark.parametrize('cid', CASES)
def test_v5_v6_records_and_dp_problems(oracle, golden, cid):
    meta, arrays = golden
    c = meta[cid]
    prm = oracle.params(c['mode'])
    ix = _index(oracle, meta, arrays, cid)
    for ri, r in enumerate(c['reads']):
        seq = _seq(arrays, cid, ri)
        (st, recs), calls = oracle.dplog(lambda: oracle.align_read(ix, seq, prm))
        assert (st == 0) == (r['v6_status'] == 0)
        got = [[c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]] for t in recs]
        assert got == r['v6_records'], (cid, ri)
        dp = [[kd, len(t), len(q), zlib.crc32(t.encode()), zlib.crc32(q.encode())] for kd, t, q in calls]
        assert dp == r['v5'], (cid, ri)


_V4_FN = {'rebuild_chain_break': 0, 'drop_misplaced_alignment_test': 1, 'merge_conjacent_alignment': 2, 'fix_simple_inv': 3}


@pytest.mark.parametrize('cid', ['A', 'B', 'C', 'D', 'G', 'H', 'I', 'N'])
def test_v4_segment_surgery(oracle, golden, cid):
    """stage vectors V4: every call the reference made to rebuild_chain_break (:23437), drop_misplaced_alignment_test (:726),
    merge_conjacent_alignment (:16736, getdupiloc_numba :16680 inside) and fix_simple_inv (:24226) while aligning the read — the oracle's
    functions reproduce each output from the captured input, so a surgery bug is localised to its function"""
    meta, arrays = golden
    c = meta[cid]
    ix = _index(oracle, meta, arrays, cid)
    ncall = {k: 0 for k in _V4_FN}
    for ri, r in enumerate(c['reads']):
        seq = _seq(arrays, cid, ri)
        need_rev = r.get('v2_score', 0) < 0
        rd = seq if not need_rev else seq.encode().translate(bytes.maketrans(b'ACGTN', b'TGCAN'))[::-1].decode()
        for e in r.get('v4', []):
            if 'in' not in e:
                continue
            fn = _V4_FN[e['fn']]
            arg = e.get('large_cost', 0) if fn == 0 else (e.get('iloc', 0) if fn == 1 else (1 if c['mode'] == 'R' else 0))
            rc, ret, out = oracle.stage_v4(ix, fn, arrays[e['in']], arg=arg, read=rd)
            assert rc == 0, (cid, ri, e['fn'])
            assert np.array_equal(out, arrays[e['out']].reshape(-1, 5)), (cid, ri, e['fn'], e.get('iloc'))
            if fn == 1:
                assert bool(ret) == e['removed']
            ncall[e['fn']] += 1
    assert ncall['rebuild_chain_break'] >= len([r for r in c['reads'] if r['v6_records']]) and ncall['fix_simple_inv'] >= 1


def test_v4_covers_every_branch(golden):
    """the captured calls exercise what they are meant to pin: at least one drop_misplaced removal, one merge that merges, one
    fix_simple_inv that moves a breakpoint"""
    meta, arrays = golden
    seen = {'drop': 0, 'merge': 0, 'fix': 0, 'fix_left_flank': 0}
    for cid in ('A', 'B', 'C', 'D', 'G', 'H', 'I', 'N'):
        for r in meta[cid]['reads']:
            for e in r.get('v4', []):
                if 'in' not in e:
                    continue
                changed = not np.array_equal(arrays[e['in']], arrays[e['out']])
                if e['fn'] == 'drop_misplaced_alignment_test' and e['removed']:
                    seen['drop'] += 1
                if e['fn'] == 'merge_conjacent_alignment' and changed:
                    seen['merge'] += 1
                if e['fn'] == 'fix_simple_inv' and changed:
                    seen['fix'] += 1
                    # the `refen_0 < refst_1` branch (:24297-24310) moves the LAST (zero-length) anchor of the left flank forward
                    # (the other branch rewrites the first anchor of the right flank)
                    i_, o_ = arrays[e['in']].reshape(-1, 5), arrays[e['out']].reshape(-1, 5)
                    for sg in np.unique(o_[:, 0]):
                        li, lo = i_[i_[:, 0] == sg], o_[o_[:, 0] == sg]
                        if len(li) and len(lo) and lo[-1, 3] == 1 and lo[-1, 4] == 0 and lo[-1, 1] > li[-1, 1] and sg + 2 <= o_[:, 0].max():
                            seen['fix_left_flank'] += 1
    assert seen['drop'] >= 3 and seen['merge'] >= 1 and seen['fix'] >= 4 and seen['fix_left_flank'] >= 1, seen


def test_golden_pin_is_wide(golden):
    """at least 150 reads from the imported reference, every mode, reads of 40 kb and more"""
    meta, arrays = golden
    cases = {k: v for k, v in meta.items() if isinstance(v, dict)}
    assert sum(len(c['reads']) for c in cases.values()) >= 150
    assert {c['mode'] for c in cases.values()} == {'H', 'L', 'S', 'R'}
    assert sum(len(c['reads']) for c in cases.values() if c['mode'] == 'S') >= 20
    assert sum(len(c['reads']) for c in cases.values() if c['mode'] == 'L' and c['k'] == 19) >= 30
    assert sum(1 for c in cases.values() for r in c['reads'] if r['len'] >= 40000) >= 3


def test_testdata_three_alignments(oracle, golden):
    """README.md:124 — testdata yields three alignments (+, -, +: a 16 kb inversion)"""
    meta, arrays = golden
    recs = meta['A']['reads'][0]['v6_records']
    assert [r[1] for r in recs] == ['+', '-', '+']
    assert len(recs) == 3


def test_primitives_properties(oracle):
    rng = np.random.default_rng(5)
    for _ in range(30):
        n, m = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        a = ''.join('ACGT'[i] for i in rng.integers(0, 4, n)); b = ''.join('ACGT'[i] for i in rng.integers(0, 4, m))
        # edit distance vs the textbook DP
        D = np.arange(m + 1)
        for i in range(1, n + 1):
            prev = D.copy(); D[0] = i
            for j in range(1, m + 1):
                D[j] = min(prev[j] + 1, D[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        assert oracle.edit_distance(a, b) == D[m]
        # CIGAR consumes both sequences exactly and re-scores to the reported score
        cg, sc = oracle.k_cigar_global(a, b, eqx=True)
        import re
        i = j = 0; s = 0
        for num, op in re.findall(r'(\d+)([=XID])', cg):
            num = int(num)
            if op in '=X':
                for x in range(num):
                    assert (a[i + x] == b[j + x]) == (op == '=')
                s += (2 if op == '=' else -4) * num; i += num; j += num
            elif op == 'D':
                s -= min(4 + 2 * num, 24 + num); i += num
            else:
                s -= min(4 + 2 * num, 24 + num); j += num
        assert (i, j) == (n, m) and s == sc
    assert oracle.edit_distance('', 'ACG') == 3
    sc, te, qe = oracle.k_extend('ACGTACGTAC', 'ACGTACGTAC')
    assert (sc, te, qe) == (20, 10, 10)
    sc, te, qe = oracle.k_extend('', 'ACGT')
    assert (sc, te, qe) == (0, 0, 0)


def test_gapfill_tie_break_is_published_ksw2_order(oracle):
    """VMX-DP-G's choice among co-optimal alignments follows minimap2's ksw_extd2 (left-aligned): diagonal > E1 > F1 > E2 > F2.
    The oracle must equal an independent Python DP written to that order on constructed exact ties (a long-piece deletion next to a
    short-piece insertion), and the ties must resolve as 'deletion, then insertion' in the CIGAR."""
    import kernel_cases as KC
    n_di = 0
    for t, q in KC.gapfill_tie_cases():
        exp, sc = KC.ksw2_order_cigar(t, q)
        got, gsc = oracle.k_cigar_global(t, q)
        assert (got, gsc) == (exp, sc), (t, q)
        if len(t) > len(q):
            assert 'D' in got and 'I' in got and got.index('D') < got.index('I'), got
            n_di += 1
    assert n_di >= 15
    rng = np.random.default_rng(77)                      # and on ordinary small problems
    for _ in range(40):
        a = KC.rand_seq(rng, int(rng.integers(1, 60)))
        b = KC.mutate(rng, a, 0.25) or 'A'
        assert oracle.k_cigar_global(a, b) == KC.ksw2_order_cigar(a, b)
