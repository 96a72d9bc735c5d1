"""-mode asm: the oracle's restatement of mammap_asm.py against fixtures generated from the reference's own Python
(tools/harness/gen_golden_asm.py -> tests/golden/asm.npz + asm.json). CPU only."""
import json, os, zlib
import numpy as np
import pytest
import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def gold():
    return json.load(open(os.path.join(GOLD, 'asm.json'))), np.load(os.path.join(GOLD, 'asm.npz'))


def case_index(meta, arr, cid):
    c = meta[cid]
    return O.Index.from_seqs(c['names'], [arr['%s_ref%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))], k=c['k'], w=c['w'])


def check_records(ix, recs, want, tag):
    got = [[ix.names[r[1]], r[2], r[3], r[4], r[5], r[6], r[7], len(r[8]), zlib.crc32(r[8].encode())] + ([r[8]] if len(r[8]) <= 4096 else []) for r in recs]
    assert got == want, '%s: records differ from the reference' % tag


@pytest.mark.parametrize('cid', ['AS1', 'AS2', 'AS3', 'AS4', 'AS5'])
def test_asm_records(gold, cid):
    """V6a: the 9-tuples of assembly_get_readmap_DP_test (mammap_asm.py:23204), per-read function and linked path, reference sizes and shrunk ones"""
    meta, arr = gold
    ix = case_index(meta, arr, cid)
    sizes = meta[cid]['sizes']
    prm = O.params('asm')
    nrec = 0
    for ci, c in enumerate(meta[cid]['contigs']):
        seq = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
        rc, recs = O.align_asm(ix, seq, prm, *sizes)
        assert (rc == 0) == (c['status'] == 0), (cid, c['name'], rc, c.get('raised'))
        check_records(ix, recs, c['records'], '%s/%s' % (cid, c['name']))
        nrec += len(recs)
    assert nrec > 0


def test_asm_decode_hit(gold):
    """V2a: decode_hit of the fork (:21280): MAPQ, signed score, primary path; AS4 goes through its GC-fast (:20738)"""
    meta, arr = gold
    for cid in ('AS1', 'AS4', 'AS5'):
        ix = case_index(meta, arr, cid)
        prm = O.params('asm')
        O.fast_counters(reset=True)
        for ci, c in enumerate(meta[cid]['contigs']):
            seq = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
            r = O.decode_hit_asm(ix, seq, prm)
            assert r['rc'] == 0
            assert r['mapq'] == c['v2_mapq'] and r['score'] == c['v2_score'], (cid, c['name'], r['mapq'], r['score'], c['v2_mapq'], c['v2_score'])
            want = arr['%s_c%d_v2_path' % (cid, ci)]
            if c['v2_score'] != 0:
                assert np.array_equal(r['paths'][0], want), (cid, c['name'])
        if cid == 'AS4':
            assert O.fast_counters()[0] >= 3


def test_asm_linked_dp(gold):
    """VL: the linked chain DPs on the carried states the reference built (:21686 GC-exact, :21504 LC; the GC-fast :20738 / :21871 directly)"""
    meta, arr = gold
    n_full = {0: 0, 2: 0}
    for cid in ('AS2', 'AS3'):
        for c in meta[cid]['contigs']:
            for e in c['linked_calls']:
                if 'key' not in e:
                    continue
                kk = e['key']
                g, S, P, SA = O.chain_linked_raw(arr[kk + '_rows'], e['which'], int(e['kw']['kmersize']), e['kw']['skipcost'], int(e['kw']['maxdiff']),
                                                 int(e['kw']['maxgap']), e['g_max_scores'], e['g_max_index'], arr[kk + '_preS'], arr[kk + '_preP'], e['prereadloc'])
                assert g == e['g']
                assert np.array_equal(S.view(np.uint64), arr[kk + '_S'].view(np.uint64)), kk
                assert np.array_equal(P, arr[kk + '_P']) and np.array_equal(SA, arr[kk + '_Sarg']), kk
                n_full[e['which']] += 1
    assert n_full[0] >= 4 and n_full[2] >= 4
    d = meta['AS4']['direct']
    g1, S1, P1, SA1 = O.chain_linked_raw(arr['AS4_direct_first'], 1, 15, 30., 50, 1000)
    assert g1 == d['g1'] and np.array_equal(S1.view(np.uint64), arr['AS4_direct_S1'].view(np.uint64))
    assert np.array_equal(P1, arr['AS4_direct_P1']) and np.array_equal(SA1, arr['AS4_direct_SA1'])
    pre_S = arr['AS4_direct_preS']
    g2, S2, P2, SA2 = O.chain_linked_raw(arr['AS4_direct_linked'], 1, 15, 30., 50, 1000, pre_S[-1], len(pre_S) - 1, pre_S, arr['AS4_direct_preP'], d['prereadloc'])
    assert g2 == d['g2'] and np.array_equal(S2.view(np.uint64), arr['AS4_direct_S2'].view(np.uint64))
    assert np.array_equal(P2, arr['AS4_direct_P2']) and np.array_equal(SA2, arr['AS4_direct_SA2'])
