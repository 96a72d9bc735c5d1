"""Shared parity checks: the HIP path (real library on a GPU, or its emulator build in CPU tests) against the oracle.
`ctx` is a vacmap_amd.lib.Context; `O` is tests/oracle_lib."""
import re
import numpy as np


def rand_seq(rng, n):
    return ''.join('ACGT'[i] for i in rng.integers(0, 4, n))


def mutate(rng, s, rate):
    out = []
    for ch in s:
        u = rng.random()
        if u < rate * 0.4:
            out.append('ACGT'[rng.integers(0, 4)])
        elif u < rate * 0.7:
            continue
        elif u < rate:
            out.append(ch); out.append('ACGT'[rng.integers(0, 4)])
        else:
            out.append(ch)
    return ''.join(out)


def check_tables(ctx, O):
    for w in range(6):
        a, b = ctx.table(w), O.table(w)
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes(), 'table %d' % w


def check_edit_distance(ctx, O, n=32, maxlen=600, seed=1, minlen=0):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for i in range(n):
        L = int(rng.integers(minlen, maxlen)) if i else max(minlen, 1)
        a = rand_seq(rng, L)
        b = mutate(rng, a, float(rng.choice([0.0, 0.05, 0.2, 0.6])))
        if i == 1:
            b = ''
        if i == 2:
            a = a[:1]
        if i == 3:
            a = a.replace('A', 'N', 2)
        qs.append(a); ts.append(b)
    got = ctx.edit_distance_batch(qs, ts)
    exp = [O.edit_distance(q, t) for q, t in zip(qs, ts)]
    assert got.tolist() == exp


def check_edit_distance_bound(ctx, O, seed=5, lens=(1, 63, 64, 65, 700, 1600, 3000, 6500), big=True, tier=2):
    """banded upper bound of the divergence filter: exact on near-diagonal pairs, never below the exact distance,
    -1 outside its eligibility window"""
    rng = np.random.default_rng(seed)
    qs, ts, kind = [], [], []
    for L in lens:
        for rate in (0.0, 0.1, 0.25):
            a = rand_seq(rng, L)
            b = mutate(rng, a, rate)
            if abs(len(a) - len(b)) > (500 if tier == 2 else 250) or not b:
                continue
            qs.append(a); ts.append(b); kind.append('exact')
            qs.append(b); ts.append(a); kind.append('exact')
    # an empty side, a one-sided length excess, unrelated sequences, and a path that leaves the band and comes back
    qs.append(''); ts.append('ACGT'); kind.append('exact')
    qs.append('ACGTN'); ts.append(''); kind.append('exact')
    a = rand_seq(rng, 1200); qs.append(a); ts.append(a[:600]); kind.append('inelig')
    qs.append(a[:500]); ts.append(a[:1100]); kind.append('inelig')
    qs.append(rand_seq(rng, 900)); ts.append(rand_seq(rng, 1000)); kind.append('bound')
    if big:
        a = rand_seq(rng, 5000)
        ins = rand_seq(rng, 1100)
        b = a[:1000] + ins + a[1000:3000] + a[4000:]         # +1100 then -1000: the optimal path runs 1100 rows off the diagonal
        qs.append(a); ts.append(b); kind.append('bound')
        qs.append(b); ts.append(a); kind.append('bound')
        a = rand_seq(rng, 4300); b = mutate(rng, a, 0.12)     # > 64 blocks: lanes are reused
        if abs(len(a) - len(b)) <= (500 if tier == 2 else 250):
            qs.append(a); ts.append(b); kind.append('exact')
    got = ctx.edit_distance_bound_batch(qs, ts, tier=tier).tolist()
    for q, t, k, g in zip(qs, ts, kind, got):
        e = O.edit_distance(q, t)
        if k == 'exact':
            assert g == e, (len(q), len(t), g, e)
        elif k == 'inelig':
            assert g == -1, (len(q), len(t), g)
        else:
            assert g >= e, (len(q), len(t), g, e)
    return got


def check_extend(ctx, O, n=64, seed=3, maxlen=700):
    rng = np.random.default_rng(seed)
    ts, qs = [], []
    for i in range(n):
        L = int(rng.integers(1, maxlen))
        a = rand_seq(rng, L)
        b = mutate(rng, a, float(rng.choice([0.0, 0.1, 0.3])))
        cut = int(rng.integers(0, len(b) + 1))
        b = b[:cut] + rand_seq(rng, int(rng.integers(0, 300)))   # diverges -> x-drop must stop
        if i == 0:
            a = ''
        if i == 1:
            b = ''
        ts.append(a); qs.append(b)
    sc, te, qe = ctx.k_extend_batch(ts, qs)
    for i in range(n):
        e = O.k_extend(ts[i], qs[i])
        assert (int(sc[i]), int(te[i]), int(qe[i])) == e, (i, len(ts[i]), len(qs[i]))


def check_gapfill(ctx, O, n=64, maxlen=500, seed=4, minlen=1):
    rng = np.random.default_rng(seed)
    ts, qs = [], []
    for i in range(n):
        L = int(rng.integers(minlen, maxlen))
        a = rand_seq(rng, L)
        b = mutate(rng, a, float(rng.choice([0.0, 0.1, 0.3])))
        if i % 7 == 3 and len(b) > 40:   # long gap -> second affine piece
            b = b[:10] + b[10 + 30:]
        if i % 7 == 5:
            b = b[:5] + rand_seq(rng, 40) + b[5:]
        if not b:
            b = 'A'
        ts.append(a); qs.append(b)
    for eqx in (False, True):
        cg, sc = ctx.k_cigar_batch(ts, qs, eqx=eqx)
        for i in range(n):
            e_cg, e_sc = O.k_cigar_global(ts[i], qs[i], eqx=eqx)
            assert int(sc[i]) == e_sc, (i, 'score')
            assert cg[i] == e_cg, (i, len(ts[i]), len(qs[i]))


def ksw2_order_cigar(t, q, match=2, mis=-4, o1=4, e1=2, o2=24, e2=1):
    """independent pure-Python restatement of VMX-DP-G with the PUBLISHED ksw2 (ksw_extd2, left-aligned) priorities: the source of H is
    the first of diagonal > E1 (deletion, short piece) > F1 (insertion, short piece) > E2 > F2 that is strictly larger than the ones
    before it; a gap state continues iff its extension is strictly better than a new opening. Small inputs only."""
    NEG = -10 ** 9
    tl, ql = len(t), len(q)
    mk = lambda: [[NEG] * (ql + 1) for _ in range(tl + 1)]
    H, E1, E2, F1, F2 = mk(), mk(), mk(), mk(), mk()
    src, x1, x2, y1, y2 = mk(), mk(), mk(), mk(), mk()
    H[0][0] = 0
    for i in range(tl + 1):
        for j in range(ql + 1):
            if i == 0 and j == 0:
                continue
            if i > 0:
                a1, a2 = H[i - 1][j] - o1, H[i - 1][j] - o2
                x1[i][j] = E1[i - 1][j] > a1; x2[i][j] = E2[i - 1][j] > a2
                E1[i][j] = max(a1, E1[i - 1][j]) - e1; E2[i][j] = max(a2, E2[i - 1][j]) - e2
            if j > 0:
                c1, c2 = H[i][j - 1] - o1, H[i][j - 1] - o2
                y1[i][j] = F1[i][j - 1] > c1; y2[i][j] = F2[i][j - 1] > c2
                F1[i][j] = max(c1, F1[i][j - 1]) - e1; F2[i][j] = max(c2, F2[i][j - 1]) - e2
            d = NEG
            if i > 0 and j > 0:
                d = H[i - 1][j - 1] + (match if (t[i - 1] == q[j - 1] and t[i - 1] in 'ACGT') else mis)
            h, sr = d, 0
            for kk, v in ((1, E1[i][j]), (3, F1[i][j]), (2, E2[i][j]), (4, F2[i][j])):       # state codes: 1 E1, 2 E2, 3 F1, 4 F2
                if v > h:
                    h, sr = v, kk
            H[i][j] = h; src[i][j] = sr
    ops = []; i, j, st = tl, ql, 0
    while i > 0 or j > 0:
        if st == 0:
            sr = src[i][j]
            if sr == 0:
                ops.append('M'); i -= 1; j -= 1
            else:
                st = sr
        elif st in (1, 2):
            ext = x1[i][j] if st == 1 else x2[i][j]
            ops.append('D'); i -= 1
            if not ext:
                st = 0
        else:
            ext = y1[i][j] if st == 3 else y2[i][j]
            ops.append('I'); j -= 1
            if not ext:
                st = 0
    ops.reverse()
    out = []; a = 0
    while a < len(ops):
        b = a
        while b < len(ops) and ops[b] == ops[a]:
            b += 1
        out.append('%d%s' % (b - a, ops[a])); a = b
    return ''.join(out), H[tl][ql]


def gapfill_tie_cases(seed=7):
    """problems with an EXACT tie between a long-piece deletion (E2) and a short-piece insertion (F1) at the same cell: a deletion of
    Ld >= 21 bases next to an insertion of Li <= 12 bases of an alphabet the deleted bases do not share ("delete then insert" and "insert
    then delete" cost the same). Published ksw2 resolves it as F1 before E2, i.e. the CIGAR reads ..D..I..."""
    import random
    rng = random.Random(seed)
    out = []
    for Ld, Li in ((40, 10), (45, 6), (60, 12), (33, 8), (30, 7), (80, 11)):
        for rep in range(3):
            P = ''.join(rng.choice('ACGT') for _ in range(20 + 7 * rep)); S = ''.join(rng.choice('ACGT') for _ in range(25 + 5 * rep))
            X = ''.join(rng.choice('AC') for _ in range(Ld)); Y = ''.join(rng.choice('GT') for _ in range(Li))
            out.append((P + X + S, P + Y + S))
            out.append((P + Y + S, P + X + S))           # the mirrored problem: a long insertion (F2) against a short deletion (E1)
    return out


def check_gapfill_ties(ctx, O):
    """E5 tie order on the device: identical to the oracle on the constructed E2 = F1 ties, through both schedules"""
    cases = gapfill_tie_cases()
    ts = [t for t, _ in cases]; qs = [q for _, q in cases]
    exp = [O.k_cigar_global(t, q)[0] for t, q in cases]
    assert sum(1 for e in exp if 'D' in e and 'I' in e and e.index('D') < e.index('I')) >= len(cases) // 2 - 2
    cg, _ = ctx.k_cigar_batch(ts, qs)
    assert cg == exp
    cg2, _, _ = ctx.k_cigar_batch_banded(ts, qs)
    assert cg2 == exp


AD_NS_MAX = 4       # mirrors of vmx_kernels.h: vmx_ad_geom / vmx_ad_margin / vmx_ad_ns (anti-diagonal band form of the gap fill)


def ad_geom(tl, ql, ns):
    """(g, dlo): a path leaving the band of 32 * ns diagonals holds >= g inserted and g deleted bases; g = 0: the band cannot hold the problem"""
    dl = ql - tl; lo = min(0, dl); hi = max(0, dl)
    slack = 32 * ns - (hi - lo + 1)
    if slack < 0:
        return 0, 0
    mb = slack // 2; dlo = lo - mb
    if dlo & 1:
        if mb + 1 <= slack:
            mb += 1; dlo -= 1
        elif mb >= 1:
            mb -= 1; dlo += 1
        else:
            return 0, 0
    return min(mb, slack - mb) + 1, dlo


def ad_margin(g, match=2, o1=4, e1=2, o2=24, e2=1):
    return match * g + 2 * min(o1 + g * e1, o2 + g * e2)


def ad_ns(tl, ql, pct=90, pct_min=65):
    if tl <= 0 or ql <= 0:
        return 0
    mn = min(tl, ql); g = 0
    for ns in range(1, AD_NS_MAX + 1):
        g, _ = ad_geom(tl, ql, ns)
        if g >= 1 and (g > mn or ad_margin(g) * 100 >= pct * mn):
            return ns
    return AD_NS_MAX if (g >= 1 and ad_margin(g) * 100 >= pct_min * mn) else 0


def gapfill_banded_cases(rng, x4_max, dp16_max, base_len, big=True, pct=90, pct_min=65):
    """adversarial (target, query) pairs for the banded gap fill (k_gapfill_fill_ns): see check_gapfill_banded"""
    ts, qs, tag = [], [], []

    def add(t, q, what):
        ts.append(t); qs.append(q if q else 'A'); tag.append(what)
    L = base_len
    ns0 = max(1, ad_ns(L, L, pct, pct_min))
    g0 = ad_geom(L, L, ns0)[0]                       # margin of the band a square problem of this size gets
    # |tl - ql| from 0 to beyond the widest band: one deletion / insertion of d bases in the middle, light substitutions around it
    for d in list(range(0, 24)) + list(range(24, 32 * AD_NS_MAX + 12, 3)):
        a = rand_seq(rng, L + (d if d > L // 2 else 0))
        add(a, mutate(rng, a[:L // 2] + a[L // 2 + d:], 0.02), 'del%d' % d)
        add(a, mutate(rng, a[:L // 2] + rand_seq(rng, d) + a[L // 2:], 0.02), 'ins%d' % d)
    # indels around the band's margin — just inside, on it, just beyond — at the start, the middle and the end
    for d in sorted(set([max(1, g0 // 2), g0 - 3, g0 - 2, g0 - 1, g0, g0 + 1, g0 + 3, 2 * g0 - 2, 2 * g0 - 1, 2 * g0, 2 * g0 + 1])):
        if d < 1 or d + 4 >= L:
            continue
        for pos in (2, L // 2, L - 2 - d):
            a = rand_seq(rng, L)
            add(a, a[:pos] + a[pos + d:], 'D%d@%d' % (d, pos))
            add(a, a[:pos] + rand_seq(rng, d) + a[pos:], 'I%d@%d' % (d, pos))
            add(a, mutate(rng, a[:pos] + a[pos + d:], 0.1), 'D%d@%d+err' % (d, pos))
    # two opposite gaps: the optimal path leaves the main line by g bases and comes back (tl == ql: the band's margin decides whether it
    # stays inside); sizes around every band width's margin, incl. the second affine piece (24 + g < 4 + 2 g for g > 20)
    gs = set([21, 22, 30])
    for ns in range(1, AD_NS_MAX + 1):
        g = ad_geom(L, L, ns)[0]
        gs.update([g - 2, g - 1, g, g + 1])
    for g in sorted(gs):
        if g < 1 or 3 * g + 6 >= L:
            continue
        a = rand_seq(rng, L)
        add(a, a[:L // 3] + a[L // 3 + g:], 'piece2 del %d' % g)
        add(a, a[:L // 3] + a[L // 3 + g:2 * L // 3] + rand_seq(rng, g) + a[2 * L // 3:], 'del+ins %d' % g)
        add(a, a[:L // 3] + rand_seq(rng, g) + a[L // 3:2 * L // 3] + a[2 * L // 3 + g:], 'ins+del %d' % g)
    # tandem repeats: many co-optimal paths far from the main line, the traceback's tie-breaks decide
    for unit in (1, 2, 3, 7, 11):
        u = rand_seq(rng, unit)
        a = (u * (L // unit + 2))[:L]
        add(a, a[unit * 2:] if len(a) > unit * 2 else a, 'tandem%d shift' % unit)
        add(a, mutate(rng, a, 0.05), 'tandem%d err' % unit)
    # unrelated sequences (every path is bad; the proof must fail or the result must still be the optimum)
    for _ in range(4):
        add(rand_seq(rng, L), rand_seq(rng, L + int(rng.integers(-5, 6))), 'unrelated')
    # sizes around the switches of the band-width rule (ns 1 -> 2 -> 3 -> 4 -> tried anyway -> not tried)
    sw = [q for q in range(8, x4_max // 2 + 2) if ad_ns(q, q, pct, pct_min) != ad_ns(q + 1, q + 1, pct, pct_min)]
    for q0 in sw:
        for q in (q0 - 1, q0, q0 + 1, q0 + 2):
            a = rand_seq(rng, q)
            add(a, mutate(rng, a, 0.08), 'switch%d' % q)
            add(a, mutate(rng, a, 0.01), 'switch%d clean' % q)
    # very oblong problems: the band cannot hold both corners, or g exceeds the short side (nothing can leave the band: proven outright)
    a = rand_seq(rng, L)
    add(a, a[:L // 4], 'oblong'); add(a[:L // 4], a, 'oblong2'); add(a[:3], a[:40], 'oblong3'); add(a[:50], a[:2], 'oblong4')
    # size-class boundaries of the three layouts: small class (tl + ql <= x4_max), packed int16 (<= dp16_max), int32 beyond
    if big:
        for tot in (x4_max - 1, x4_max, x4_max + 1, dp16_max - 1, dp16_max, dp16_max + 1):
            tl = tot // 2; ql = tot - tl
            a = rand_seq(rng, tl)
            b = mutate(rng, a, 0.06)
            while len(b) < ql:
                b += rand_seq(rng, 1)
            add(a, b[:ql], 'tot%d' % tot)
    # ambiguous bases on both sides: an N never matches, not even an N in the same column (all layouts must agree)
    for Ln in (L, L // 2, 3 * L):
        a = list(rand_seq(rng, Ln))
        for p0 in (3, Ln // 2, Ln - 6):
            a[p0:p0 + 3] = 'NNN'
        a = ''.join(a)
        add(a, a, 'N both %d' % Ln); add(a, mutate(rng, a, 0.05), 'N both err %d' % Ln); add(a.replace('N', 'A'), a, 'N query %d' % Ln)
    # empty / one-base sides
    add('', 'ACGT', 'empty t'); add('ACGT', 'A', 'tiny q'); add('A', 'A', '1x1'); add('A', 'C', '1x1 mismatch'); add('AC', 'A', '2x1')
    return ts, qs, tag


def check_gapfill_banded(ctx, O, x4_max, dp16_max, base_len, seed=44, big=True, min_counts=(10, 10, 5), pct=90, pct_min=65, redo_pk_min=384):
    """E5 through the schedule of the batched path (vm_k_cigar_batch_banded -> k_gapfill_fill_ns: anti-diagonal band fill of eight problems
    per wave, optimality proof, redo queue, per-problem layout flag read by k_gapfill_trace) vs the oracle's full DP (mammap_clrnano.py:21554,
    :21598 call sites): identical CIGARs with eqx on and off, in shuffled order (waves mix proven, unproven, never-tried and idle rows and
    band widths), and both branches provably taken. pct / pct_min: the band-width rule the library is running with (VMX_AD_PCT, VMX_AD_PCT_MIN)."""
    rng = np.random.default_rng(seed)
    ts, qs, tag = gapfill_banded_cases(rng, x4_max, dp16_max, base_len, big=big, pct=pct, pct_min=pct_min)
    perm = rng.permutation(len(ts))
    ts = [ts[i] for i in perm]; qs = [qs[i] for i in perm]; tag = [tag[i] for i in perm]
    expect = {eqx: [O.k_cigar_global(t, q, eqx=eqx)[0] for t, q in zip(ts, qs)] for eqx in (False, True)}
    small = [bool(t) and bool(q) and len(t) + len(q) <= x4_max for t, q in zip(ts, qs)]
    want_ns = [ad_ns(len(t), len(q), pct, pct_min) if sm else 0 for t, q, sm in zip(ts, qs, small)]
    for eqx in (False, True):
        cg, flag, st = ctx.k_cigar_batch_banded(ts, qs, eqx=eqx)
        for i in range(len(ts)):
            assert cg[i] == expect[eqx][i], (tag[i], len(ts[i]), len(qs[i]), int(flag[i]), eqx)
            # layout flags: 0 striped four-per-wave, 1 whole-wave packed (the second launch's larger problems), 16 + ns anti-diagonal band
            assert int(flag[i]) == 0 or (want_ns[i] > 0 and 16 + want_ns[i] <= int(flag[i]) <= 16 + AD_NS_MAX) or \
                (int(flag[i]) == 1 and small[i] and len(ts[i]) + len(qs[i]) >= redo_pk_min), (tag[i], int(flag[i]), want_ns[i])
        elig = sum(1 for w in want_ns if w > 0)
        assert st['eligible'] == elig and st['not_eligible'] == len(ts) - sum(small), (st, elig, sum(small))
        assert st['proven'] == int((flag > 16).sum()) and st['redo'] == sum(small) - st['proven'], st
        assert st['proven'] >= min_counts[0] and st['redo'] >= min_counts[1] and st['not_eligible'] >= min_counts[2], st      # both branches and the plain forms ran
        # the full-matrix entry agrees too (same problems, no band)
        cg0, _ = ctx.k_cigar_batch(ts, qs, eqx=eqx)
        assert cg0 == cg
    # tiny batches: idle rows in the only wave
    for n in (1, 2, 3, 5, 9):
        cg, _, _ = ctx.k_cigar_batch_banded(ts[:n], qs[:n])
        assert cg == expect[False][:n]
    st['ns_kept'] = sorted(set(int(f) - 16 for f in flag if f > 16))
    st['redo_packed'] = int((flag == 1).sum())
    return st


def check_chain_global_golden(ctx, O, golden, cases=('A', 'B', 'C', 'D')):
    meta, arrays = golden
    for cid in cases:
        c = meta[cid]
        prm = ctx.lib.params(c['mode'])
        oprm = O.params(c['mode'])
        al, rl = [], []
        for ri, r in enumerate(c['reads']):
            al.append(arrays['%s_r%d_anchors' % (cid, ri)].reshape(-1, 4)); rl.append(r['len'])
        res = ctx.chain_global_batch(prm, c['k'], al, rl, want_raw=True)
        for ri, r in enumerate(c['reads']):
            key = '%s_r%d' % (cid, ri)
            g = res[ri]
            assert g['need_reverse'] == r['v1_flag'], key
            if 'v2f_gmax' in r:          # the reference took GC-fast (:23570 / :23577): its raw arrays are the ones that count
                assert g['fast_used'] == 1, key
                assert g['gmax'] == r['v2f_gmax'], key
                assert np.array_equal(g['S'].view(np.uint64), arrays[key + '_v2f_S'].view(np.uint64)), key + ' GC-fast S'
                assert np.array_equal(g['P'], arrays[key + '_v2f_P']), key + ' GC-fast P'
                assert np.array_equal(g['S_arg'], arrays[key + '_v2f_Sarg']), key + ' GC-fast S_arg'
            elif 'v2_gmax' in r:
                assert g['gmax'] == r['v2_gmax'], key
                assert np.array_equal(g['S'].view(np.uint64), arrays[key + '_v2_S'].view(np.uint64)), key + ' S'
                assert np.array_equal(g['P'], arrays[key + '_v2_P']), key + ' P'
                assert np.array_equal(g['S_arg'], arrays[key + '_v2_Sarg']), key + ' S_arg'
            assert g['mapq'] == r['v2_mapq'], key
            assert g['score'] == r['v2_score'], key
            assert [p.tolist() for p in g['paths']] == r['v2_paths'], key
            # and against the oracle run live (same inputs)
            o = O.decode_hit(al[ri], rl[ri], c['k'], oprm)
            assert [p.tolist() for p in o['paths']] == [p.tolist() for p in g['paths']]


def check_chain_global_fast_synth(ctx, O, seed=31, n_reads=3, L=260, per_pos=7, mode='H'):
    """GC-fast (G3) on synthetic repeat-dense anchor sets (more than 5 anchors per read base): raw S / P / S_arg, best index and the
    selected paths vs the oracle"""
    rng = np.random.default_rng(seed)
    prm = ctx.lib.params(mode); oprm = O.params(mode)
    k = 15
    al, rl = [], []
    for t in range(n_reads):
        rows = []
        step = 1 + t % 2
        copies = [int(x) for x in rng.integers(1000, 2_000_000, (per_pos + 1) * step + 3)]
        for q in range(0, L - k, step):
            for c in copies[:(per_pos + 1) * step + (q % 3)]:
                jitter = int(rng.integers(-2, 3)) if rng.random() < 0.2 else 0
                if rng.random() < 0.85:
                    rows.append((q, c + q + jitter, 1, k))
                else:
                    rows.append((q, c + 5000 - q, -1, k))
        a = np.array(rows, dtype=np.int64)
        a = a[rng.permutation(len(a))]
        assert len(a) / L > 5
        al.append(a); rl.append(L)
    res = ctx.chain_global_batch(prm, k, al, rl, want_raw=True)
    for a, L_, g in zip(al, rl, res):
        flag, fl = O.strand_flip(a.copy(), L_)
        srt = fl[np.argsort(fl[:, 0], kind='stable')]
        eg, eS, eP, eSA = O.chain_global_raw(srt, k, oprm.global_skipcost, oprm.global_maxdiff, 1000, 1, mode)
        assert g['fast_used'] and g['need_reverse'] == flag
        assert g['gmax'] == eg
        assert np.array_equal(g['S'].view(np.uint64), eS.view(np.uint64)), 'GC-fast S'
        assert np.array_equal(g['P'], eP) and np.array_equal(g['S_arg'], eSA), 'GC-fast P / S_arg'
        o = O.decode_hit(a, L_, k, oprm)
        assert g['mapq'] == o['mapq'] and g['score'] == o['score']
        assert [p.tolist() for p in o['paths']] == [p.tolist() for p in g['paths']]


def check_align_random(ctx, O, mode='R', n=8, seed=51, reflen=120000, mean_len=5000, err=0.10, nthreads=8):
    """whole path on seeded reads from an SV donor (+ a short read that stays unmapped / raises) vs the oracle run on the same inputs"""
    from vacmap_amd import synth
    from vacmap_amd.lib import Index, align_batch
    contigs = synth.make_reference([reflen, reflen // 3], seed=seed)
    L = reflen
    ops = [('INV', L // 10, 2500), ('DEL', 3 * L // 10, 1200), ('INS', 4 * L // 10, 600, 5), ('DUP', 6 * L // 10, 2000, 2), ('INVDUP', 8 * L // 10, 1500)]
    d0 = synth.implant_svs(contigs[0], ops)
    k = 19 if mode == 'L' else 15
    cat, off, _ = synth.sample_reads_concat([d0, contigs[1]], n, seed=seed + 1, mean_len=mean_len, err=err, shape='hifi' if mode == 'L' else 'ont',
                                            **({'min_len': 3000} if mode == 'L' else {}))
    seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)] + ['ACGTTGCAAC' * 3]
    names = ['chrA', 'chrB']
    gi = Index.from_seqs(ctx, names, [synth.tostr(c) for c in contigs], k=k, w=10)
    oi = O.Index.from_seqs(names, [synth.tostr(c) for c in contigs], k=k, w=10)
    status, recs, stats = align_batch(ctx, gi, ctx.lib.params(mode), seqs)
    ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=nthreads)
    assert [(int(s) == 0) for s in status] == [(int(s) == 0) for s in ost], (list(status), list(ost))
    assert recs == orecs, 'mode %s: records differ from the oracle' % mode
    return len(orecs)


def check_align_indel_donor(ctx, O, mode='H', n=10, seed=5, reflen=200000, mean_len=7000, err=0.08):
    """whole path on reads from a donor with an insertion or deletion of 30-600 bp every ~2.5 kb (tools/bigverify.py's third leg, small): CIGARs with paired indels, i.e. reads
    the reference runs a second time with nofilter (mammap_clrnano.py:24079-24080: pass 1 of the extend stage); records vs the oracle. Returns the batch statistics."""
    from vacmap_amd import synth
    from vacmap_amd.lib import Index, align_batch
    rng = np.random.default_rng(seed)
    ref = synth.make_reference([reflen], seed=seed + 2)
    ops = []
    for p_ in range(5000, reflen - 5000, 2500):
        n_ = int(rng.integers(30, 600))
        ops.append(('DEL', p_, n_) if rng.random() < 0.5 else ('INS', p_, n_, int(rng.integers(1 << 30))))
    donor = synth.implant_svs(ref[0], ops)
    cat, off, _ = synth.sample_reads_concat([donor], n, seed=seed + 94, mean_len=mean_len, err=err)
    seqs = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(n)]
    gi = Index.from_seqs(ctx, ['chr1'], [ref[0].tobytes()], k=15, w=10)
    oi = O.Index.from_seqs(['chr1'], [ref[0].tobytes()], k=15, w=10)
    status, recs, stats = align_batch(ctx, gi, ctx.lib.params(mode), seqs)
    ost, orecs = O.align_batch(oi, seqs, O.params(mode), nthreads=8)
    assert [(int(x) == 0) for x in status] == [(int(x) == 0) for x in ost]
    assert recs == orecs, 'indel donor, mode %s: records differ from the oracle' % mode
    assert len(orecs) >= n // 2
    return stats


def check_seed_sparse_noise(ctx, O, ref_mb=10, read_len=4000, seed=71, min_hits=900):
    """`.map()` in the regime of an hg38-size index: one true locus plus a thousand-odd ISOLATED stray hits (22-mers of the read planted every
    7 kb of a random reference, some in pairs 3 kb apart = small clusters, some 6 kb apart = neighbours that are not a cluster) — the
    filtered form of k_cluster_big (isolated hits are never sorted; the missing clusters come from a radix select). Anchors equal the
    oracle's for check_num below, at and above the number of multi-hit clusters, and beyond the filtered form's limit (general path)."""
    from vacmap_amd import synth
    from vacmap_amd.lib import Index
    rng = np.random.default_rng(seed)
    ref = synth.make_reference([int(ref_mb * 1e6)], seed=seed)[0]
    p0 = len(ref) // 3
    src = ref[p0:p0 + read_len].copy()
    pos = 50000; j = 0
    while pos + 7000 < len(ref):
        if not (p0 - 20000 < pos < p0 + read_len + 20000):
            a = int(rng.integers(0, read_len - 30)); ln = int(rng.integers(18, 26))
            ref[pos:pos + ln] = src[a:a + ln]
            if j % 9 == 4:                      # a second stray hit 3 kb on: one cluster of two across bins
                b = int(rng.integers(0, read_len - 30)); ref[pos + 3000:pos + 3000 + 22] = src[b:b + 22]
            if j % 11 == 7:                     # a neighbour 6 kb on: adjacent bin, but its own cluster
                b = int(rng.integers(0, read_len - 30)); ref[pos + 6000:pos + 6000 + 22] = src[b:b + 22]
            j += 1
        pos += 7000 + int(rng.integers(0, 2000))
    gi = Index.from_seqs(ctx, ['noise'], [ref], k=15, w=10)
    oi = O.Index.from_seqs(['noise'], [ref], k=15, w=10)
    reads = [synth.mutate(src, 0.03, rng).tobytes(), synth.revcomp(synth.mutate(src[500:3500], 0.02, rng)).tobytes(), src[:700].tobytes()]
    most = 0
    for check_num in (100, 3, 1, 700, 1024, 3000, -1):
        got = ctx.map_batch(gi, reads, check_num=check_num, mid_occ=200)
        for i, rd in enumerate(reads):
            exp = oi.map(rd, check_num, 200)
            assert np.array_equal(got[i], exp), (i, check_num, len(got[i]), len(exp))
            most = max(most, len(exp))
    assert most >= min_hits, most
    gi.close()


def _case_index(ctx, O, meta, arrays, cid):
    from vacmap_amd.lib import Index
    c = meta[cid]
    contigs = [arrays['%s_contig%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))]
    return Index.from_seqs(ctx, c['names'], contigs, k=c['k'], w=c['w']), O.Index.from_seqs(c['names'], contigs, k=c['k'], w=c['w'])


def check_seed_golden(ctx, O, golden, cases=('A', 'B', 'C', 'D')):
    meta, arrays = golden
    for cid in cases:
        c = meta[cid]
        gi, oi = _case_index(ctx, O, meta, arrays, cid)
        assert gi.mid_occ == oi.mid_occ and gi.names == oi.names and gi.offsets == oi.offsets
        gh, gp = gi.minimizers(); oh, op = oi.minimizers()
        assert np.array_equal(gh, oh) and np.array_equal(gp, op), 'index minimizers differ'
        seqs = [arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode() for ri in range(len(c['reads']))]
        sk = ctx.sketch_batch(c['k'], c['w'], seqs)
        for s, (h, p, z) in zip(seqs, sk):
            eh, ep, ez = O.sketch(s, c['k'], c['w'])
            assert np.array_equal(h, eh) and np.array_equal(p, ep) and np.array_equal(z, ez), 'sketch differs'
        mp = ctx.map_batch(gi, seqs)
        for ri, a in enumerate(mp):
            assert np.array_equal(a, arrays['%s_r%d_anchors' % (cid, ri)].reshape(-1, 4)), (cid, ri, 'anchors differ')
        assert gi.seq(0, 5, 25) == oi.seq(0, 5, 25)


def check_seed_many_hits(ctx, O, copies=30, unit=2500, read_len=4000, seed=61, min_hits=9000):
    """`.map()` on reads whose hit count exceeds one and several LDS sort tiles of k_cluster (a diverged tandem array queried with a
    raised occurrence cap): anchors equal the oracle's, cluster ranking and check_num cut included"""
    from vacmap_amd import synth
    from vacmap_amd.lib import Index
    rng = np.random.default_rng(seed)
    u = synth.make_reference([unit], seed=seed)[0]
    parts = [synth.make_reference([20000], seed=seed + 1)[0]]
    for c in range(copies):
        parts.append(synth.mutate(u, 0.02, rng))
        if c % 7 == 3:
            parts.append(synth.make_reference([6000], seed=seed + 10 + c)[0])      # a gap > 5000 splits the array into several clusters
    parts.append(synth.make_reference([20000], seed=seed + 2)[0])
    ref = np.concatenate(parts)
    gi = Index.from_seqs(ctx, ['rep'], [ref], k=15, w=10)
    oi = O.Index.from_seqs(['rep'], [ref], k=15, w=10)
    reads = []
    for i in range(3):
        body = np.concatenate([u, u])[i * 300:i * 300 + read_len]
        reads.append(synth.mutate(body, 0.05, rng).tobytes())
    reads.append(synth.mutate(ref[1000:1000 + read_len], 0.05, rng).tobytes())       # an ordinary read next to them (small class)
    for check_num, occ in ((100, 4 * copies), (3, 4 * copies), (-1, 4 * copies), (100, -1)):
        got = ctx.map_batch(gi, reads, check_num=check_num, mid_occ=occ)
        tot = 0
        for i, rd in enumerate(reads):
            exp = oi.map(rd, check_num, occ)
            assert np.array_equal(got[i], exp), (i, check_num, occ, len(got[i]), len(exp))
            tot = max(tot, len(exp))
        if check_num == -1:
            assert tot >= min_hits, tot
    gi.close()


def check_align_golden(ctx, O, golden, cases=('A', 'B', 'C', 'D'), reads=None, min_ext_retries=None):
    """the whole batched path (vm_align_batch) vs the reference's records (V6) and vs the oracle run live"""
    from vacmap_amd.lib import align_batch
    meta, arrays = golden
    for cid in cases:
        c = meta[cid]
        gi, oi = _case_index(ctx, O, meta, arrays, cid)
        prm = ctx.lib.params(c['mode']); oprm = O.params(c['mode'])
        idx = list(range(len(c['reads']))) if reads is None else [i for i in reads if i < len(c['reads'])]
        seqs = [arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode() for ri in idx]
        status, recs, stats = align_batch(ctx, gi, prm, seqs)
        for x, ri in enumerate(idx):
            r = c['reads'][ri]
            mine = [[c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]] for t in recs if t[0] == x]
            ost, orecs = O.align_read(oi, seqs[x], oprm)
            omine = [[c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]] for t in orecs]
            assert (status[x] == 0) == (ost == 0), (cid, ri, int(status[x]), ost)
            assert mine == omine, (cid, ri, 'records differ from the oracle')
            assert (status[x] == 0) == (r['v6_status'] == 0) and mine == r['v6_records'], (cid, ri, 'records differ from the reference golden')
        assert stats['n_reads'] == len(seqs)
        if min_ext_retries is not None:
            assert stats['n_ext_retries'] >= min_ext_retries, (cid, stats['n_ext_retries'])
    return stats


_COMP = bytes.maketrans(b'ACGTN', b'TGCAN')


def check_local_golden(ctx, O, golden, cases=('A', 'B', 'C', 'D')):
    """local re-seeding + local chain: HIP vs the reference goldens (V3) and vs the oracle run on the same inputs"""
    from vacmap_amd.lib import local_chain_batch
    meta, arrays = golden
    for cid in cases:
        c = meta[cid]
        gi, oi = _case_index(ctx, O, meta, arrays, cid)
        prm = ctx.lib.params(c['mode']); oprm = O.params(c['mode'])
        seqs, paths, keys = [], [], []
        for ri, r in enumerate(c['reads']):
            if not r['v2_paths']:
                continue
            s = arrays['%s_r%d_seq' % (cid, ri)].tobytes()
            seqs.append(s if r['v2_score'] > 0 else s.translate(_COMP)[::-1])
            paths.append([np.array(p, dtype=np.int64) for p in r['v2_paths']]); keys.append(ri)
        res = local_chain_batch(ctx, gi, prm, seqs, paths)
        for x, ri in enumerate(keys):
            r = c['reads'][ri]; key = '%s_r%d' % (cid, ri); g = res[x]
            o = O.local_chain(oi, seqs[x], paths[x], oprm)
            oraw = o['raw'][np.argsort(o['raw'][:, 0] + (0 if c['mode'] == 'R' else o['raw'][:, 3]), kind='stable')] if len(o['raw']) else o['raw']
            assert g['status'] == 0, (key, g['status'])
            assert np.array_equal(g['raw'], oraw), key + ' raw local anchors differ from the oracle'
            assert g['variant'] == o['variant'] and g['score'] == o['score'] and np.array_equal(g['chain'], o['chain']), key + ' chain vs oracle'
            if 'v3_score' in r:
                if key + '_v3_raw' in arrays:
                    assert np.array_equal(g['raw'], arrays[key + '_v3_raw'].reshape(-1, 4)), key + ' raw vs golden'
                else:                # dense cases: count + checksum of the raw local anchors
                    import zlib
                    assert len(g['raw']) == r['v3_raw_n'] and zlib.crc32(np.ascontiguousarray(g['raw'].astype(np.int64)).tobytes()) == r['v3_raw_crc'], key + ' raw vs golden'
                assert g['variant'] == r['v3_variant'] and g['score'] == r['v3_score'], key
                assert np.array_equal(g['chain'], arrays[key + '_v3_path'].reshape(-1, 4)), key + ' chain vs golden'


def _end_markers(rows):
    """segment lists with the first and last anchor of every segment written as zero-length end markers — (q, r or r + l on the '-'
    strand, s, 0) and (q + l, r + l or r on the '-' strand, s, 0): the form split_alignment_test (:21530-21547) gives the last anchor and
    the kernel of stage 5 gives both ends in place. Same end points, comparable lists."""
    rows = np.array(rows, dtype=np.int64).reshape(-1, 5).copy()
    for sg in np.unique(rows[:, 0]):
        ix = np.nonzero(rows[:, 0] == sg)[0]
        a, b = ix[0], ix[-1]
        _, q, r, st, ln = rows[b]
        rows[b] = (sg, q + ln, r + ln if st == 1 else r, st, 0)
        if a != b:
            _, q, r, st, ln = rows[a]
            rows[a] = (sg, q, r if st == 1 else r + ln, st, 0)
    return rows


def check_stage_trace_golden(ctx, O, golden, cases=('A', 'B', 'I'), reads=None):
    """E1 / E3 / E4 stage by stage (golden V4): the product's segment lists after rebuild_chain_break, after the extension rounds + the
    drop_misplaced loop, and after merge_conjacent + fix_simple_inv (vm_align_trace) equal what the REFERENCE held at those points"""
    from vacmap_amd.lib import align_trace
    meta, arrays = golden
    checked = {0: 0, 3: 0, 5: 0}
    for cid in cases:
        c = meta[cid]
        gi, _ = _case_index(ctx, O, meta, arrays, cid)
        prm = ctx.lib.params(c['mode'])
        sel = [ri for ri in range(len(c['reads'])) if reads is None or ri in reads]
        seqs = [arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode() for ri in sel]
        got = {st: align_trace(ctx, gi, prm, seqs, st) for st in (0, 3, 5)}
        for x, ri in enumerate(sel):
            v4 = c['reads'][ri].get('v4', [])
            if not v4:
                assert all(len(got[st][x]) == 0 for st in (0, 3, 5)), (cid, ri)
                continue
            run0 = []
            for e in v4:                                     # the first extend_func run = everything before a second rebuild_chain_break
                if e['fn'] == 'rebuild_chain_break' and run0:
                    break
                run0.append(e)
            exp0 = arrays[run0[0]['out']].reshape(-1, 5)
            assert np.array_equal(got[0][x], exp0), (cid, ri, 'rebuild_chain_break')
            checked[0] += 1
            merge = [e for e in run0 if e['fn'] == 'merge_conjacent_alignment']
            fix = [e for e in run0 if e['fn'] == 'fix_simple_inv']
            drops = [e for e in run0 if e['fn'] == 'drop_misplaced_alignment_test' and e['removed']]
            if drops or merge:
                exp3 = arrays[(drops[-1]['out'] if drops else merge[0]['in'])].reshape(-1, 5)
                assert np.array_equal(got[3][x], exp3), (cid, ri, 'extension + drop_misplaced')
                checked[3] += 1
            if fix:
                assert np.array_equal(_end_markers(got[5][x]), _end_markers(arrays[fix[0]['out']])), (cid, ri, 'merge_conjacent + fix_simple_inv')
                checked[5] += 1
    return checked


def check_reference_call_shapes(ctx, O, golden, small=False):
    """The three exports that carry the reference's own call shape, and the `vacmap_index`-shaped shim on top of them (vacmap_amd/aligner.py),
    against the oracle and against their batched siblings:
      index_object.map(seq, check_num=, mid_occ=)                         mammap_clrnano.py:23985   -> vm_map
      mp.k_cigar(t, q, 2, -4, 4, 2, 24, 1, bw=-1, zdropvalue=-1, eqx=)     :21554, :21598           -> vm_k_cigar (global, traceback)
      mp.k_cigar(t, q, 2, -4, 4, 4, 4, 4, bw=100, zdropvalue=50)           :2381, :2410, :2477, :2505 -> vm_k_cigar (z-drop extension: q_e, t_e)
      edlib.align(query=, target=, task='distance')['editDistance']       :19251                   -> vm_edit_distance"""
    from vacmap_amd import aligner as AL
    meta, arrays = golden
    gi, oi = _case_index(ctx, O, meta, arrays, 'A')
    AL.use_context(ctx)
    al = AL.Aligner(index=gi, ctx=ctx)
    # the Aligner surface the reference reads (vacmap:344-363, :24024)
    assert bool(al) and al.k == 15 and al.w == 10
    assert [(n.decode(), ln, off) for n, ln, off in al.seq_offset] == list(zip(gi.names, gi.lens, gi.offsets))
    c0 = arrays['A_contig0'].tobytes().decode()
    assert al.seq(gi.names[0]) == c0 and al.seq(gi.names[0], 100, 260) == c0[100:260]
    rng = np.random.default_rng(91)
    read = arrays['A_r0_seq'].tobytes().decode()
    pieces = [read[:3000], read[7000:7000 + (1500 if small else 6000)], mutate(rng, c0[2000:4000], 0.1), 'ACGT' * 5, 'A']
    for sq in pieces:
        for cn, mo in ((100, -1), (-1, -1), (3, 50)):
            rows = al.map(sq, check_num=cn, mid_occ=mo)
            exp = [tuple(int(v) for v in r) for r in oi.map(sq, check_num=cn, mid_occ=mo)]
            assert rows == exp, (len(sq), cn, mo, len(rows), len(exp))
            assert rows == [tuple(int(v) for v in r) for r in ctx.map_batch(gi, [sq], check_num=cn, mid_occ=mo)[0]]
    # k_cigar, global parameterisation
    ts, qs = [], []
    for i in range(6 if small else 24):
        a = rand_seq(rng, int(rng.integers(1, 120 if small else 420)))
        b = mutate(rng, a, float(rng.choice([0.0, 0.1, 0.25]))) or 'C'
        ts.append(a); qs.append(b)
    ts += [t for t, _ in gapfill_tie_cases()[:4]]; qs += [q for _, q in gapfill_tie_cases()[:4]]
    for eqx in (False, True):
        batch, bsc = ctx.k_cigar_batch(ts, qs, eqx=eqx)
        for i, (t, q) in enumerate(zip(ts, qs)):
            cg, zc, q_e, t_e, nd, ni = AL.k_cigar(t, q, match=2, mismatch=-4, gap_open_1=4, gap_extend_1=2, gap_open_2=24, gap_extend_2=1, bw=-1, zdropvalue=-1, eqx=eqx)
            assert cg == O.k_cigar_global(t, q, eqx=eqx)[0] == batch[i], (i, eqx)
            assert (q_e, t_e) == (len(q), len(t))
            ql = sum(int(n) for n, op in re.findall(r'(\d+)([MIDX=])', cg) if op in 'MIX=')
            assert ql == len(q)                                  # what the reference asserts on the merged CIGAR (:20779-20781)
    # k_cigar, z-drop extension parameterisation: the reference uses q_e / t_e only
    es, et, eq = ctx.k_extend_batch(ts, qs, 2, -4, 4, 4, 100, 50)
    for i, (t, q) in enumerate(zip(ts, qs)):
        cg, zc, q_e, t_e, nd, ni = AL.k_cigar(t, q, 2, -4, 4, 4, 4, 4, bw=100, zdropvalue=50)
        osc, ote, oqe = O.k_extend(t, q, 2, -4, 4, 4, 100, 50)
        assert (t_e, q_e) == (ote, oqe) == (int(et[i]), int(eq[i])), i
    # edlib.align(...)['editDistance']
    eb = ctx.edit_distance_batch(qs, ts)
    for i, (t, q) in enumerate(zip(ts, qs)):
        d = AL.edlib_align(query=q, target=t, task='distance')['editDistance']
        assert d == O.edit_distance(q, t) == int(eb[i]), i


# ------------------------------------------------------------------------------------------------ -mode asm (mammap_asm.py)
def asm_golden():
    import json, os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    return json.load(open(os.path.join(g, 'asm.json'))), np.load(os.path.join(g, 'asm.npz'))


def _asm_index(ctx, O, meta, arr, cid):
    from vacmap_amd.lib import Index
    c = meta[cid]
    contigs = [arr['%s_ref%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))]
    return Index.from_seqs(ctx, c['names'], contigs, k=c['k'], w=c['w']), O.Index.from_seqs(c['names'], contigs, k=c['k'], w=c['w'])


def check_asm_golden(ctx, O, cases=('AS1',), contigs=None, want_unsupported=(), vs_golden=True):
    """vm_align_batch in VM_MODE_ASM: every assembly contig below 500 kb takes the fork's per-read function (mammap_asm.py:19681). Records must
    equal the reference's (golden V6a) and the oracle's run live; contigs named in want_unsupported must come back as VM_READ_UNSUPPORTED (-22)
    with no records (500 kb and more: the linked path is not on the device)"""
    import zlib
    from vacmap_amd.lib import align_batch
    meta, arr = asm_golden()
    n_ok = 0
    for cid in cases:
        c = meta[cid]
        gi, oi = _asm_index(ctx, O, meta, arr, cid)
        prm = ctx.lib.params('asm'); oprm = O.params('asm')
        assert prm.eqx == 1 and prm.check_num == -1 and prm.maxdivergence == 1.0
        idx = list(range(len(c['contigs']))) if contigs is None else [i for i in contigs if i < len(c['contigs'])]
        seqs = [arr['%s_c%d_seq' % (cid, ci)].tobytes().decode() for ci in idx]
        status, recs, stats = align_batch(ctx, gi, prm, seqs)
        for x, ci in enumerate(idx):
            g = c['contigs'][ci]
            mine = [t for t in recs if t[0] == x]
            if g['name'] in want_unsupported:
                assert status[x] == -22 and not mine, (cid, g['name'], int(status[x]))
                continue
            ost, orecs = O.align_asm(oi, seqs[x], oprm)
            assert (status[x] == 0) == (ost == 0), (cid, g['name'], int(status[x]), ost)
            assert [t[1:] for t in mine] == [t[1:] for t in orecs], (cid, g['name'], 'records differ from the oracle')
            got = [[c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], len(t[8]), zlib.crc32(t[8].encode())] + ([t[8]] if len(t[8]) <= 4096 else []) for t in mine]
            if vs_golden:        # (AS3's goldens were made with shrunk size constants: there only the live oracle, at the reference's sizes, applies)
                assert (status[x] == 0) == (g['status'] == 0) and got == g['records'], (cid, g['name'], 'records differ from the reference golden')
            n_ok += 1
    return n_ok


def check_asm_decode_hit(ctx, O, cases=('AS1', 'AS4')):
    """stage entry vm_chain_global_batch in VM_MODE_ASM (GC-exact / GC-fast of the fork + its hit2work_1 / decode_hit) vs golden V2a"""
    meta, arr = asm_golden()
    n = 0
    for cid in cases:
        c = meta[cid]
        gi, oi = _asm_index(ctx, O, meta, arr, cid)
        prm = ctx.lib.params('asm')
        seqs = [arr['%s_c%d_seq' % (cid, ci)].tobytes().decode() for ci in range(len(c['contigs']))]
        anchors = ctx.map_batch(gi, seqs, check_num=-1)
        res = ctx.chain_global_batch(prm, c['k'], anchors, [len(s) for s in seqs])
        for ci, g in enumerate(c['contigs']):
            r = res[ci]
            assert r['mapq'] == g['v2_mapq'] and r['score'] == g['v2_score'], (cid, g['name'], r['mapq'], r['score'], g['v2_mapq'], g['v2_score'])
            if g['v2_score'] != 0:
                assert len(r['paths']) == 1 and np.array_equal(r['paths'][0], arr['%s_c%d_v2_path' % (cid, ci)]), (cid, g['name'])
                assert bool(r['fast_used']) == (cid == 'AS4'), (cid, g['name'])
                n += 1
    return n


def check_asm_linked_golden(ctx, O, cases=('AS2', 'AS3'), max_calls=None):
    """vm_chain_linked (k_chain_linked.hip) on the carried states the REFERENCE built (golden VL): S and P of every anchor bit-equal, the hot
    part of the score index = the tail of the reference's S_arg, g_max_index equal; and the state the device carries into the next batch
    (k_link_carry, mammap_asm.py:23250-23272) = the pre_S / pre_P / rows the reference fed to its NEXT linked call"""
    meta, arr = asm_golden()
    n_full = {0: 0, 2: 0}; n_carry = 0; n_cold = 0
    for cid in cases:
        for c in meta[cid]['contigs']:
            calls = c['linked_calls']
            for ei, e in enumerate(calls):
                if 'key' not in e:
                    continue
                if max_calls is not None and n_full[e['which']] >= max_calls:
                    continue
                kk = e['key']
                r = ctx.chain_linked(arr[kk + '_rows'], e['which'], int(e['kw']['kmersize']), e['kw']['skipcost'], int(e['kw']['maxdiff']), int(e['kw']['maxgap']),
                                     e['g_max_scores'], e['g_max_index'], arr[kk + '_preS'], arr[kk + '_preP'], e['prereadloc'])
                assert r['gmax'] == e['g'], (kk, r['gmax'], e['g'])
                assert np.array_equal(r['S'].view(np.uint64), arr[kk + '_S'].view(np.uint64)), kk
                assert np.array_equal(r['P'], arr[kk + '_P']), kk
                sa = arr[kk + '_Sarg']
                assert r['n_hot'] + r['n_cold'] == len(sa) and np.array_equal(r['S_arg_hot'], sa[len(sa) - r['n_hot']:]), (kk, 'hot index differs from the tail of S_arg')
                n_full[e['which']] += 1; n_cold += r['n_cold']
                # the next linked call of the same round was fed what the reference carried out of this one
                nxt = calls[ei + 1] if ei + 1 < len(calls) and calls[ei + 1]['which'] == e['which'] else None
                if nxt is not None and nxt['n_pre'] > 0 and 'key' in nxt:
                    k2 = nxt['key']
                    assert r['carry_status'] == 0 and r['saved'] == 1 and r['n_carry'] == nxt['n_pre'], (kk, r['carry_status'], r['saved'], r['n_carry'], nxt['n_pre'])
                    assert np.array_equal(r['carry_S'].view(np.uint64), arr[k2 + '_preS'].view(np.uint64)) and np.array_equal(r['carry_P'], arr[k2 + '_preP'])
                    assert np.array_equal(r['carry_rows'], arr[k2 + '_rows'][:nxt['n_pre']])
                    assert r['carry_g_max_scores'] == nxt['g_max_scores'] and r['carry_prereadloc'] == nxt['prereadloc'] and r['n_carry'] - 1 == nxt['g_max_index']
                    n_carry += 1
    return n_full, n_carry, n_cold


def check_asm_linked_noise(ctx, O, seed=5, noise_per_anchor=8, which=0, cid='AS3', contig=1):
    """k_chain_linked on a linked batch of the goldens with `noise_per_anchor` isolated random anchors mixed in per true anchor (what
    check_num = -1 gives on a human-size reference: every noise anchor hangs itself onto the best chain at the skip penalty). S / P of every
    anchor, g_max_index, the stored tail of S_arg and the carried state must equal the oracle's (which keeps the whole index like the reference)"""
    meta, arr = asm_golden()
    c = meta[cid]['contigs'][contig]
    e = [x for x in c['linked_calls'] if 'key' in x and x['which'] == which][0]
    kk = e['key']
    rows = arr[kk + '_rows']; n_pre = e['n_pre']
    rng = np.random.default_rng(seed)
    new = rows[n_pre:]
    m = noise_per_anchor * len(new)
    q = rng.integers(new[:, 0].min(), new[:, 0].max() + 1, m)
    noise = np.stack([q, rng.integers(0, 3_000_000_000, m), rng.choice([-1, 1], m), np.full(m, 15 if which == 0 else 9)], axis=1).astype(np.int64)
    allnew = np.concatenate([new, noise])
    allnew = allnew[np.argsort(allnew[:, 0], kind='stable')]
    linked = np.ascontiguousarray(np.concatenate([rows[:n_pre], allnew]))
    kw = e['kw']
    args = (int(kw['kmersize']), kw['skipcost'], int(kw['maxdiff']), int(kw['maxgap']), e['g_max_scores'], e['g_max_index'], arr[kk + '_preS'], arr[kk + '_preP'], e['prereadloc'])
    g, S, P, SA = O.chain_linked_raw(linked, which, *args)
    r = ctx.chain_linked(linked, which, *args)
    assert r['gmax'] == g and np.array_equal(r['S'].view(np.uint64), S.view(np.uint64)) and np.array_equal(r['P'], P)
    assert r['n_hot'] + r['n_cold'] == len(SA) and np.array_equal(r['S_arg_hot'], SA[len(SA) - r['n_hot']:])
    # mammap_asm.py:23250-23272 on the oracle's arrays
    if P[g] < 0:
        assert r['carry_status'] == 0 and r['saved'] == 0
    else:
        low = S[SA[-1]] - kw['skipcost'] - 36 - 20
        sl = len(S) - 1
        while low < S[SA[sl]]:
            sl -= 1
            if sl == 0:
                break
        assert r['carry_status'] == 0 and r['saved'] == 1 and r['n_carry'] == len(S) - sl
        assert np.array_equal(r['carry_S'].view(np.uint64), (S[SA[sl:]] - S[SA[sl]] + 1000).view(np.uint64)) and np.array_equal(r['carry_P'], -P[SA[sl:]])
        assert np.array_equal(r['carry_rows'], linked[SA[sl:]]) and r['carry_prereadloc'] == linked[SA[sl:], 0].max()
    return r['n_cold'], r['n_hot']


def check_asm_long_golden(ctx, O, cid='AS3', contigs=None, vs_golden=True):
    """vm_align_asm on contigs that take the LINKED path (assembly_get_readmap_DP_test :23208-23422) with the sizes the goldens were made with:
    records = the reference's (golden V6a) = the oracle's run live"""
    import zlib
    from vacmap_amd.lib import align_asm
    meta, arr = asm_golden()
    c = meta[cid]
    gi, oi = _asm_index(ctx, O, meta, arr, cid)
    prm = ctx.lib.params('asm'); oprm = O.params('asm')
    sizes = c['sizes']
    n_ok = 0
    for ci, g in enumerate(c['contigs']):
        if contigs is not None and ci not in contigs:
            continue
        seq = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
        st, recs = align_asm(ctx, gi, prm, seq, *sizes)
        ost, orecs = O.align_asm(oi, seq, oprm, *sizes)
        assert (st == 0) == (ost == 0), (cid, g['name'], st, ost)
        assert [t[1:] for t in recs] == [t[1:] for t in orecs], (cid, g['name'], 'records differ from the oracle')
        if vs_golden:
            got = [[c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], len(t[8]), zlib.crc32(t[8].encode())] + ([t[8]] if len(t[8]) <= 4096 else []) for t in recs]
            assert (st == 0) == (g['status'] == 0) and got == g['records'], (cid, g['name'], 'records differ from the reference golden')
        n_ok += 1
    return n_ok


def check_asm_linked_fast_golden(ctx, O):
    """the fork's GC-fast and its LINKED form on the device (k_chain_linked_fast, mammap_asm.py:20738 / :21871) against the direct calls of the
    reference kept in golden AS4: S, P, the (int(S), diagonal)-ordered index, g_max_index; the carried state against the one the generator built"""
    meta, arr = asm_golden()
    d = meta['AS4']['direct']
    r1 = ctx.chain_linked(arr['AS4_direct_first'], 1, 15, 30., 50, 1000)
    assert r1['gmax'] == d['g1'] and np.array_equal(r1['S'].view(np.uint64), arr['AS4_direct_S1'].view(np.uint64))
    assert np.array_equal(r1['P'], arr['AS4_direct_P1']) and np.array_equal(r1['S_arg_hot'], arr['AS4_direct_SA1'])
    pre_S = arr['AS4_direct_preS']
    assert r1['carry_status'] == 0 and r1['saved'] == 1 and r1['n_carry'] == d['n_pre']
    assert np.array_equal(r1['carry_S'].view(np.uint64), pre_S.view(np.uint64)) and np.array_equal(r1['carry_P'], arr['AS4_direct_preP'])
    assert np.array_equal(r1['carry_rows'], arr['AS4_direct_linked'][:d['n_pre']]) and r1['carry_prereadloc'] == d['prereadloc']
    r2 = ctx.chain_linked(arr['AS4_direct_linked'], 1, 15, 30., 50, 1000, pre_S[-1], len(pre_S) - 1, pre_S, arr['AS4_direct_preP'], d['prereadloc'])
    assert r2['gmax'] == d['g2'] and np.array_equal(r2['S'].view(np.uint64), arr['AS4_direct_S2'].view(np.uint64))
    assert np.array_equal(r2['P'], arr['AS4_direct_P2']) and np.array_equal(r2['S_arg_hot'], arr['AS4_direct_SA2'])
    return d['n_first'], d['n_linked']


def check_asm_long_forced_fast(ctx, O, monkeypatch, cid='AS3', contig=2, factor=0.5):
    """the bail-out route of the long-contig loop (mammap_asm.py:23246-23247): with max_factor lowered in the oracle and in the library (test hooks)
    every first-round batch leaves GC-exact for the fork's linked GC-fast; records must equal the oracle's"""
    from vacmap_amd.lib import align_asm
    meta, arr = asm_golden()
    c = meta[cid]
    gi, oi = _asm_index(ctx, O, meta, arr, cid)
    prm = ctx.lib.params('asm'); oprm = O.params('asm')
    seq = arr['%s_c%d_seq' % (cid, contig)].tobytes().decode()
    monkeypatch.setenv('VMX_TEST_ASM_MAX_FACTOR', str(factor))
    O.lib().vmo_test_asm_max_factor(factor)
    try:
        O.fast_counters(reset=True)
        ost, orecs = O.align_asm(oi, seq, oprm, *c['sizes'])
        nfast = O.fast_counters()[0]
        st, recs = align_asm(ctx, gi, prm, seq, *c['sizes'])
    finally:
        O.lib().vmo_test_asm_max_factor(1000.0)
    assert nfast >= 2, nfast
    assert (st == 0) == (ost == 0) and [t[1:] for t in recs] == [t[1:] for t in orecs], 'records differ from the oracle'
    return nfast, len(recs)


def check_asm_ragged(ctx, O):
    """edge contigs of -mode asm in one batch: tiny, all-N, N runs inside, lower case, unmappable, a contig spanning two reference contigs — statuses and
    records = the oracle's"""
    from vacmap_amd.lib import align_batch
    from vacmap_amd import synth
    meta, arr = asm_golden()
    gi, oi = _asm_index(ctx, O, meta, arr, 'AS1')
    ref0 = arr['AS1_ref0'].tobytes().decode(); ref1 = arr['AS1_ref1'].tobytes().decode()
    rng = np.random.default_rng(77)
    base = synth.tostr(synth.mutate(np.frombuffer(ref0[120000:150000].encode(), np.uint8), 0.004, rng))
    withn = base[:8000] + 'N' * 700 + base[8700:20000] + 'N' + base[20001:]
    contigs = ['ACGTACGTACGT', 'N' * 4000, withn, base.lower(), synth.tostr(synth.make_reference([6000], seed=78)[0]),
               ref0[280000:299000] + ref1[1000:21000], base[:300]]
    prm = ctx.lib.params('asm'); oprm = O.params('asm')
    status, recs, _ = align_batch(ctx, gi, prm, contigs)
    n = 0
    for x, s in enumerate(contigs):
        ost, orecs = O.align_asm(oi, s, oprm)
        mine = [t[1:] for t in recs if t[0] == x]
        assert (status[x] == 0) == (ost == 0), (x, int(status[x]), ost)
        assert mine == [t[1:] for t in orecs], (x, 'records differ from the oracle')
        n += len(mine)
    return n


def check_local_many_chains(ctx, O, copies=80, unit=900, seed=91, modes=('S', 'H'), min_copies=64):
    """mode S re-seeds EVERY chain of hit2work_1 (mammap_sensitive.py: no budget): a read with far more than 64 chains (the cap the local stage
    had, DESIGN D5) — one per dispersed copy of its sequence — through vm_local_chain_batch vs the oracle: raw local anchors, variant, score, chain"""
    from vacmap_amd.lib import Index, local_chain_batch
    rng = np.random.default_rng(seed)
    base = rand_seq(rng, unit)
    parts, starts, pos = [], [], 0
    for c in range(copies):
        spacer = rand_seq(rng, 2600)
        cp = mutate(rng, base, 0.02) if c else base
        parts += [spacer, cp]; starts.append(pos + len(spacer)); pos += len(spacer) + len(cp)
    ref = ''.join(parts) + rand_seq(rng, 3000)
    gi = Index.from_seqs(ctx, ['c'], [np.frombuffer(ref.encode(), np.uint8)], k=15, w=10)
    oi = O.Index.from_seqs(['c'], [ref], k=15, w=10)
    read = mutate(rng, base, 0.03)
    # one chain per copy, the true one first (descending read order, rows (q, r, strand, len)): anchors every ~150 bases along the copy's diagonal
    qs = list(range(unit - 120, 40, -150))
    paths = [np.array([[q, starts[c] + q, 1, 15] for q in qs], dtype=np.int64) for c in range(copies)]
    assert len(paths) > min_copies
    for mode in modes:
        prm = ctx.lib.params(mode); oprm = O.params(mode)
        g = local_chain_batch(ctx, gi, prm, [read.encode()], [paths])[0]
        o = O.local_chain(oi, read.encode(), paths, oprm)
        oraw = o['raw'][np.argsort(o['raw'][:, 0] + o['raw'][:, 3], kind='stable')] if len(o['raw']) else o['raw']
        assert g['status'] == 0, (mode, g['status'])
        assert np.array_equal(g['raw'], oraw), mode + ': raw local anchors differ from the oracle'
        assert g['variant'] == o['variant'] and g['score'] == o['score'] and np.array_equal(g['chain'], o['chain']), mode
        if mode == 'S':
            rr = g['raw'][:, 1]
            assert sum(bool(np.any((rr >= st) & (rr < st + unit))) for st in starts) > min_copies      # anchors inside more than 64 copies: every chain was re-seeded
    gi.close()
