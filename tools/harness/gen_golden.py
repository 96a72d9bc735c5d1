#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python (build container only).

Vectors:
  V1 get_reversed_chain_numpy_rough      anchors -> (flag, anchors)
  V2 hit2work_1 / decode_hit             anchors -> paths, scores, mapq, secondaries; raw S, P, S_arg of GC-exact
  V3 ..._guide_list                      guide paths + read -> raw local anchors (argument of the LC DP), (score, path)
  V4 segment surgery                     rebuild_chain_break :23437, drop_misplaced_alignment_test :726, merge_conjacent_alignment :16736 (+ getdupiloc_numba
                                         :16680 inside it), fix_simple_inv :24226 — segment lists in / out of every call the read makes
  V5 DP problem list                     every k_cigar / edlib call the reference makes for the read (kind, |t|, |q|, crc)
  V6 get_readmap_DP_test                 read -> onemapinfolist 9-tuples (the record type of the path)
  V7 tests/test_nm_from_cigar.py         the reference's 9 known-answer triples for nm_from_cigar
  V8 cost tables                         tools/gen_tables.py (tests/golden/tables.json)
Inputs (reference contigs, reads) are stored next to the outputs, so the fixtures are self-contained.
The reference sources are imported in place and never copied.
"""
import json, os, sys, zlib
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, _HERE)
import refrun
from refrun import O
from vacmap_amd import synth

GOLD = os.path.join(_ROOT, 'tests', 'golden')


VACSIM_TEXT_H = """Specified{INV:300:600,DUP:300:600:1:2,TRA:400:800:1;number=2}
Specified{DEL:100:200,INS:100:400,INV:300:500,DUP:300:500:0:3,TRA:400:800:0;number=1}
Specified{INV:400:800,NML:100:200,INV:400:800;number=1}
Random{eventset=["DEL:100:200,INV:300:600","INS:100:300,NML:100:200","DUP:300:600","TRA:400:800"];eventcount=[2,6];number=2}
"""


def rows(x):
    return [[int(v) for v in r] for r in x]


def seg_rows(al):
    """list of segments (lists of 4-tuples) -> int64 rows (segment index, q, r, s, l)"""
    out = [[si, int(a[0]), int(a[1]), int(a[2]), int(a[3])] for si, seg in enumerate(al) for a in seg]
    return np.array(out, dtype=np.int64).reshape(-1, 5)


def run_case(cid, mode, names, contigs, reads, k, arrays, meta, v4=False, light=False):
    """light: bulk cases keep the per-read outputs (V1, V2 paths / score / MAPQ, V3 score + chain + checksum of the raw local anchors, V5, V6) but not
    the raw S / P / S_arg arrays and raw local-anchor rows, so that the fixture stays small"""
    ix = O.Index.from_seqs(names, contigs, k=k, w=10)
    al = refrun.Aligner(oracle_index=ix)
    ctx = refrun.RefContext(mode, al)
    m = ctx.m
    case = {'mode': mode, 'k': k, 'w': 10, 'names': names, 'reads': []}
    for ci, c in enumerate(contigs):
        arrays['%s_contig%d' % (cid, ci)] = np.frombuffer(c.encode(), dtype=np.uint8)
    for ri, (rname, seq) in enumerate(reads):
        key = '%s_r%d' % (cid, ri)
        arrays[key + '_seq'] = np.frombuffer(seq.encode(), dtype=np.uint8)
        L = len(seq)
        rec = {'name': rname, 'len': L}
        anchors = np.array(al.map(seq, check_num=100, mid_occ=-1), dtype=np.int64).reshape(-1, 4)
        arrays[key + '_anchors'] = anchors
        # V1
        flag, flipped = m.get_reversed_chain_numpy_rough(anchors.copy(), L)
        rec['v1_flag'] = bool(flag)
        arrays[key + '_v1'] = np.ascontiguousarray(flipped)
        # V2 raw GC-exact on the q-sorted flipped anchors + hit2work_1
        if len(flipped) > 2 and not light:
            srt = np.ascontiguousarray(flipped)[np.argsort(np.ascontiguousarray(flipped)[:, 0])]
            g, S, P, SA, fac = m.get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all(
                srt, kmersize=k, skipcost=ctx.option['golbal_skipcost'], maxdiff=ctx.option['golbal_maxdiff'], maxgap=1000)
            rec['v2_gmax'] = int(g)
            arrays[key + '_v2_S'] = np.asarray(S, dtype=np.float64)
            arrays[key + '_v2_P'] = np.asarray(P, dtype=np.int64)
            arrays[key + '_v2_Sarg'] = np.asarray(SA, dtype=np.int64)
            if g == -1 or len(srt) / L > 5:      # the reference switches to GC-fast (:23570, :23577): keep its raw arrays too
                gf, Sf, Pf, SAf = m.get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all(
                    srt, kmersize=k, skipcost=ctx.option['golbal_skipcost'], maxdiff=ctx.option['golbal_maxdiff'], maxgap=1000)
                rec['v2f_gmax'] = int(gf)
                arrays[key + '_v2f_S'] = np.asarray(Sf, dtype=np.float64)
                arrays[key + '_v2f_P'] = np.asarray(Pf, dtype=np.int64)
                arrays[key + '_v2f_Sarg'] = np.asarray(SAf, dtype=np.int64)
        try:
            mapq, scores, path, factor, rpl = m.decode_hit(al, ctx.index2contig, seq, L, ctx.contig2start, k, ctx.contig2seq,
                                                         skipcost=(ctx.option['golbal_skipcost'],) * 2,
                                                         maxdiff=(ctx.option['golbal_maxdiff'],) * 2, maxgap=200, check_num=100,
                                                         c_bias=5000, bin_size=100, overlapprecentage=0.5, hastra=False, H=False, mid_occ=-1)
        except UnboundLocalError:      # mode R: `factor` is unbound when <= 2 anchors survive (mammap_noprefercloser.py:24417)
            mapq, scores, path, factor, rpl = 0, 0., [], 0, []
            rec['v2_raised'] = True
        rec['v2_mapq'] = int(mapq); rec['v2_score'] = float(scores)
        rec['v2_paths'] = [rows(p) for p in rpl]
        # V3: capture the LC DP's input (raw local anchors, sorted by q+l) and its output
        cap = {}
        names_lc = ['get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list',
                    'get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_mismatch',
                    'get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_scar']       # variant 0 / 1 / 2 (mode R)
        if mode == 'R':
            names_lc = [None, None, names_lc[2]]
        orig = {n: getattr(m, n) for n in names_lc if n}

        def mk(n):
            def f(one_mapinfo, **kw):
                cap['variant'] = names_lc.index(n); cap['raw'] = np.array(one_mapinfo); cap['kw'] = {a: float(b) for a, b in kw.items()}
                return orig[n](one_mapinfo, **kw)
            return f
        for n in names_lc:
            if n:
                setattr(m, n, mk(n))
        names_fast = [n + '_fast' for n in names_lc[:2] if n]
        orig_fast = {n: getattr(m, n) for n in names_fast}

        def mkf(n):
            def f(one_mapinfo, **kw):
                cap['fast'] = names_fast.index(n)
                return orig_fast[n](one_mapinfo, **kw)
            return f
        for n in names_fast:
            setattr(m, n, mkf(n))
        # V4: every call of the segment-surgery functions during the read's extend_func run(s) (:19238; a second run follows with
        # nofilter when pairedindel fires, :24079-24080): inputs and outputs as segment-row arrays
        v4log = []
        v4orig = {n: getattr(m, n) for n in ('rebuild_chain_break', 'drop_misplaced_alignment_test', 'merge_conjacent_alignment', 'fix_simple_inv')} if v4 else {}
        if v4:
            def put(tag, al):
                kk = '%s_v4_%d_%s' % (key, len(v4log), tag)
                arrays[kk] = seg_rows(al)
                return kk

            def w_rebuild(contig2start, raw, **kw):
                e = {'fn': 'rebuild_chain_break', 'large_cost': int(kw.get('large_cost', 0))}
                e['in'] = put('in', [list(raw)])
                out = v4orig['rebuild_chain_break'](contig2start, raw, **kw)
                e['out'] = put('out', out); v4log.append(e)
                return out

            def w_drop(al, iloc, **kw):
                snap = [list(x) for x in al]
                rem = v4orig['drop_misplaced_alignment_test'](al, iloc, **kw)
                first = not any(x['fn'] == 'drop_misplaced_alignment_test' and x['run'] == sum(1 for y in v4log if y['fn'] == 'rebuild_chain_break') for x in v4log)
                e = {'fn': 'drop_misplaced_alignment_test', 'iloc': int(iloc), 'removed': bool(rem), 'run': sum(1 for y in v4log if y['fn'] == 'rebuild_chain_break')}
                if rem or first:         # snapshots only where something happens (and once per run, to pin the state the loop starts from)
                    e['in'] = put('in', snap); e['out'] = put('out', al)
                v4log.append(e)
                return rem

            def w_merge(al, contig2start):
                e = {'fn': 'merge_conjacent_alignment'}
                e['in'] = put('in', al)
                v4orig['merge_conjacent_alignment'](al, contig2start)
                e['out'] = put('out', al); v4log.append(e)

            def w_fix(al, contig2start, contig2seq, testseq):
                e = {'fn': 'fix_simple_inv'}
                e['in'] = put('in', al)
                v4orig['fix_simple_inv'](al, contig2start, contig2seq, testseq)
                e['out'] = put('out', al); v4log.append(e)
            m.rebuild_chain_break, m.drop_misplaced_alignment_test, m.merge_conjacent_alignment, m.fix_simple_inv = w_rebuild, w_drop, w_merge, w_fix
        refrun.DPLOG = []
        try:
            st, one = ctx.align(rname, seq)
        finally:
            for n, f in v4orig.items():
                setattr(m, n, f)
        dplog = refrun.DPLOG; refrun.DPLOG = None
        if v4:
            rec['v4'] = v4log
        for n in names_lc:
            if n:
                setattr(m, n, orig[n])
        for n in names_fast:
            setattr(m, n, orig_fast[n])
        rec['v3_fast'] = cap.get('fast', -1)
        rec['v6_status'] = st
        rec['v6_records'] = [[t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), t[8]] for t in one]
        if 'raw' in cap:
            rec['v3_variant'] = cap['variant']; rec['v3_kw'] = cap['kw']
            raw = cap['raw'].astype(np.int64).reshape(-1, 4)
            rec['v3_raw_n'] = int(len(raw)); rec['v3_raw_crc'] = zlib.crc32(np.ascontiguousarray(raw).tobytes())
            if len(raw) <= 20000 and not light:        # dense and bulk cases: only the count and a checksum of the raw local anchors travel
                arrays[key + '_v3_raw'] = raw
            if scores != 0:
                need_rev = scores < 0
                rd = seq if not need_rev else synth.tostr(synth.revcomp(np.frombuffer(seq.encode(), np.uint8)))
                from numba.typed import List
                npl = List([np.array(p) for p in rpl])
                rc_rd = synth.tostr(synth.revcomp(np.frombuffer(rd.encode(), np.uint8)))
                lsc, lpath = m.get_localmap_multi_all_forDP_inv_guide_list(
                    npl, rd, rc_rd, ctx.contig2start, ctx.contig2seq, kmersize=9, skipcost=ctx.option['local_skipcost'],
                    maxdiff=ctx.option['local_maxdiff'], maxgap=(50 if mode == 'L' else 99), shift=1)
                rec['v3_score'] = float(lsc)
                arrays[key + '_v3_path'] = np.array(lpath, dtype=np.int64).reshape(-1, 4)
        # V5: DP problems, compact (kind, tl, ql, crc32(t), crc32(q))
        rec['v5'] = [[kd, len(t), len(q), zlib.crc32(t.encode()), zlib.crc32(q.encode())] for kd, t, q in dplog]
        case['reads'].append(rec)
    meta[cid] = case


def main():
    os.makedirs(GOLD, exist_ok=True)
    arrays, meta = {}, {}
    # case A: the reference's testdata (config 1)
    ref = refrun.read_fasta('/root/reference/testdata/reference.fasta')
    rds = refrun.read_fasta('/root/reference/testdata/read.fasta')
    run_case('A', 'H', [n for n, _ in ref], [s.upper() for _, s in ref], [(n, s.upper()) for n, s in rds], 15, arrays, meta, v4=True)
    # case B: ONT-shape reads from an SV donor (mode H)
    L = 120000
    contigs = synth.make_reference([L, 40000], seed=101)
    ops = [('INV', 10000, 2500), ('DEL', 25000, 900), ('INS', 40000, 500, 5), ('DUP', 55000, 1800, 2), ('INVDUP', 70000, 1500),
           ('INV', 85000, 400), ('DEL', 95000, 200)]
    d0 = synth.implant_svs(contigs[0], ops)
    d0 = np.concatenate([d0[:105000], contigs[1][2000:5000], d0[105000:]])
    rB = synth.sample_reads([d0, contigs[1]], 8, mean_len=7000, err=0.10, seed=102, shape='ont', min_len=2000, max_len=12000)
    run_case('B', 'H', ['chrA', 'chrB'], [synth.tostr(c) for c in contigs], [(n, synth.tostr(s)) for n, s, _ in rB], 15, arrays, meta, v4=True)
    # case C: HiFi-shape reads (mode L, k=19)
    rC = synth.sample_reads([d0, contigs[1]], 5, mean_len=8000, err=0.005, seed=103, shape='hifi', min_len=4000, sd=1500)
    run_case('C', 'L', ['chrA', 'chrB'], [synth.tostr(c) for c in contigs], [(n, synth.tostr(s)) for n, s, _ in rC], 19, arrays, meta, v4=True)
    # case D: chimeras, repeats, unmappable, short and N-bearing reads (mode H)
    rng = np.random.default_rng(104)
    c0 = contigs[0].copy()
    elem = synth.make_reference([2000], seed=105)[0]
    for t in range(12):
        p = int(rng.integers(0, len(c0) - 2000)); c0[p:p + 2000] = synth.mutate(elem, 0.02, rng, ratio=(1, 0, 0))[:2000]
    unit = synth.make_reference([31], seed=106)[0]
    c0[60000:60000 + 31 * 60] = np.tile(unit, 60)
    rr = synth.sample_reads([c0, contigs[1]], 6, mean_len=6000, err=0.08, seed=107, shape='ont', min_len=3000, max_len=9000)
    rD = [('chim0', np.concatenate([rr[0][1][:3000], rr[1][1][1000:]])), ('chim1', np.concatenate([rr[2][1][:2500], synth.revcomp(rr[3][1][:3000])])),
          ('rep0', rr[4][1]), ('rand0', synth.make_reference([3000], seed=108)[0]), ('short0', rr[5][1][:200]), ('tiny0', rr[5][1][:12])]
    wn = rr[5][1].copy(); wn[500:530] = ord('N'); wn[2000] = ord('N')
    rD.append(('withN', wn))
    tandem = synth.mutate(c0[59000:63500], 0.08, rng)
    rD.append(('tandem0', tandem))
    run_case('D', 'H', ['chrA', 'chrB'], [synth.tostr(c0), synth.tostr(contigs[1])], [(n, synth.tostr(s)) for n, s in rD], 15, arrays, meta, v4=True)
    # cases E (mode L) and F (mode H): repeat-dense reads that drive the *_fast chain variants (G3, L5): a 2.5 kb element in 64 copies
    # (GC-fast: more than 5 anchors per read base), a period-29 tandem array with a diverged second copy on the other contig
    # (more than one guide chain + dense local anchors: LC-mm-fast) and a period-23 array without a copy (LC-fast)
    rng = np.random.default_rng(120)
    e0, e1 = synth.make_reference([420000, 60000], seed=121)
    elem = synth.make_reference([2500], seed=122)[0]
    step = (len(e0) - 40000) // 65
    starts = []
    for t in range(64):
        p = 3000 + t * step
        e0[p:p + 2500] = synth.mutate(elem, 0.01, rng, ratio=(1, 0, 0))[:2500]
        starts.append(p)
    t1 = len(e0) - 32000
    e0[t1:t1 + 29 * 240] = np.tile(synth.make_reference([29], seed=123)[0], 240)
    t2 = len(e0) - 16000
    e0[t2:t2 + 23 * 400] = np.tile(synth.make_reference([23], seed=124)[0], 400)
    seg = synth.mutate(e0[t1 - 2500:t1 + 6960 + 2500], 0.004, rng, ratio=(1, 0, 0))
    e1[20000:20000 + len(seg)] = seg
    cE = [synth.tostr(e0), synth.tostr(e1)]
    rE = [('elem%d' % i, synth.mutate(e0[starts[7 * i + 3] - 300 + 100 * i:starts[7 * i + 3] + 2900], 0.005, rng)) for i in range(3)]
    run_case('E', 'L', ['chrA', 'chrB'], cE, [(n, synth.tostr(x)) for n, x in rE], 19, arrays, meta)
    rF = [('dupArray0', synth.mutate(e0[t1 - 1500:t1 + 6960 + 1500], 0.08, rng)),
          ('dupArray1', synth.revcomp(synth.mutate(e0[t1 + 6960 - 2500:t1 + 6960 + 2500], 0.08, rng))),
          ('soloArray0', synth.mutate(e0[t2 - 1500:t2 + 9200 + 1500], 0.08, rng))]
    run_case('F', 'H', ['chrA', 'chrB'], cE, [(n, synth.tostr(x)) for n, x in rF], 15, arrays, meta)
    # case G: mode R (mammap_noprefercloser.py; BASELINE config 5 shape): reads from the SV donor of case B + a chimera + an unmappable read
    rG = synth.sample_reads([d0, contigs[1]], 6, mean_len=7000, err=0.08, seed=130, shape='ont', min_len=2500, max_len=11000)
    lG = [(n_, synth.tostr(s_)) for n_, s_, _ in rG]
    lG.append(('chimR', synth.tostr(np.concatenate([rG[0][1][:2500], synth.revcomp(rG[1][1][:2500])]))))
    lG.append(('randR', synth.tostr(synth.make_reference([2500], seed=131)[0])))
    run_case('G', 'R', ['chrA', 'chrB'], [synth.tostr(c) for c in contigs], lG, 15, arrays, meta, v4=True)
    # case H: BASELINE configs[4] shape — a donor made by the vacsim-GRAMMAR implanter (vacmap_amd/vacsim.py: Specified{} / Random{} lines with nested
    # INV, DUP:..:rev:times, TRA between the two contigs, NML spacers), reads across the implanted complex SVs on both strands, mode R
    from vacmap_amd import vacsim
    hc = synth.make_reference([150000, 90000], seed=141)
    donorH, piecesH, eventsH = vacsim.implant(hc, VACSIM_TEXT_H, seed=7)
    rngH = np.random.default_rng(142)
    lH = []
    for ei in range(0, len(eventsH), max(1, len(eventsH) // 7)):
        ev = eventsH[ei]; ci = ev['contig']
        dpos = [ds + ev['start'] - ss for ds, de, sc, ss, se, stx in piecesH[ci] if sc == ci and stx > 0 and ss <= ev['start'] <= se]
        if not dpos:
            continue
        a = max(0, dpos[0] - 3500); b = min(len(donorH[ci]), dpos[0] + 4500)
        frag = donorH[ci][a:b]
        if len(lH) % 2:
            frag = synth.revcomp(frag)
        lH.append(('sv%d_%s' % (ev['sv'], ev['type']), synth.tostr(synth.mutate(frag, 0.06, rngH))))
    run_case('H', 'R', ['chrA', 'chrB'], [synth.tostr(c) for c in hc], lH, 15, arrays, meta, v4=True)
    meta['H']['vacsim_text'] = VACSIM_TEXT_H; meta['H']['vacsim_seed'] = 7; meta['H']['ref_seed'] = 141
    # case I: inputs built to drive the rare branches of the segment surgery (V4): inversions whose breakpoints carry a 6 / 9 / 12 bp
    # inverted repeat (fix_simple_inv :24226 shifts the breakpoints across the micro-homology) and small segments copied from 1.5-4 kb
    # downstream between paired DEL / INS (drop_misplaced_alignment_test :726 removes them)
    ri_ = synth.make_reference([200000], seed=151)[0]
    rngI = np.random.default_rng(152)
    sites = [(60000 + 7000 * b, 3000, b) for b in (6, 9, 12)]
    for p_, ml_, b_ in sites:
        ri_[p_ + ml_ - b_:p_ + ml_] = synth.revcomp(ri_[p_:p_ + b_])
    lI = []
    for p_, ml_, b_ in sites:
        don = np.concatenate([ri_[:p_], synth.revcomp(ri_[p_:p_ + ml_]), ri_[p_ + ml_:]])
        lI.append(('inv_ir%d' % b_, synth.tostr(synth.mutate(don[p_ - 4000:p_ + ml_ + 4000], 0.005, rngI))))
    for sz_, sh_ in ((300, 2500), (450, 4000), (700, 1500)):
        p_ = 30000
        don = np.concatenate([ri_[:p_], ri_[p_ + sh_:p_ + sh_ + sz_], ri_[p_ + 400:]])
        lI.append(('misplaced%d' % sz_, synth.tostr(synth.mutate(don[p_ - 5000:p_ + sz_ + 5000], 0.03, rngI))))
    run_case('I', 'H', ['chrA'], [synth.tostr(ri_)], lI, 15, arrays, meta, v4=True)
    # ---- round 3: a wider pin
    # case J: mode S (mammap_sensitive.py): ONT reads at 10-13 % error from the SV donor of case B, a chimera and an unmappable read
    rJ = synth.sample_reads([d0, contigs[1]], 18, mean_len=5000, err=0.12, seed=160, shape='ont', min_len=1500, max_len=9000)
    lJ = [(n_, synth.tostr(s_)) for n_, s_, _ in rJ]
    lJ.append(('chimS', synth.tostr(np.concatenate([rJ[0][1][:2000], synth.revcomp(rJ[1][1][:2500])]))))
    lJ.append(('randS', synth.tostr(synth.make_reference([2500], seed=161)[0])))
    run_case('J', 'S', ['chrA', 'chrB'], [synth.tostr(c) for c in contigs], lJ, 15, arrays, meta, light=True)
    # case K: mode L at k = 19, 24 more HiFi-shape reads (0.5 % and 1 % error) over the SV donor, both strands
    rK = synth.sample_reads([d0, contigs[1]], 16, mean_len=7000, err=0.005, seed=170, shape='hifi', min_len=3000, sd=2000)
    rK += synth.sample_reads([d0, contigs[1]], 8, mean_len=5000, err=0.01, seed=171, shape='hifi', min_len=2500, sd=1500)
    run_case('K', 'L', ['chrA', 'chrB'], [synth.tostr(c) for c in contigs], [('k%d_%s' % (i, n_), synth.tostr(s_)) for i, (n_, s_, _) in enumerate(rK)], 19, arrays, meta, light=True)
    # case M: reads of 40 kb and more (several gap-fill chunks, long chains): two ONT reads (mode H) and one HiFi read (mode L, case M2)
    bigc = synth.make_reference([260000], seed=180)[0]
    bigd = synth.implant_svs(bigc, [('INV', 60000, 3000), ('DEL', 110000, 1200), ('DUP', 150000, 2500, 2), ('INS', 200000, 700, 9)])
    rngM = np.random.default_rng(181)
    lM = [('long45k', synth.tostr(synth.mutate(bigd[30000:75000], 0.10, rngM))), ('long52k_rc', synth.tostr(synth.revcomp(synth.mutate(bigd[95000:147000], 0.09, rngM))))]
    run_case('M', 'H', ['chrL'], [synth.tostr(bigc)], lM, 15, arrays, meta, light=True)
    lM2 = [('hifi41k', synth.tostr(synth.mutate(bigd[140000:181000], 0.005, rngM)))]
    run_case('M2', 'L', ['chrL'], [synth.tostr(bigc)], lM2, 19, arrays, meta, light=True)
    # case N: the remaining branch of fix_simple_inv (:24226-24312, `refen_0 < refst_1`: the left flank ends BEFORE the inverted segment's
    # reference start and the right flank starts as many bases early). Found by tools/harness/find_inv_branch.py (seed 547): an inversion whose
    # two breakpoints carry 25-bp inverted-repeat arms, 1 % error
    rngN = np.random.default_rng(547)
    rN = synth.make_reference([200000], seed=151)[0]
    pN = 60000 + int(rngN.integers(0, 50000)); mlN = int(rngN.integers(900, 3000)); bN = int(rngN.integers(5, 30))
    kindN = int(rngN.integers(0, 4))
    assert kindN == 2, kindN
    rN[pN - bN:pN] = synth.revcomp(rN[pN + mlN - bN:pN + mlN]); rN[pN + mlN:pN + mlN + bN] = synth.revcomp(rN[pN:pN + bN])
    donN = np.concatenate([rN[:pN], synth.revcomp(rN[pN:pN + mlN]), rN[pN + mlN:]])
    errN = float(rngN.choice([0.0, 0.003, 0.01]))
    lN = [('inv_under_extended', synth.tostr(synth.mutate(donN[pN - 3500:pN + mlN + 3500], errN, rngN)))]
    run_case('N', 'H', ['chrA'], [synth.tostr(rN)], lN, 15, arrays, meta, v4=True)
    # case O: 30 more ONT reads (mode H) over a second SV donor on two contigs; case P: 20 more mode-R reads over the vacsim-grammar donor of case H
    oc = synth.make_reference([180000, 70000], seed=190)
    od = synth.implant_svs(oc[0], [('DEL', 20000, 400), ('INV', 45000, 1800), ('DUP', 80000, 1200, 3), ('INS', 110000, 350, 11), ('INVDUP', 140000, 1000)])
    rO = synth.sample_reads([od, oc[1]], 30, mean_len=6000, err=0.10, seed=191, shape='ont', min_len=1500, max_len=14000)
    run_case('O', 'H', ['chrA', 'chrB'], [synth.tostr(c) for c in oc], [(n_, synth.tostr(s_)) for n_, s_, _ in rO], 15, arrays, meta, light=True)
    rP = synth.sample_reads([donorH[0], donorH[1]], 20, mean_len=6000, err=0.07, seed=195, shape='ont', min_len=2000, max_len=12000)
    run_case('P', 'R', ['chrA', 'chrB'], [synth.tostr(c) for c in hc], [(n_, synth.tostr(s_)) for n_, s_, _ in rP], 15, arrays, meta, light=True)
    # V7: the reference's own known-answer tests for nm_from_cigar (tests/test_nm_from_cigar.py) — evaluate the inputs of each
    # test through the reference function and store (cigar, query, ref, expected NM)
    of = refrun.refload.load_output_functions()
    import ast
    src = open('/root/reference/tests/test_nm_from_cigar.py').read()
    v7 = []
    for fn in ast.parse(src).body:
        if not isinstance(fn, ast.FunctionDef):
            continue
        env = {}
        for st in fn.body:   # local string constants (test_mixed_ops)
            if isinstance(st, ast.Assign) and isinstance(st.value, ast.Constant):
                env[st.targets[0].id] = st.value.value
        for node in ast.walk(fn):
            if isinstance(node, ast.Compare) and isinstance(node.left, ast.Call) and getattr(node.left.func, 'id', '') == 'nm_from_cigar':
                args = [env[a.id] if isinstance(a, ast.Name) else ast.literal_eval(a) for a in node.left.args]
                expected = ast.literal_eval(node.comparators[0])
                got = int(of.nm_from_cigar(*args))
                assert got == expected, (fn.name, got, expected)
                v7.append({'test': fn.name, 'args': args, 'nm': expected})
    meta['V7_nm_from_cigar'] = v7
    np.savez_compressed(os.path.join(GOLD, 'cases.npz'), **arrays)
    json.dump(meta, open(os.path.join(GOLD, 'cases.json'), 'w'), indent=0, sort_keys=True)
    nrec = sum(len(r['v6_records']) for c in meta.values() if isinstance(c, dict) for r in c['reads'])
    print('cases', [k for k in meta], 'records', nrec, 'npz bytes', os.path.getsize(os.path.join(GOLD, 'cases.npz')),
          'json bytes', os.path.getsize(os.path.join(GOLD, 'cases.json')))


if __name__ == '__main__':
    main()
