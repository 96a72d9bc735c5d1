"""Own SV implanter that honours the vacsim grammar (BASELINE configs[4]).

Grammar and semantics follow /root/reference/vacsim (README.md:20-33, example_parameterfile; vacsim.py:93-142 `decode_sim_sv_info`,
:143-166 `add_SV`, :433-462 `decode_parameterfile`); the reference tool itself depends on mappy / Bio / heapdict and on SURVIVOR for
reads, so this is a from-scratch, seeded NumPy implementation that ALSO returns the truth a test needs:

    Specified{DEL:100:200,INS:100:1000,INV:100:200,DUP:100:200:0:4,TRA:200:400:1;number=2}
    Random{eventset=["DEL:100:200,INV:100:200","DUP:100:200","TRA:200:400"];eventcount=[4,20];number=2}

One line describes `number` complex SVs. A complex SV is a chain of events laid out left to right on one contig, consecutive events
separated by 200 unaltered bases (the spacer `match = 200`, vacsim.py:96; it also precedes the first event), each with a length drawn
uniformly from [lo, hi):
    DEL:lo:hi             the segment is removed
    INS:lo:hi             a random sequence is inserted (no reference base consumed)
    INV:lo:hi             the segment is replaced by its reverse complement
    DUP:lo:hi:rev:times   `times` extra copies of the segment (reverse-complemented when rev = 1) follow it
    TRA:lo:hi:rev         the segment is exchanged with an equally long segment of ANOTHER contig (both reverse-complemented when
                          rev = 1); the partner segments of one complex SV are laid out on the second contig 200 bases apart
    NML:lo:hi             unaltered bases (an explicit spacer inside the chain)
`Random{}` draws `eventcount` in [lo, hi) and then items of `eventset` (an item may itself be a comma-separated chain) until that many
non-NML events are placed; DUP drawn this way is `:0:1`, TRA gets a random `rev` (vacsim.py:415-428).

`implant(contigs, text, seed)` places the complex SVs at random non-overlapping loci (200 bases clear of contig ends and of each
other), edits the contigs and returns
    donor     list of uint8 arrays (one per contig)
    pieces    per contig: the donor as a list of (donor_start, donor_end, src_contig or -1 for inserted bases, src_start, src_end, strand)
    events    truth: one dict per event {sv, type, contig, start, end, (contig2, start2, end2), rev, times}
"""
import re

import numpy as np

from .synth import _ACGT, revcomp

MATCH = 200          # spacer between consecutive events (vacsim.py:96)
EDGE = 200           # clearance to contig ends / other SVs (vacsim.py:170 `edge_size`)


def parse(text, rng):
    """parameter text -> list of complex SVs, each a list of (type, length, rev, times) with lengths already drawn"""
    svs = []
    for line in text.splitlines():
        line = ''.join(line.split())
        if not line or line.startswith('#'):
            continue
        m = re.fullmatch(r'(Specified|Random)\{(.*)\}', line)
        if not m:
            continue                                            # other lines are ignored (vacsim.py:441-442)
        kind, body = m.group(1), m.group(2)
        if kind == 'Specified':
            chain, _, tail = body.partition(';')
            number = int(tail.split('=')[1])
            for _ in range(number):
                svs.append([_draw(op, rng) for op in chain.split(',')])
        else:
            fields = dict(f.split('=', 1) for f in body.split(';'))
            eventset = re.findall(r'"([^"]*)"', fields['eventset'])
            lo, hi = [int(x) for x in fields['eventcount'].strip('[]').split(',')]
            for _ in range(int(fields['number'])):
                want = int(rng.integers(lo, hi)); have = 0; chain = []
                while have < want:
                    for op in eventset[int(rng.integers(0, len(eventset)))].split(','):
                        t = op.split(':')[0]
                        if t == 'DUP' and op.count(':') == 2:
                            op += ':0:1'
                        elif t == 'TRA' and op.count(':') == 2:
                            op += ':%d' % int(rng.integers(0, 2))
                        chain.append(_draw(op, rng))
                        have += t != 'NML'
                svs.append(chain)
    return svs


def _draw(op, rng):
    f = op.split(':')
    t, lo, hi = f[0], int(f[1]), int(f[2])
    if t not in ('DEL', 'INS', 'INV', 'DUP', 'TRA', 'NML'):
        raise ValueError('unknown event %r' % op)
    ln = int(rng.integers(lo, hi))
    rev = int(f[3]) if len(f) > 3 else 0
    times = int(f[4]) if len(f) > 4 else 1
    return (t, ln, rev, times)


def _spans(chain):
    """reference bases a complex SV occupies on its own contig and on the TRA partner contig (vacsim.py:93-131)"""
    s1, s2 = MATCH, MATCH
    for t, ln, rev, times in chain:
        if t != 'INS':
            s1 += ln
        if t == 'TRA':
            s2 += ln + MATCH
        s1 += MATCH
    return s1, (s2 if any(t == 'TRA' for t, _, _, _ in chain) else 0)


def implant(contigs, text, seed=0):
    rng = np.random.default_rng(seed)
    contigs = [np.asarray(c, dtype=np.uint8) for c in contigs]
    svs = parse(text, rng)
    free = [[(EDGE, len(c) - EDGE)] for c in contigs]           # usable intervals per contig

    def take(ci, span):
        """a random sub-interval of length `span` out of contig ci's usable intervals (uniform over feasible starts); splits the interval"""
        cand = [(a, b) for a, b in free[ci] if b - a >= span]
        if not cand:
            return None
        wts = np.array([b - a - span + 1 for a, b in cand], dtype=np.float64)
        a, b = cand[int(rng.choice(len(cand), p=wts / wts.sum()))]
        st = int(rng.integers(a, b - span + 1))
        free[ci].remove((a, b))
        if st - EDGE - a > 0:
            free[ci].append((a, st - EDGE))
        if b - (st + span + EDGE) > 0:
            free[ci].append((st + span + EDGE, b))
        return st

    edits = [[] for _ in contigs]        # per contig: (ref_start, ref_end, replacement pieces [(src_contig, src_start, src_end, strand) or ('ins', bases)])
    events = []
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    for si, chain in enumerate(sorted(svs, key=lambda ch: -_spans(ch)[0])):      # largest first (vacsim.py:459-460)
        s1, s2 = _spans(chain)
        for _attempt in range(50):
            c1 = int(rng.choice(len(contigs), p=lens / lens.sum()))
            c2 = -1
            if s2:
                others = [i for i in range(len(contigs)) if i != c1]
                if not others:
                    raise ValueError('TRA needs a second contig')
                c2 = int(rng.choice(others, p=lens[others] / lens[others].sum()))
            keep = [list(f) for f in free]
            p1 = take(c1, s1)
            p2 = take(c2, s2) if s2 else 0
            if p1 is not None and p2 is not None:
                break
            free[:] = keep
        else:
            raise ValueError('no room left for a complex SV of %d bases' % s1)
        a, b = p1 + MATCH, p2 + MATCH
        for t, ln, rev, times in chain:
            ev = {'sv': si, 'type': t, 'contig': c1, 'start': a, 'end': a + (0 if t == 'INS' else ln), 'rev': rev, 'times': times}
            if t == 'DEL':
                edits[c1].append((a, a + ln, []))
            elif t == 'INS':
                edits[c1].append((a, a, [('ins', _ACGT[rng.integers(0, 4, size=ln)])]))
                ev['length'] = ln
            elif t == 'INV':
                edits[c1].append((a, a + ln, [(c1, a, a + ln, -1)]))
            elif t == 'DUP':
                edits[c1].append((a + ln, a + ln, [(c1, a, a + ln, -1 if rev else 1)] * times))
            elif t == 'TRA':
                s = -1 if rev else 1
                edits[c1].append((a, a + ln, [(c2, b, b + ln, s)]))
                edits[c2].append((b, b + ln, [(c1, a, a + ln, s)]))
                ev.update(contig2=c2, start2=b, end2=b + ln)
                b += ln + MATCH
            if t != 'NML':
                events.append(ev)
            if t != 'INS':
                a += ln
            a += MATCH
    donor, pieces = [], []
    for ci, c in enumerate(contigs):
        out, pc, pos, dpos = [], [], 0, 0

        def emit(arr, src):
            nonlocal dpos
            if len(arr):
                out.append(arr); pc.append((dpos, dpos + len(arr)) + src); dpos += len(arr)
        for st, en, rep in sorted(edits[ci], key=lambda e: (e[0], e[1])):
            emit(c[pos:st], (ci, pos, st, 1))
            for r in rep:
                if r[0] == 'ins':
                    emit(r[1], (-1, 0, len(r[1]), 1))
                else:
                    sc, ss, se, strand = r
                    seg = contigs[sc][ss:se]
                    emit(revcomp(seg) if strand < 0 else seg, (sc, ss, se, strand))
            pos = en
        emit(c[pos:], (ci, pos, len(c), 1))
        donor.append(np.concatenate(out) if out else np.zeros(0, np.uint8)); pieces.append(pc)
    return donor, pieces, events


def event_positions(pieces, events):
    """where the implanted events lie in the DONOR: (contig index array, donor position array), through the forward piece of an event's own
    contig that holds its left end (events whose left end fell into a moved or inverted piece are left out) — the `around` argument of
    synth.sample_reads_concat, so that reads are drawn across the SVs"""
    import numpy as np
    ev_c, ev_p = [], []
    for ci in range(len(pieces)):
        own = [(ss, se, ds) for ds, de, sc, ss, se, st in pieces[ci] if sc == ci and st > 0]
        keys = np.asarray([ev['start'] for ev in events if ev['contig'] == ci], dtype=np.int64)
        if not own or not len(keys):
            continue
        ss_a, se_a, ds_a = (np.asarray([o[i] for o in own], dtype=np.int64) for i in range(3))
        j = np.clip(np.searchsorted(ss_a, keys, side='right') - 1, 0, len(own) - 1)
        ok = (ss_a[j] <= keys) & (keys <= se_a[j])
        ev_c += [ci] * int(ok.sum()); ev_p += (ds_a[j][ok] + keys[ok] - ss_a[j][ok]).tolist()
    return np.asarray(ev_c, dtype=np.int64), np.asarray(ev_p, dtype=np.int64)


def read_truth(pieces_of_contig, d_start, d_end, strand, min_piece=1):
    """the reference pieces a donor interval [d_start, d_end) is made of, in READ order: [(src_contig, src_start, src_end, strand)];
    strand '-' reads see the pieces reversed and flipped. Pieces shorter than min_piece (after clipping) are dropped."""
    out = []
    for ds, de, sc, ss, se, st in pieces_of_contig:
        lo, hi = max(ds, d_start), min(de, d_end)
        if hi - lo < min_piece:
            continue
        if st > 0:
            a, b = ss + (lo - ds), ss + (hi - ds)
        else:
            a, b = se - (hi - ds), se - (lo - ds)
        out.append((sc, a, b, st))
    if strand < 0:
        out = [(sc, a, b, -st) for sc, a, b, st in out[::-1]]
    return out


def junctions(truth, min_flank=300):
    """adjacent pairs of reference pieces (both at least min_flank long, neither inserted sequence): the breakpoints a split alignment
    of the read should show — (contig, position, side) of the left piece's end and of the right piece's start in read direction"""
    js = []
    for (c1, a1, b1, s1), (c2, a2, b2, s2) in zip(truth, truth[1:]):
        if c1 < 0 or c2 < 0 or b1 - a1 < min_flank or b2 - a2 < min_flank:
            continue
        js.append(((c1, b1 if s1 > 0 else a1), (c2, a2 if s2 > 0 else b2)))
    return js


def record_junctions(records, min_indel=30):
    """breakpoints implied by a read's records (9-tuples with contig INDEX at [1]): between consecutive records in read order, plus
    indels of at least min_indel bases inside a CIGAR. Each junction = ((contig, ref pos leaving), (contig, ref pos entering))."""
    segs = []
    for r in records:
        contig, strand, q_st, q_en, r_st, r_en, cigar = r[1], r[2], r[3], r[4], r[5], r[6], r[8]
        # walk the CIGAR: sub-segments split at long indels; q in aligned-strand coordinates
        q, ref = 0, r_st
        cur_q, cur_r = None, None
        parts = []
        for n, op in re.findall(r'(\d+)([MIDNSHP=X])', cigar):
            n = int(n)
            if op in 'SH':
                q += n
            elif op in 'M=X':
                if cur_q is None:
                    cur_q, cur_r = q, ref
                q += n; ref += n
            elif op == 'I':
                if n >= min_indel and cur_q is not None:
                    parts.append((cur_q, q, cur_r, ref)); cur_q = None
                q += n
            elif op in 'DN':
                if n >= min_indel and cur_q is not None:
                    parts.append((cur_q, q, cur_r, ref)); cur_q = None
                ref += n
        if cur_q is not None:
            parts.append((cur_q, q, cur_r, ref))
        qlen = q
        for qa, qb, ra, rb in parts:
            if strand == '+':
                segs.append((qa, qb, contig, ra, rb, 1))
            else:                        # '-' records index the reverse-complemented read: back to the read's own coordinates
                segs.append((qlen - qb, qlen - qa, contig, ra, rb, -1))
    segs.sort()
    js = []
    for (qa1, qb1, c1, ra1, rb1, s1), (qa2, qb2, c2, ra2, rb2, s2) in zip(segs, segs[1:]):
        js.append(((c1, rb1 if s1 > 0 else ra1), (c2, ra2 if s2 > 0 else rb2)))
    return js


def matched(truth_js, rec_js, tol=50):
    """how many truth junctions have a record junction with both sides within tol (either orientation of the pair)"""
    def near(a, b):
        return a[0] == b[0] and abs(a[1] - b[1]) <= tol
    hit = 0
    for t in truth_js:
        if any((near(t[0], r[0]) and near(t[1], r[1])) or (near(t[0], r[1]) and near(t[1], r[0])) for r in rec_js):
            hit += 1
    return hit
