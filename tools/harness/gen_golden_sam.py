#!/usr/bin/env python3
"""Golden vectors V8 for the SAM emitter (build container only): the reference's own get_bam_dict_str / _comments
(/root/reference/src/vacmap/mammap_clrnano.py:20841, :21022) run on the records of the existing goldens (tests/golden/cases.*), under
several option sets. Output: tests/golden/sam.json = inputs (case, read, records, options) + the reference's SAM lines.

Records carrying M operators are also given in =/X form (rewritten here from the sequences) so that the MD / cs code paths
(:19012, :19062) produce non-empty strings."""
import copy, json, os, re, sys
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, _HERE)
import refload
GOLD = os.path.join(_ROOT, 'tests', 'golden')
_COMP = bytes.maketrans(b'ACGTN', b'TGCAN')


def eqx(cigar, query, ref, q_st):
    """M -> =/X using the sequences; query is the read in alignment orientation, the CIGAR starts with its clip"""
    out = []; q = 0; r = 0
    for n, op in re.findall(r'(\d+)([MIDNSHP=X])', cigar):
        n = int(n)
        if op == 'M':
            run_op, run = None, 0
            for i in range(n):
                o = '=' if query[q + i] == ref[r + i] else 'X'
                if o == run_op:
                    run += 1
                else:
                    if run_op: out.append('%d%s' % (run, run_op))
                    run_op, run = o, 1
            out.append('%d%s' % (run, run_op))
            q += n; r += n
        else:
            out.append('%d%s' % (n, op))
            if op in 'IS': q += n
            elif op in 'DN': r += n
            elif op in '=X': q += n; r += n
    return ''.join(out)


def digest(line):
    import hashlib
    return hashlib.sha256(line.encode()).hexdigest()


def head(line):
    import zlib
    return '\t'.join(c if len(c) <= 60 else '%s..[%d:%08x]' % (c[:24], len(c), zlib.crc32(c.encode())) for c in line.split('\t'))


def main():
    import refrun                     # registers the vacmap_index shim the reference module imports
    m = refrun.load('H') if hasattr(refrun, 'load') else refload.load('H')
    meta = json.load(open(os.path.join(GOLD, 'cases.json')))
    arrays = np.load(os.path.join(GOLD, 'cases.npz'))
    out = []
    optsets = [dict(md=False, shortcs=True, cigar2cg=False, markunbalancetra=True, H=False, fakecigar=False),
               dict(md=False, shortcs=True, cigar2cg=False, markunbalancetra=False, H=True, fakecigar=True, rg='grp1'),
               dict(md=True, shortcs=True, cigar2cg=True, markunbalancetra=True, H=False, fakecigar=False, eqx=True),
               dict(md=True, shortcs=False, cigar2cg=False, markunbalancetra=True, H=True, fakecigar=True, eqx=True, comments='XC:Z:kept\\tNM:i:9\\tbad\\tYY:q:1'),
               # the driver's default: no --rg-* option given -> read group {'ID': '1', 'SM': 'sample'}, rg-id '1' (src/vacmap/vacmap:211-214)
               dict(md=False, shortcs=True, cigar2cg=False, markunbalancetra=True, H=False, fakecigar=False, rg='1'),
               # --MD without --eqx: M CIGARs, empty MD / cs strings (vacmap:192, :19143)
               dict(md=True, shortcs=True, cigar2cg=False, markunbalancetra=True, H=False, fakecigar=False, rg='1')]
    for cid in ('A', 'B', 'D', 'G'):
        c = meta[cid]
        contigs = {n: arrays['%s_contig%d' % (cid, i)].tobytes().decode() for i, n in enumerate(c['names'])}
        for ri, r in enumerate(c['reads']):
            if not r['v6_records']:
                continue
            query = arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode()
            rcq = query.encode().translate(_COMP)[::-1].decode()
            qual = ''.join(chr(33 + (7 * i) % 40) for i in range(len(query)))
            for oi, o in enumerate(optsets):
                recs = []
                for t in r['v6_records']:
                    t = list(t)
                    rec = [r['name'], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]]
                    if o.get('eqx'):
                        rec[8] = eqx(rec[8], query if rec[2] == '+' else rcq, contigs[rec[1]][rec[5]:rec[6]], rec[3])
                    recs.append(tuple(rec))
                option = {'H': o['H'], 'fakecigar': o['fakecigar']}
                if 'rg' in o:
                    option['rg-id'] = o['rg']
                inp = copy.deepcopy(recs)
                try:
                    if 'comments' in o:
                        lines = m.get_bam_dict_str_comments(copy.deepcopy(recs), query, qual if oi != 1 else None, o['comments'].replace('\\t', '\t'), {}, contigs, o['md'], o['shortcs'], o['cigar2cg'],
                                                            o['markunbalancetra'], option)
                    else:
                        lines = m.get_bam_dict_str(copy.deepcopy(recs), query, qual if oi != 1 else None, {}, contigs, o['md'], o['shortcs'], o['cigar2cg'], o['markunbalancetra'], option)
                    raised = None
                except Exception as e:       # the worker skips such reads (:24127-24134): the emitter must raise too
                    lines, raised = None, type(e).__name__
                # the inputs are re-derived by the test from cases.json (+ the M -> =/X rewrite of tests/sam_cases.py); a line is stored as its
                # sha256 plus a readable head (SEQ / QUAL / long CIGAR-bearing columns replaced by length:crc32)
                out.append({'case': cid, 'read': ri, 'opt': o, 'qual': oi != 1, 'digest': None if lines is None else [digest(x) for x in lines],
                            'head': None if lines is None else [head(x) for x in lines], 'raised': raised})
    json.dump(out, open(os.path.join(GOLD, 'sam.json'), 'w'), indent=0)
    print('sam goldens', len(out), 'entries,', sum(len(x['digest'] or []) for x in out), 'lines,', sum(1 for x in out if x['raised']), 'raised,',
          os.path.getsize(os.path.join(GOLD, 'sam.json')), 'bytes')


if __name__ == '__main__':
    main()
