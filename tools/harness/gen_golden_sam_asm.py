#!/usr/bin/env python3
"""Golden SAM lines of -mode asm (build container only): the reference's own iterator_get_bam_dict_str / _comments
(/root/reference/src/vacmap/mammap_asm.py:22757, :22942) on the records of the asm goldens (tests/golden/asm.*: cases AS1, AS5 and the 600 kb
contig of AS2 whose long CIGAR is recomputed by the oracle, which the record goldens pin), under several option sets.
Output: tests/golden/sam_asm.json (options + per line a sha256 and a readable head)."""
import copy, json, os, sys, zlib
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, _HERE); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import refrun
from gen_golden_sam import digest, head
GOLD = os.path.join(_ROOT, 'tests', 'golden')


def records_of(O, meta, arr, cid, ci):
    """the contig's 9-tuples: from the golden when every CIGAR is stored whole, else from the oracle (checked against the golden's crc)"""
    c = meta[cid]; g = c['contigs'][ci]
    if all(len(r) == 10 for r in g['records']):
        return [(g['name'], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[9]) for r in g['records']]
    oi = O.Index.from_seqs(c['names'], [arr['%s_ref%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))], k=c['k'], w=c['w'])
    seq = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
    rc, recs = O.align_asm(oi, seq, O.params('asm'), *c['sizes'])
    out = []
    for r, t in zip(g['records'], recs):
        assert zlib.crc32(t[8].encode()) == r[8] and len(t[8]) == r[7]
        out.append((g['name'], c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]))
    return out


def main():
    import oracle_lib as O
    m = refrun.load('asm')
    meta = json.load(open(os.path.join(GOLD, 'asm.json'))); arr = np.load(os.path.join(GOLD, 'asm.npz'))
    optsets = [dict(md=False, shortcs=True, cigar2cg=False, markunbalancetra=False, H=False, fakecigar=False, rg='1'),
               dict(md=False, shortcs=True, cigar2cg=False, markunbalancetra=True, H=True, fakecigar=True, rg='grp1'),
               dict(md=True, shortcs=True, cigar2cg=True, markunbalancetra=False, H=False, fakecigar=False),
               dict(md=True, shortcs=False, cigar2cg=False, markunbalancetra=True, H=True, fakecigar=True, comments='XC:Z:kept\\tNM:i:9\\tbad\\tYY:q:1')]
    out = []
    for cid, cis in (('AS1', range(10)), ('AS5', range(4)), ('AS2', [0])):
        c = meta[cid]
        contigs = {n: arr['%s_ref%d' % (cid, i)].tobytes().decode() for i, n in enumerate(c['names'])}
        for ci in cis:
            recs = records_of(O, meta, arr, cid, ci)
            if not recs:
                continue
            query = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
            for oi_, o in enumerate(optsets):
                if cid == 'AS2' and oi_ not in (0, 2):
                    continue
                option = {'H': o['H'], 'fakecigar': o['fakecigar']}
                if 'rg' in o:
                    option['rg-id'] = o['rg']
                try:
                    if 'comments' in o:
                        lines = list(m.iterator_get_bam_dict_str_comments(copy.deepcopy(recs), query, None, o['comments'].replace('\\t', '\t'), {}, contigs, o['md'], o['shortcs'],
                                                                         o['cigar2cg'], o['markunbalancetra'], option))
                    else:
                        lines = list(m.iterator_get_bam_dict_str(copy.deepcopy(recs), query, None, {}, contigs, o['md'], o['shortcs'], o['cigar2cg'], o['markunbalancetra'], option))
                    raised = None
                except Exception as e:
                    lines, raised = None, type(e).__name__
                out.append({'case': cid, 'contig': ci, 'opt': o, 'digest': None if lines is None else [digest(x) for x in lines],
                            'head': None if lines is None else [head(x) for x in lines], 'raised': raised})
    json.dump(out, open(os.path.join(GOLD, 'sam_asm.json'), 'w'), indent=0)
    print('asm sam goldens', len(out), 'entries,', sum(len(x['digest'] or []) for x in out), 'lines,', sum(1 for x in out if x['raised']), 'raised')


if __name__ == '__main__':
    main()
