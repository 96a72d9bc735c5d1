"""SAM emitter (vacmap_amd/sam.py) against the reference's own get_bam_dict_str / _comments output captured in
tests/golden/sam.json (tools/harness/gen_golden_sam.py): default options, hard clip + approximate SA CIGARs + RG, MD / cs short and long
on =/X CIGARs, CG tag switch, comment copying. CPU only."""
import json, os
import pytest
import sam_cases as SC
from vacmap_amd import sam

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_sam_lines_match_reference(golden):
    meta, arrays = golden
    entries = json.load(open(os.path.join(GOLD, 'sam.json')))
    assert len(entries) >= 80
    nlines = 0
    for e in entries:
        recs, query, qual, contigs = SC.inputs(e, meta, arrays)
        o = e['opt']
        kw = dict(md=o['md'], shortcs=o['shortcs'], cigar2cg=o['cigar2cg'], markunbalancetra=o['markunbalancetra'], hardclip=o['H'],
                  fakecigar=o['fakecigar'], rg_id=o.get('rg'), comments=o['comments'].replace('\\t', '\t') if 'comments' in o else None)
        if e['raised']:
            with pytest.raises(Exception):
                sam.sam_lines(recs, query, qual, lambda c, a, b: contigs[c][a:b], **kw)
            continue
        lines = sam.sam_lines(recs, query, qual, lambda c, a, b: contigs[c][a:b], **kw)
        assert [SC.head(x) for x in lines] == e['head'], (e['case'], e['read'], o)
        assert [SC.digest(x) for x in lines] == e['digest'], (e['case'], e['read'], o)
        nlines += len(lines)
    assert nlines >= 100


def test_nm_from_cigar_known_answers(golden):
    """the reference's own tests/test_nm_from_cigar.py vectors (golden V7)"""
    meta, _ = golden
    for v in meta['V7_nm_from_cigar']:
        assert sam.nm_from_cigar(*v['args']) == v['nm'], v['test']


def test_header_and_format():
    h = sam.header_lines([('chr1', 1000), ('chr2', 50)], 'vacmap -ref r.fa', rg={'ID': 'g', 'SM': 's'})
    assert h[0] == '@HD\tVN:1.0' and h[1] == '@SQ\tSN:chr1\tLN:1000' and h[3] == '@RG\tID:g\tSM:s' and h[4].startswith('@PG\tID:VACmap\tPN:VACmap\tVN:1.0.2\tCL:')
    assert sam.format_line({'QNAME': 'r', 'FLAG': '0', 'NM': 3, 'XX': 1.5}) == 'r\t0\t*\t0\t255\t*\t*\t0\t0\t*\t*\tNM:i:3\tXX:f:1.5'
