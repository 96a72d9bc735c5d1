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


def _raw_from_tuples(VL, recs, names):
    """a RawBatch-shaped object holding the 9-tuples of one read as vm_record structs + CIGAR blob"""
    import ctypes as C
    import numpy as np
    R = (VL.Record * max(len(recs), 1))(); blob = b''
    for i, r in enumerate(recs):
        cg = r[8].encode()
        R[i].read_idx = 0; R[i].contig = names.index(r[1]); R[i].strand = 1 if r[2] == '+' else -1; R[i].mapq = r[7]
        R[i].q_st, R[i].q_en, R[i].r_st, R[i].r_en = r[3], r[4], r[5], r[6]; R[i].cigar_off = len(blob); R[i].cigar_len = len(cg); blob += cg + b'\0'

    class Raw:
        pass
    raw = Raw(); raw.recs = R; raw.nrec = len(recs); raw.blob = C.c_char_p(blob); raw.status = np.zeros(1, np.int32); raw._keep = blob
    return raw


def test_native_sam_emitter_matches_reference(golden, oracle):
    """vm_sam_emit (the C++ emitter the driver uses at GPU rates) against the same reference lines as sam.py: all option sets, raises included"""
    import numpy as np
    import emu_lib, kernel_cases as KC
    from vacmap_amd import lib as VL
    ctx = emu_lib.context()
    meta, arrays = golden
    entries = json.load(open(os.path.join(GOLD, 'sam.json')))
    idx = {}
    nlines = nraised = 0
    for e in entries:
        cid = e['case']
        if cid not in idx:
            idx[cid] = KC._case_index(ctx, oracle, meta, arrays, cid)[0]
        recs, query, qual, contigs = SC.inputs(e, meta, arrays)
        o = e['opt']
        raw = _raw_from_tuples(VL, recs, meta[cid]['names'])
        opts = VL.SamOpts(int(o['md']), int(o['shortcs']), int(o['cigar2cg']), int(o['markunbalancetra']), int(o['H']), int(o['fakecigar']), o['rg'].encode() if 'rg' in o else None)
        nm = recs[0][0].encode()
        com = o['comments'].replace('\\t', '\t').encode() if 'comments' in o else None
        buf, off, nl, ns = VL.sam_emit(ctx.lib, idx[cid], opts, np.frombuffer(nm, np.uint8), [0, len(nm)], np.frombuffer(query.encode(), np.uint8), [0, len(query)], raw,
                                       quals=np.frombuffer(qual.encode(), np.uint8) if qual else None, qual_off=[0, len(qual)] if qual else None,
                                       comments=np.frombuffer(com, np.uint8) if com else None, com_off=[0, len(com)] if com else None, nthreads=2)
        lines = buf.tobytes().decode().split('\n')[:-1] if len(buf) else []
        if e['raised']:
            assert ns == 1 and not lines, (cid, e['read'], o)
            nraised += 1
            continue
        assert [SC.head(x) for x in lines] == e['head'], (cid, e['read'], o)
        assert [SC.digest(x) for x in lines] == e['digest'], (cid, e['read'], o)
        assert nl == len(lines) and off[-1] == len(buf)
        nlines += len(lines)
    assert nlines >= 150 and nraised >= 10


def test_native_fastx_reader(tmp_path):
    """vm_fastx_read: FASTA (multi-line, lower case, blank lines), FASTQ, gzip, comments after blank / tab, chunked reads, CRLF"""
    import gzip
    import emu_lib
    from vacmap_amd import lib as VL
    from vacmap_amd import driver
    emu_lib.context()
    L = emu_lib.context().lib
    fa = tmp_path / 'a.fa'
    fa.write_text('>r1 first comment\nACGTacgt\nNNAC\n\n>r2\tXC:Z:tab\nGG\n>r3\n\n>r4 x\r\nAC\r\nGT\r\n>r5\tRG:Z:a b\nAC\n>r6 c\td\nTT\n')
    fq = tmp_path / 'b.fq.gz'
    with gzip.open(fq, 'wt') as f:
        for i in range(70):
            f.write('@q%d c%d\n%s\n+\n%s\n' % (i, i, 'acgtn' * (i + 1), 'I' * (5 * (i + 1))))
    for path in (str(fa), str(fq)):
        exp = list(driver.read_fastx(path, want_comment=True))
        got = []
        rd = VL.Fastx(path, lib=L)
        while True:
            ch = rd.read(max_reads=16)
            if ch is None:
                break
            for i in range(len(ch['seqs_off']) - 1):
                f = lambda k: ch[k][ch[k + '_off'][i]:ch[k + '_off'][i + 1]].tobytes().decode()
                got.append((f('names'), f('seqs'), f('quals') or None, f('comments') or None))
        rd.close()
        assert got == [(n, s.upper(), q, c) for n, s, q, c in exp], path
    assert len(got) == 70
    # the name ends at the FIRST blank or tab (kseq, what mp.fastx_read does): a tab before a blank must not leak into QNAME
    rd = VL.Fastx(str(fa), lib=L); ch = rd.read(max_reads=16); rd.close()
    names = [ch['names'][ch['names_off'][i]:ch['names_off'][i + 1]].tobytes().decode() for i in range(6)]
    coms = [ch['comments'][ch['comments_off'][i]:ch['comments_off'][i + 1]].tobytes().decode() for i in range(6)]
    assert names == ['r1', 'r2', 'r3', 'r4', 'r5', 'r6'] and coms[4] == 'RG:Z:a b' and coms[5] == 'c\td'
    # a truncated .gz is an I/O error, not a clean end of input (the driver must not write a partial SAM and exit 0)
    import pytest
    raw = open(fq, 'rb').read()
    bad = tmp_path / 'trunc.fq.gz'
    bad.write_bytes(raw[:len(raw) // 2])
    rd = VL.Fastx(str(bad), lib=L)
    with pytest.raises(VL.VmxError):
        while rd.read(max_reads=16) is not None:
            pass
    rd.close()


def test_blob_gather_parts_restores_input_order():
    """vm_blob_gather_parts (the driver's writer): entries of several (blob, offsets) parts merged in ascending key order, empty entries and
    empty parts included"""
    import numpy as np
    import emu_lib
    from vacmap_amd import lib as VL
    L = emu_lib.context().lib
    rng = np.random.default_rng(3)
    n = 500
    texts = [bytes(rng.integers(65, 91, int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(n)]
    perm = rng.permutation(n)
    cuts = [0, 120, 120, 333, n]                                   # four parts, one of them empty
    blobs, offs, keys = [], [], []
    for a, b in zip(cuts, cuts[1:]):
        idx = np.sort(perm[a:b])                                   # a batch holds its reads in window order
        blobs.append(np.frombuffer(b''.join(texts[i] for i in idx), dtype=np.uint8))
        offs.append(np.concatenate([[0], np.cumsum([len(texts[i]) for i in idx])]).astype(np.int64))
        keys.append(idx.astype(np.int64))
    out = VL.blob_gather_parts(L, blobs, offs, keys)
    assert out.tobytes() == b''.join(texts)
    # the same merge written to a file descriptor (writev, > 1024 pieces: several calls; neighbours in memory coalesced)
    import os, tempfile
    big = [bytes(rng.integers(65, 91, int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(5000)]
    perm2 = rng.permutation(5000)
    b2, o2, k2 = [], [], []
    for a, b in ((0, 2500), (2500, 2500), (2500, 5000)):
        idx = np.sort(perm2[a:b])
        b2.append(np.frombuffer(b''.join(big[i] for i in idx) or b'\0', dtype=np.uint8)); k2.append(idx.astype(np.int64))
        o2.append(np.concatenate([[0], np.cumsum([len(big[i]) for i in idx])]).astype(np.int64))
    with tempfile.TemporaryFile() as f:
        f.write(b'head\n'); f.flush()
        w = VL.blob_write_parts(L, f.fileno(), b2, o2, k2)
        assert w == sum(len(t) for t in big)
        f.seek(0)
        assert f.read() == b'head\n' + b''.join(big)
    # the same through the shared mapping (vm_blob_write_parts_mmap: several threads copy into the file's own pages; the file grows from an offset
    # that is not page-aligned, twice in a row), then text written the ordinary way behind it
    huge = [bytes(rng.integers(65, 91, int(rng.integers(2000, 9000)), dtype=np.uint8)) for _ in range(6000)]      # ~33 MB: above the one-thread limit
    perm3 = rng.permutation(6000)
    b3, o3, k3 = [], [], []
    for a, b in ((0, 3100), (3100, 6000)):
        idx = np.sort(perm3[a:b])
        b3.append(np.frombuffer(b''.join(huge[i] for i in idx), dtype=np.uint8)); k3.append(idx.astype(np.int64))
        o3.append(np.concatenate([[0], np.cumsum([len(huge[i]) for i in idx])]).astype(np.int64))
    with tempfile.TemporaryFile() as f:
        f.write(b'head\n'); f.flush()
        w = VL.blob_write_parts(L, f.fileno(), b3, o3, k3, file_off=5, nthreads=4)
        w2 = VL.blob_write_parts(L, f.fileno(), b2, o2, k2, file_off=5 + w, nthreads=4)
        assert w == sum(len(t) for t in huge) and w2 == sum(len(t) for t in big)
        f.write(b'tail\n'); f.flush()
        f.seek(0)
        assert f.read() == b'head\n' + b''.join(huge) + b''.join(big) + b'tail\n'
    r_, w_ = os.pipe()                                             # a descriptor that cannot be mapped: the writev stream takes over
    try:
        assert VL.blob_write_parts(L, w_, [np.frombuffer(b'abc', dtype=np.uint8)], [np.array([0, 3], np.int64)], [np.array([0], np.int64)], file_off=0) == 3
        assert os.read(r_, 10) == b'abc'
    finally:
        os.close(r_); os.close(w_)


def _asm_sam_entries():
    """(entry, records as 9-tuples, query, contig dict, case meta) of tests/golden/sam_asm.json (tools/harness/gen_golden_sam_asm.py)"""
    import zlib
    import numpy as np
    import oracle_lib as O
    meta = json.load(open(os.path.join(GOLD, 'asm.json'))); arr = np.load(os.path.join(GOLD, 'asm.npz'))
    entries = json.load(open(os.path.join(GOLD, 'sam_asm.json')))
    cache = {}
    for e in entries:
        cid, ci = e['case'], e['contig']
        c = meta[cid]; g = c['contigs'][ci]
        contigs = {n: arr['%s_ref%d' % (cid, i)].tobytes().decode() for i, n in enumerate(c['names'])}
        query = arr['%s_c%d_seq' % (cid, ci)].tobytes().decode()
        if all(len(r) == 10 for r in g['records']):
            recs = [(g['name'], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[9]) for r in g['records']]
        else:                                    # long CIGARs are stored as (length, crc): take them from the oracle, which the record goldens pin
            if (cid, ci) not in cache:
                oi = O.Index.from_seqs(c['names'], [contigs[n] for n in c['names']], k=c['k'], w=c['w'])
                rc, orecs = O.align_asm(oi, query, O.params('asm'), *c['sizes'])
                assert [zlib.crc32(t[8].encode()) for t in orecs] == [r[8] for r in g['records']]
                cache[(cid, ci)] = [(g['name'], c['names'][t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]) for t in orecs]
            recs = cache[(cid, ci)]
        yield e, recs, query, contigs, c


def test_sam_lines_asm_match_reference():
    """-mode asm's emitter (sam.sam_lines(asm=True)) against the reference's iterator_get_bam_dict_str / _comments (mammap_asm.py:22757, :22942) on the
    records of the asm goldens (tests/golden/sam_asm.json, tools/harness/gen_golden_sam_asm.py)"""
    nlines = 0
    for e, recs, query, contigs, c in _asm_sam_entries():
        cid, ci, o = e['case'], e['contig'], e['opt']
        kw = dict(md=o['md'], shortcs=o['shortcs'], cigar2cg=o['cigar2cg'], markunbalancetra=o['markunbalancetra'], hardclip=o['H'], fakecigar=o['fakecigar'],
                  rg_id=o.get('rg'), comments=o['comments'].replace('\\t', '\t') if 'comments' in o else None, asm=True)
        if e['raised']:
            with pytest.raises(Exception):
                sam.sam_lines(recs, query, None, lambda cn, a, b: contigs[cn][a:b], **kw)
            continue
        lines = sam.sam_lines(recs, query, None, lambda cn, a, b: contigs[cn][a:b], **kw)
        assert [SC.head(x) for x in lines] == e['head'], (cid, ci, o)
        assert [SC.digest(x) for x in lines] == e['digest'], (cid, ci, o)
        nlines += len(lines)
    assert nlines >= 100


def test_native_sam_emitter_asm_matches_reference():
    """vm_sam_emit with asm_mode (what the driver's -mode asm path calls) against the same 118 reference lines"""
    import numpy as np
    import emu_lib
    from vacmap_amd import lib as VL
    ctx = emu_lib.context()
    idx = {}
    nlines = 0
    for e, recs, query, contigs, c in _asm_sam_entries():
        cid, ci, o = e['case'], e['contig'], e['opt']
        if cid not in idx:
            idx[cid] = VL.Index.from_seqs(ctx, c['names'], [contigs[n] for n in c['names']], k=c['k'], w=c['w'])
        raw = _raw_from_tuples(VL, recs, c['names'])
        opts = VL.SamOpts(int(o['md']), int(o['shortcs']), int(o['cigar2cg']), int(o['markunbalancetra']), int(o['H']), int(o['fakecigar']),
                          o['rg'].encode() if 'rg' in o else None, 1)
        nm = recs[0][0].encode()
        com = o['comments'].replace('\\t', '\t').encode() if 'comments' in o else None
        buf, off, nl, ns = VL.sam_emit(ctx.lib, idx[cid], opts, np.frombuffer(nm, np.uint8), [0, len(nm)], np.frombuffer(query.encode(), np.uint8), [0, len(query)], raw,
                                       comments=np.frombuffer(com, np.uint8) if com else None, com_off=[0, len(com)] if com else None, nthreads=2)
        lines = buf.tobytes().decode().split('\n')[:-1] if len(buf) else []
        if e['raised']:
            assert ns == 1 and not lines, (cid, ci, o)
            continue
        assert [SC.head(x) for x in lines] == e['head'], (cid, ci, o)
        assert [SC.digest(x) for x in lines] == e['digest'], (cid, ci, o)
        nlines += len(lines)
    assert nlines >= 100
