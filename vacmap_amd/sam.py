"""SAM emission for the records of vm_align_batch (the consumer of the path).

Mirrors the reference's `get_bam_dict_str` (/root/reference/src/vacmap/mammap_clrnano.py:20841-21020) and its helpers —
`reassign_mapq` :11661, `mergecigar_` :4773, `mergecigar_md_` :19113, `get_MD_CSshort` :19012, `get_MD_CSlong` :19062,
`P_alignmentstring` :5391, `nm_from_cigar` (output_functions.py:300) — so that the same records give the same SAM lines:
longest alignment first and primary, the others supplementary (FLAG 2048), SA tag over all other records, NM from the CIGAR, optional
MD / cs (need --eqx CIGARs), hard clipping, approximate SA CIGARs, CG tag for > 65535 operators. Pure Python; no GPU involved.
"""
import re

_COMP = bytes.maketrans(b'ACGTNacgtn', b'TGCANtgcan')


def revcomp(seq):
    """Bio.Seq(query).reverse_complement() for the alphabets the driver sees (other letters are kept as they are)"""
    return seq.encode().translate(_COMP)[::-1].decode()


def merge_cigar(cigar):
    """:4773 — merge consecutive operators of the same kind; returns the flat [count, op, count, op, ...] list of strings"""
    ops = []
    num = 0
    pre_op, pre_num = '0', 0
    for ch in cigar:
        if '0' <= ch <= '9':
            num = num * 10 + ord(ch) - 48
        else:
            if pre_op == ch:
                pre_num += num
                ops[-2] = str(pre_num)
            else:
                pre_num = num
                ops.append(str(num)); ops.append(ch)
                pre_op = ch
            num = 0
    return ops


def merge_cigar_nm(cigar):
    """mergecigar_nm_ of the -mode asm fork (mammap_asm.py:23125-23155): merge_cigar plus the edit distance the fork reports as NM — the counts
    of X, D and I operators, where a run that continues the operator before it is merged but NOT counted (:23142-23145), as in the reference"""
    ops = []
    num = 0
    pre_op, pre_num = '0', 0
    nm = 0
    for ch in cigar:
        if '0' <= ch <= '9':
            num = num * 10 + ord(ch) - 48
        else:
            if pre_op == ch:
                pre_num += num
                ops[-2] = str(pre_num)
            else:
                pre_num = num
                ops.append(str(num)); ops.append(ch)
                if ch in 'XDI':
                    nm += num
                pre_op = ch
            num = 0
    return ops, nm


def nm_from_cigar(cigar, query, ref):
    """NM of a SAM record from its CIGAR (the reference's `nm_from_cigar`, output_functions.py:300, pinned by its nine known-answer
    tests): inserted + deleted + X-run bases, plus the mismatching columns of M runs (compared case-insensitively). S / I consume the
    query, D / N the reference, = / X / M both; H and P consume nothing. An M run that leaves either sequence raises IndexError, like
    the reference's per-base indexing does."""
    nm = qpos = rpos = run = 0
    for ch in cigar:
        d = ord(ch) - 48
        if 0 <= d <= 9:
            run = run * 10 + d
            continue
        n, run = run, 0
        if ch == 'M':
            qs, rs = query[qpos:qpos + n], ref[rpos:rpos + n]
            if len(qs) < n or len(rs) < n:
                raise IndexError('M run beyond the sequence')
            if qs != rs:
                nm += sum(1 for x, y in zip(qs.upper(), rs.upper()) if x != y)
            qpos += n; rpos += n
        elif ch == '=':
            qpos += n; rpos += n
        elif ch == 'X':
            nm += n; qpos += n; rpos += n
        elif ch == 'I':
            nm += n; qpos += n
        elif ch == 'D':
            nm += n; rpos += n
        elif ch == 'S':
            qpos += n
        elif ch == 'N':
            rpos += n
    return nm


def md_cs(ops, target, query, short=True):
    """:19012 (short) / :19062 (long): MD and cs strings of a merged =/X/I/D CIGAR; any other operator but S/H gives ('', '')"""
    md, cs = [], []
    refloc = readloc = 0
    preop = ''
    equal = 0
    for i in range(1, len(ops), 2):
        n, op = int(ops[i - 1]), ops[i]
        if op == 'X':
            if equal > 0:
                md.append(str(equal))
            elif preop == 'D':
                md.append('0')
            md.append(target[refloc])
            cs.append('*' + (target[refloc] + query[readloc]).lower())
            for j in range(1, n):
                md.append('0' + target[refloc + j])
                cs.append('*' + (target[refloc + j] + query[readloc + j]).lower())
            refloc += n; readloc += n
            equal = 0
        elif op == '=':
            if short:
                cs.append(':' + ops[i - 1])
            else:
                cs.append('=' + target[refloc:refloc + n].upper())
            refloc += n; readloc += n
            equal += n
        elif op == 'D':
            if equal > 0:
                md.append(str(equal))
            elif preop == 'X':
                md.append('0')
            md.append('^' + target[refloc:refloc + n])
            cs.append('-' + target[refloc:refloc + n].lower())
            refloc += n
            equal = 0
        elif op == 'I':
            cs.append('+' + query[readloc:readloc + n].lower())
            readloc += n
            continue
        elif op in ('S', 'H'):
            continue
        else:
            return '', ''
        preop = op
    if equal > 0:
        md.append(str(equal))
    return ''.join(md), ''.join(cs)


def reassign_mapq(records):
    """:11661 — records that do not continue the previous kept one on the same contig (unbalanced translocation pieces) get MAPQ 0"""
    keep = [0]
    n = len(records)
    while keep[-1] < n - 1:
        i = keep[-1]
        b = records[i]
        hit = False
        t = i
        while t + 1 < n:
            t += 1
            x = records[t]
            if x[1] != b[1]:
                continue
            if x[2] == '+':
                refgap = x[5] - b[6]
            else:
                refgap = b[5] - x[6]
            if abs(refgap) > 100000:
                continue
            if refgap < 10:
                keep.append(t); hit = True
                break
        if not hit:
            keep.append(i + 1)
    out = []
    for i, r in enumerate(records):
        r = list(r)
        if i not in keep:
            r[7] = 0
        out.append(r)
    return out


_FIELDS = ('QNAME', 'FLAG', 'RNAME', 'POS', 'MAPQ', 'CIGAR', 'RNEXT', 'PNEXT', 'TLEN', 'SEQ', 'QUAL')


def format_line(d, comments=None):
    """:5391 — the eleven mandatory columns (defaults * 4 * 0 255 * * 0 0 * *) followed by the tags in insertion order; with
    `comments` (:20686, --copycomments): tab-separated `XX:T:value` fields of the FASTA/Q comment whose tag is not present yet"""
    cols = ['*', '4', '*', '0', '255', '*', '*', '0', '0', '*', '*']
    tags = set(_FIELDS) | {'SA', 'NM', 'MD', 'cs'}
    for k, v in d.items():
        if k in _FIELDS:
            cols[_FIELDS.index(k)] = v
        else:
            code = 'i' if type(v) is int else ('f' if type(v) is float else 'Z')
            cols.append('%s:%s:%s' % (k, code, v))
            tags.add(k)
    if isinstance(comments, str):
        for one in comments.split('\t'):
            info = one.split(':')
            if len(info) == 3 and len(info[0]) == 2 and info[0] not in tags and info[1] in ('A', 'i', 'f', 'Z', 'H', 'B'):
                cols.append(one)
                tags.add(info[0])
    return '\t'.join(cols)


def _fake_cigar(r, qlen, clip):
    top = '%d%s' % (r[3], clip) if r[3] > 0 else ''
    tail = '%d%s' % (qlen - r[4], clip) if qlen - r[4] > 0 else ''
    diff = r[4] - r[3] - r[6] + r[5]
    if diff > 0:
        body = '%dM%dI' % (r[6] - r[5], diff)
    elif diff < 0:
        body = '%dM%dD' % (r[4] - r[3], -diff)
    else:
        body = '%dM' % (r[4] - r[3])
    return top + body + tail


def sam_lines(records, query, qual, refseq, md=False, shortcs=True, cigar2cg=False, markunbalancetra=True, hardclip=False,
              fakecigar=False, rg_id=None, comments=None, asm=False):
    """SAM lines of one read (:20841-21020; with `comments`: the `_comments` twin :21022, which appends the FASTQ comment).

    records: 9-tuples (qname, contig, strand, q_st, q_en, r_st, r_en, mapq, cigar) in the path's order; q_st/q_en index the read for
    '+' and its reverse complement for '-'. refseq(contig, start, end) -> reference bases. Raises what the reference raises (the
    caller skips the read, :24127-24134).
    asm: the emitter of -mode asm, `iterator_get_bam_dict_str` / `_comments` (mammap_asm.py:22757-22940, :22942): NM comes from the CIGAR's X / D / I
    counts (`mergecigar_nm_` :23125), MAPQ is written as 60 (or 1 for 0) in the MAPQ column and in SA, and the second-longest record is the
    primary one when the longest has MAPQ 1 and it has not (:22847-22850)."""
    recs = reassign_mapq(records) if markunbalancetra else [list(r) for r in records]
    rc_query = revcomp(query)
    recs.sort(key=lambda x: x[4] - x[3])          # stable, then reversed: longest first, later ones first among equals
    recs = recs[::-1]
    nms, mds, css, ncig, fakes = [], [], [], [], []
    clip = 'H' if hardclip else 'S'
    for r in recs:
        if not md:
            ops, nm_asm = merge_cigar_nm(r[8]) if asm else (merge_cigar(r[8]), None)
            r[8] = ''.join(ops)
            nms.append(nm_asm if asm else nm_from_cigar(r[8], query if r[2] == '+' else rc_query, refseq(r[1], r[5], r[6])))
            ncig.append(len(ops))
        else:
            q = (query if r[2] == '+' else rc_query)[r[3]:r[4]]
            t = refseq(r[1], r[5], r[6])
            ops, nm_asm = merge_cigar_nm(r[8]) if asm else (merge_cigar(r[8]), None)
            m_, c_ = md_cs(ops, t, q, shortcs)
            r[8] = ''.join(ops)
            nms.append(nm_asm if asm else nm_from_cigar(r[8], q, t))
            mds.append(m_); css.append(c_); ncig.append(len(ops))
        if fakecigar:
            fakes.append(_fake_cigar(r, len(query), clip))
    has_qual = qual is not None and len(qual) == len(query)
    rc_qual = qual[::-1] if has_qual else None
    lines = []
    primary = 1 if asm and len(recs) > 1 and recs[0][7] == 1 and recs[1][7] != 1 else 0
    mq_out = (lambda v: 60 if v != 0 else 1) if asm else (lambda v: v)
    for i, r in enumerate(recs):
        d = {}
        if rg_id is not None:
            d['RG'] = rg_id
        d['QNAME'] = r[0]
        d['RNAME'] = r[1]
        base = 0 if i == primary else 2048
        d['FLAG'] = str(base if r[2] == '+' else 16 + base)
        d['POS'] = str(r[5] + 1)
        if ncig[i] > 65535 and cigar2cg:
            d['CG'] = r[8]
        else:
            d['CIGAR'] = r[8]
        if len(recs) > 1:
            sa = []
            for t, x in enumerate(recs):
                if t == i:
                    continue
                sa.append('%s,%d,%s,%s,%d,%d;' % (x[1], x[5] + 1, x[2], fakes[t] if fakecigar else x[8], mq_out(x[7]), nms[t]))
            d['SA'] = ''.join(sa)
        d['MAPQ'] = str(mq_out(r[7]))
        seq, ql = (query, qual) if r[2] == '+' else (rc_query, rc_qual)
        if not hardclip:
            d['SEQ'] = seq
            if has_qual:
                d['QUAL'] = ql
        else:
            d['SEQ'] = seq[r[3]:r[4]]
            if has_qual:
                d['QUAL'] = ql[r[3]:r[4]]
        d['NM'] = nms[i]
        if md:
            d['MD'] = mds[i]
            d['cs'] = css[i]
        lines.append(format_line(d, comments))
    return lines


def header_lines(contigs, command_line, rg=None):
    """@HD / @SQ / @RG / @PG of the driver (src/vacmap/vacmap:349-370). contigs: [(name, length)]; rg: dict of RG fields incl. ID"""
    out = ['@HD\tVN:1.0']
    for name, ln in contigs:
        out.append('@SQ\tSN:%s\tLN:%d' % (name, ln))
    if rg:
        out.append('@RG\t' + '\t'.join('%s:%s' % (k, v) for k, v in rg.items()))
    out.append('@PG\tID:VACmap\tPN:VACmap\tVN:1.0.2\tCL:%s' % command_line)
    return out
