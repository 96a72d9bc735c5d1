"""Inputs of the SAM-emitter goldens (tests/golden/sam.json), re-derived from the path goldens exactly as tools/harness/gen_golden_sam.py
derived them for the reference's get_bam_dict_str."""
import hashlib, re, zlib

_COMP = bytes.maketrans(b'ACGTN', b'TGCAN')


def eqx(cigar, query, ref):
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
                    if run_op:
                        out.append('%d%s' % (run, run_op))
                    run_op, run = o, 1
            out.append('%d%s' % (run, run_op))
            q += n; r += n
        else:
            out.append('%d%s' % (n, op))
            if op in 'IS':
                q += n
            elif op in 'DN':
                r += n
            elif op in '=X':
                q += n; r += n
    return ''.join(out)


def inputs(entry, meta, arrays):
    cid, ri, o = entry['case'], entry['read'], entry['opt']
    c = meta[cid]; r = c['reads'][ri]
    contigs = {n: arrays['%s_contig%d' % (cid, i)].tobytes().decode() for i, n in enumerate(c['names'])}
    query = arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode()
    rcq = query.encode().translate(_COMP)[::-1].decode()
    qual = ''.join(chr(33 + (7 * i) % 40) for i in range(len(query))) if entry['qual'] else None
    recs = []
    for t in r['v6_records']:
        rec = [r['name'], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]]
        if o.get('eqx'):
            rec[8] = eqx(rec[8], query if rec[2] == '+' else rcq, contigs[rec[1]][rec[5]:rec[6]])
        recs.append(tuple(rec))
    return recs, query, qual, contigs


def digest(line):
    return hashlib.sha256(line.encode()).hexdigest()


def head(line):
    return '\t'.join(c if len(c) <= 60 else '%s..[%d:%08x]' % (c[:24], len(c), zlib.crc32(c.encode())) for c in line.split('\t'))
