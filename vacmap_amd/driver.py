"""Command-line driver: reads in, SAM out — the counterpart of the reference's `vacmap` script (src/vacmap/vacmap:75-152 options,
:324-370 index + header, :430-517 read loop) around the MI355X library.

    python -m vacmap_amd.driver -ref ref.fa -read reads.fq -mode H -o out.sam [-t 8] [--eqx] [--MD] [--cs[=long]] [--H] ...

One process drives one GPU: the index is built (or loaded from `<ref>.w<w>_k<k>.vmx`, the reference's `.mmi` naming rule, vacmap:326)
and kept in HBM, reads are aligned in batches by `vm_align_batch`, and the records are turned into SAM lines (vacmap_amd/sam.py) by
`-t` worker processes, in input order. Like the reference's worker (:24116-24134) a read whose path or whose emission raises is
skipped, and a read without records produces no line. BAM output, `-mode asm` and read-name de-duplication are not provided.
"""
import argparse, gzip, os, sys
from multiprocessing import Pool

from . import sam


def read_fastx(path, want_comment=False):
    """yields (name, seq, qual or None, comment or None) from FASTA / FASTQ, plain or .gz (mp.fastx_read shape, vacmap:445)"""
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rt') as f:
        line = f.readline()
        while line:
            line = line.rstrip('\n')
            if not line:
                line = f.readline(); continue
            if line[0] == '>':
                head = line[1:]; seqs = []
                line = f.readline()
                while line and line[0] != '>':
                    seqs.append(line.strip()); line = f.readline()
                name, _, com = head.partition(' ')
                if not com and '\t' in name:
                    name, _, com = head.partition('\t')
                yield name, ''.join(seqs), None, (com if want_comment and com else None)
            elif line[0] == '@':
                head = line[1:]
                seq = f.readline().strip(); f.readline(); qual = f.readline().strip()
                name, _, com = head.partition(' ')
                if not com and '\t' in name:
                    name, _, com = head.partition('\t')
                yield name, seq, qual, (com if want_comment and com else None)
                line = f.readline()
            else:
                raise ValueError('not FASTA/FASTQ: %r' % line[:40])


_G = {}


def _emit_init(contigs, kw):
    _G['contigs'] = contigs; _G['kw'] = kw


def _emit(job):
    name, seq, qual, comments, recs = job
    try:
        c = _G['contigs']
        return sam.sam_lines(recs, seq, qual, lambda n, a, b: c[n][a:b], comments=comments, **_G['kw'])
    except Exception:          # the reference's worker skips reads whose emission raises (:24127-24134)
        return None


def build_parser():
    p = argparse.ArgumentParser(prog='vacmapx', description='MI355X-native VACmap path: seed, non-linear chain, extend; SAM output')
    p.add_argument('-ref', required=True); p.add_argument('-read', required=True, nargs='+', action='append')
    p.add_argument('-mode', required=True, choices=['H', 'L', 'S', 'R'])
    p.add_argument('-o', default='-'); p.add_argument('--force', action='store_true'); p.add_argument('--nowriteindex', action='store_true')
    p.add_argument('-t', type=int, default=8); p.add_argument('-k', type=str, default='15'); p.add_argument('-w', type=str, default='10')
    p.add_argument('-c', type=int, default=100); p.add_argument('-maxdivergence', type=float)
    p.add_argument('-globalpenalty', type=float); p.add_argument('-localpenalty', type=float)
    p.add_argument('-globalmaxdiff', type=int, default=50); p.add_argument('-localmaxdiff', type=int, default=30)
    p.add_argument('--eqx', action='store_true'); p.add_argument('--MD', action='store_true')
    p.add_argument('--cs', nargs='?', const='short', default=None); p.add_argument('--L', action='store_true')
    p.add_argument('--markunbalancetra', action='store_true'); p.add_argument('--nodiscard', action='store_true')
    p.add_argument('--copycomments', action='store_true'); p.add_argument('--H', action='store_true')
    p.add_argument('--fakecigar', action='store_true'); p.add_argument('--Q', action='store_true')
    p.add_argument('--rg-id', dest='rg_id'); p.add_argument('--rg-sm', dest='rg_sm')
    p.add_argument('--device', type=int, default=0); p.add_argument('--batch-reads', type=int, default=4096)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.o != '-' and not args.o.endswith('.sam'):
        sys.exit('vacmapx writes SAM: -o must be "-" or end in .sam')
    if args.o != '-' and os.path.exists(args.o) and not args.force:
        sys.exit('%s exists (use --force)' % args.o)
    from .lib import Context, Index, align_batch, load
    lib = load(); ctx = Context(args.device)
    k, w = int(args.k), int(args.w)
    vmx = '%s.w%d_k%d.vmx' % (args.ref, w, k)
    if os.path.exists(vmx):
        index = Index.load(ctx, vmx)
    else:
        index = Index.from_fasta(ctx, args.ref, k=k, w=w)
        if not args.nowriteindex:
            try:
                index.save(vmx)
            except Exception:
                pass
    prm = lib.params(args.mode)                     # mode defaults (vacmap:257-296), then the explicit options
    prm.check_num = args.c; prm.global_maxdiff = args.globalmaxdiff; prm.local_maxdiff = args.localmaxdiff
    prm.eqx = 1 if (args.eqx or args.MD or args.cs) else 0; prm.hardclip = 1 if args.H else 0
    if args.nodiscard: prm.nodiscard = 1
    if args.maxdivergence is not None: prm.maxdivergence = args.maxdivergence
    if args.globalpenalty is not None: prm.global_skipcost = args.globalpenalty
    if args.localpenalty is not None: prm.local_skipcost = args.localpenalty
    names = index.names
    contigs = {n: index.seq(i) for i, n in enumerate(names)}
    md = bool(args.MD or args.cs)
    mark = args.markunbalancetra or args.mode in ('H', 'L')          # mode defaults of vacmap:286-296
    kw = dict(md=md, shortcs=(args.cs != 'long'), cigar2cg=args.L, markunbalancetra=mark, hardclip=args.H, fakecigar=args.fakecigar, rg_id=args.rg_id)
    out = sys.stdout if args.o == '-' else open(args.o, 'w')
    rg = None
    if args.rg_id:
        rg = {'ID': args.rg_id}
        if args.rg_sm: rg['SM'] = args.rg_sm
    for ln in sam.header_lines([(n, len(contigs[n])) for n in names], ' '.join(sys.argv if argv is None else ['vacmapx'] + list(argv)), rg):
        out.write(ln + '\n')
    pool = Pool(max(1, args.t), initializer=_emit_init, initargs=(contigs, kw)) if args.t > 1 else None
    if pool is None:
        _emit_init(contigs, kw)
    n_reads = n_lines = n_skipped = 0

    def flush(batch):
        nonlocal n_lines, n_skipped
        if not batch:
            return
        status, recs, _ = align_batch(ctx, index, prm, [b[1].upper() for b in batch])
        per = {}
        for t in recs:
            per.setdefault(t[0], []).append(t)
        jobs = []
        for i, (name, seq, qual, com) in enumerate(batch):
            if status[i] != 0:
                n_skipped += 1
                continue
            rr = per.get(i)
            if not rr:
                continue
            jobs.append((name, seq.upper(), None if args.Q else qual, com,
                         [(name, names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]) for t in rr]))
        res = pool.map(_emit, jobs, chunksize=16) if pool else [_emit(j) for j in jobs]
        for lines in res:
            if lines is None:
                n_skipped += 1
                continue
            for ln in lines:
                out.write(ln + '\n'); n_lines += 1

    batch = []
    for group in args.read:
        for path in group:
            for rec in read_fastx(path, want_comment=args.copycomments):
                batch.append(rec); n_reads += 1
                if len(batch) >= args.batch_reads:
                    flush(batch); batch = []
    flush(batch)
    if pool:
        pool.close(); pool.join()
    if out is not sys.stdout:
        out.close()
    sys.stderr.write('vacmapx: %d reads, %d SAM lines, %d reads skipped\n' % (n_reads, n_lines, n_skipped))
    return 0


if __name__ == '__main__':
    sys.exit(main())
