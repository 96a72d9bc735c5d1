"""Command-line driver: reads in, SAM out — the counterpart of the reference's `vacmap` script (src/vacmap/vacmap:75-152 options,
:186-218 read group, :324-370 index + header, :430-517 read loop with name de-duplication and BAM input; writer
src/vacmap/output_functions.py:172-235) around the MI355X library.

    python -m vacmap_amd.driver -ref ref.fa -read reads.fq -mode H -o out.sam [-t 8] [--eqx] [--MD] [--cs[=long]] [--H] ...
    python -m torch.distributed.run --nproc-per-node N -m vacmap_amd.driver ...        (one rank per GPU)

One process drives one GPU. The index is built on the GPU (or loaded from `<ref>.w<w>_k<k>.vmx`, the reference's `.mmi` naming rule,
vacmap:326; a minimap2 `.mmi` of that name is read too) and kept in HBM. Reads flow through `vacmap_amd.pipeline`: windows of
`--window-batches` x `--batch-reads` reads, length-binned batches, several batches in flight; finished batches are turned into SAM
lines (vacmap_amd/sam.py) by `-t` worker processes while the next batches align, and every window is written in input order. Like the
reference's worker (:24116-24134) a read whose path or whose emission raises is skipped, and a read without records produces no line.
With N ranks, rank 0 builds the index and broadcasts it over RCCL (vacmap_amd/dist.py), batch i of a window goes to rank i mod N, and
rank 0 gathers and writes the lines. `-mode asm` is not provided.
"""
import argparse, gzip, os, shutil, struct, subprocess, sys, threading, queue
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


_BAM_NT16 = '=ACMGRSVTWYHKDBN'


def read_bam(path):
    """unaligned or aligned BAM -> (name, seq, qual or None, None), the fields the reference takes from pysam (vacmap:455-471): a
    reverse-strand record is turned back to the read's own orientation, qualities 0xff mean absent. BGZF is a series of gzip members."""
    with gzip.open(path, 'rb') as f:
        if f.read(4) != b'BAM\x01':
            raise ValueError('not a BAM file: %s' % path)
        l_text, = struct.unpack('<i', f.read(4)); f.read(l_text)
        n_ref, = struct.unpack('<i', f.read(4))
        for _ in range(n_ref):
            l_name, = struct.unpack('<i', f.read(4)); f.read(l_name + 4)
        while True:
            h = f.read(4)
            if len(h) < 4:
                return
            bs, = struct.unpack('<i', h)
            rec = f.read(bs)
            if len(rec) < bs:
                raise ValueError('truncated BAM record')
            l_rn, n_cig, flag, l_seq = rec[8], struct.unpack_from('<H', rec, 12)[0], struct.unpack_from('<H', rec, 14)[0], struct.unpack_from('<i', rec, 16)[0]
            p = 32
            name = rec[p:p + l_rn - 1].decode(); p += l_rn + 4 * n_cig
            packed = rec[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
            seq = ''.join(_BAM_NT16[b >> 4] + _BAM_NT16[b & 15] for b in packed)[:l_seq]
            q = rec[p:p + l_seq]
            qual = None if (l_seq == 0 or q[0] == 0xff) else bytes(x + 33 for x in q).decode('ascii')
            if l_seq == 0:
                continue                                   # "no sequence in BAM record" (vacmap:462)
            if flag & 16:
                seq = sam.revcomp(seq.upper()); qual = qual[::-1] if qual is not None else None
            yield name, seq, qual, None


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


RG_ARGS = (('rg-id', 'ID'), ('rg-sm', 'SM'), ('rg-lb', 'LB'), ('rg-pl', 'PL'), ('rg-ds', 'DS'), ('rg-dt', 'DT'), ('rg-pu', 'PU'), ('rg-pi', 'PI'),
           ('rg-pg', 'PG'), ('rg-cn', 'CN'), ('rg-fo', 'FO'), ('rg-ks', 'KS'), ('rg-pm', 'PM'), ('rg-bc', 'BC'))       # vacmap:45-60


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
    for a, _ in RG_ARGS:
        p.add_argument('--' + a, dest=a.replace('-', '_'))
    p.add_argument('--device', type=int, default=None); p.add_argument('--batch-reads', type=int, default=4096)
    p.add_argument('--window-batches', type=int, default=8); p.add_argument('--inflight', type=int, default=3)
    return p


def _open_output(path):
    """'-' / .sam: text; .bam / .sorted.bam: a `samtools view -b` / `samtools sort --write-index` pipe (output_functions.py:200-208)"""
    if path == '-':
        return sys.stdout, None
    if path.endswith('.sam'):
        return open(path, 'w'), None
    if not shutil.which('samtools'):
        sys.exit('writing %s needs the samtools binary on PATH (the reference pipes SAM text into it too); write .sam instead' % path)
    cmd = ['samtools', 'sort', '-@', '8', '--write-index', '-o', path, '-'] if path.endswith('sorted.bam') else ['samtools', 'view', '-b', '-@', '8', '-o', path, '-']
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, encoding='utf-8', bufsize=64 * 1024)
    return proc.stdin, proc


def main(argv=None, comm=None):
    """comm: an initialised torch.distributed module (tests); under torchrun (WORLD_SIZE > 1) the process group is created here"""
    args, _unknown = build_parser().parse_known_args(argv)          # unknown flags are ignored like the reference's parse_known_args (vacmap:152)
    if args.o != '-' and not (args.o.endswith('.sam') or args.o.endswith('.bam')):
        sys.exit("Output path must end with .sam, .bam, .sorted.bam, or be '-' for stdout.")
    world, rank, local_rank = 1, 0, 0
    own_group = False
    if comm is None and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch, torch.distributed as comm
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        comm.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        own_group = True
    if comm is not None:
        world, rank = comm.get_world_size(), comm.get_rank()
    if rank == 0 and args.o != '-' and os.path.exists(args.o) and not args.force:
        sys.exit('%s exists (use --force)' % args.o)
    from .lib import Context, Index, load
    from . import pipeline
    lib = load()
    device = args.device if args.device is not None else local_rank
    ctx = Context(device)
    k, w = int(args.k), int(args.w)
    index = None
    if rank == 0:
        from .indexfile import find_index
        index = find_index(ctx, args.ref, k, w, write=not args.nowriteindex)
    if world > 1:
        import torch
        from .dist import broadcast_index
        dev = torch.device('cuda', device) if torch.cuda.is_available() else torch.device('cpu')
        index, _ = broadcast_index(ctx, index, src=0, device=dev)
    prm = lib.params(args.mode)                     # mode defaults (vacmap:257-296), then the explicit options
    prm.check_num = args.c; prm.global_maxdiff = args.globalmaxdiff; prm.local_maxdiff = args.localmaxdiff
    prm.eqx = 1 if args.eqx else 0; prm.hardclip = 1 if args.H else 0
    if args.nodiscard: prm.nodiscard = 1
    if args.maxdivergence is not None: prm.maxdivergence = args.maxdivergence
    if args.globalpenalty is not None: prm.global_skipcost = args.globalpenalty
    if args.localpenalty is not None: prm.local_skipcost = args.localpenalty
    names = index.names
    contigs = {n: index.seq(i).upper() for i, n in enumerate(names)}
    # read group: always present, like the reference (vacmap:186-218) — {'ID': '1', 'SM': 'sample'} unless --rg-* options are given
    rg = {}
    for a, tag in RG_ARGS:
        v = getattr(args, a.replace('-', '_'))
        if v is not None:
            rg[tag] = str(v)
    if rg and 'ID' not in rg:
        sys.exit('The --rg-id option is required when any other --rg-* option is supplied.')
    if not rg:
        rg = {'ID': '1', 'SM': 'sample'}
    mark = args.markunbalancetra or args.mode in ('H', 'L')          # mode defaults of vacmap:286-296
    kw = dict(md=bool(args.MD), shortcs=(args.cs != 'long'), cigar2cg=args.L, markunbalancetra=mark, hardclip=args.H, fakecigar=args.fakecigar, rg_id=rg['ID'])
    out, proc = (None, None)
    if rank == 0:
        out, proc = _open_output(args.o)
        for ln in sam.header_lines([(n, len(contigs[n])) for n in names], ' '.join(sys.argv if argv is None else ['vacmapx'] + list(argv)), rg):
            out.write(ln + '\n')
    pool = Pool(max(1, args.t), initializer=_emit_init, initargs=(contigs, kw)) if args.t > 1 else None
    if pool is None:
        _emit_init(contigs, kw)
    pipe = pipeline.Pipeline(index, prm, device=device, inflight=args.inflight, first_ctx=ctx)
    counts = {'reads': 0, 'lines': 0, 'skipped': 0}
    win_reads = max(1, args.batch_reads * args.window_batches)

    def windows():
        """input records in arrival order, upper-cased, de-duplicated by name (vacmap:457,475,487), cut into windows"""
        seen = set(); cur = []
        for group in args.read:
            for path in group:
                it = read_bam(path) if path.endswith('.bam') else read_fastx(path, want_comment=args.copycomments)
                for name, seq, qual, com in it:
                    if name in seen:
                        continue
                    seen.add(name)
                    cur.append((name, seq.upper(), None if args.Q else qual, com))
                    if len(cur) >= win_reads:
                        yield cur; cur = []
        if cur:
            yield cur

    wq = queue.Queue(maxsize=2)

    def reader():
        try:
            for wnd in windows():
                wq.put(wnd)
            wq.put(None)
        except BaseException as e:
            wq.put(e)

    threading.Thread(target=reader, daemon=True).start()
    outq = queue.Queue(maxsize=2)
    werr = []

    def writer():
        """collects a window's emission results, gathers the ranks' lines on rank 0 and writes them in input order"""
        try:
            while True:
                item = outq.get()
                if item is None:
                    return
                pending = item                               # [(read index in window, AsyncResult or list)]
                mine = {}
                for ids, res in pending:
                    lines = res.get() if hasattr(res, 'get') else res
                    for ridx, ls in zip(ids, lines):
                        if ls is None:
                            counts['skipped'] += 1
                        else:
                            mine[ridx] = ls
                if world > 1:
                    from .dist import gather_lines
                    parts = gather_lines(mine, dst=0)
                    if rank != 0:
                        continue
                    mine = {}
                    for d in parts:
                        mine.update(d)
                for ridx in sorted(mine):
                    for ln in mine[ridx]:
                        out.write(ln + '\n'); counts['lines'] += 1
        except BaseException as e:
            werr.append(e)

    wt = threading.Thread(target=writer)
    wt.start()
    import numpy as np
    while True:
        wnd = wq.get()
        if wnd is None:
            break
        if isinstance(wnd, BaseException):
            outq.put(None); wt.join()
            raise wnd
        counts['reads'] += len(wnd)
        plan = pipeline.plan_batches(np.fromiter((len(r[1]) for r in wnd), dtype=np.int64, count=len(wnd)), args.batch_reads, args.window_batches)
        plan = [plan[i] for i in range(rank, len(plan), world)]        # static sharding: batch i -> rank i mod N
        pending = []

        def on_result(i, res, plan=plan, wnd=wnd, pending=pending):
            status, recs, _ = res
            per = {}
            for t in recs:
                per.setdefault(t[0], []).append(t)
            jobs, ids = [], []
            for j, ridx in enumerate(plan[i]):
                if status[j] != 0:
                    counts['skipped'] += 1
                    continue
                rr = per.get(j)
                if not rr:
                    continue
                name, seq, qual, com = wnd[int(ridx)]
                jobs.append((name, seq, qual, com, [(name, names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]) for t in rr]))
                ids.append(int(ridx))
            pending.append((ids, pool.map_async(_emit, jobs, chunksize=16) if pool else [_emit(jb) for jb in jobs]))

        pipe.run_host([[wnd[int(r)][1] for r in b] for b in plan], on_result=on_result)
        outq.put(pending)
        if werr:
            break
    outq.put(None); wt.join()
    if werr:
        raise werr[0]
    if pool:
        pool.close(); pool.join()
    pipe.close()
    if rank == 0:
        if proc is not None:
            out.close()
            rc = proc.wait()
            if rc != 0:
                sys.stderr.write('Error: samtools exited with code %d\n' % rc)
        elif out is not sys.stdout:
            out.close()
        sys.stderr.write('vacmapx: %d reads, %d SAM lines, %d reads skipped\n' % (counts['reads'], counts['lines'], counts['skipped']))
    if own_group:
        comm.barrier(); comm.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
