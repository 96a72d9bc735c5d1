"""Command-line driver: reads in, SAM out — the counterpart of the reference's `vacmap` script (src/vacmap/vacmap:75-152 options,
:186-218 read group, :324-370 index + header, :430-517 read loop with name de-duplication and BAM input; writer
src/vacmap/output_functions.py:172-235) around the MI355X library.

    python -m vacmap_amd.driver -ref ref.fa -read reads.fq -mode H -o out.sam [-t 8] [--eqx] [--MD] [--cs[=long]] [--H] ...
    python -m torch.distributed.run --nproc-per-node N -m vacmap_amd.driver ...        (one rank per GPU)

One process drives one GPU. The index is built on the GPU (or loaded from `<ref>.w<w>_k<k>.vmx`, the reference's `.mmi` naming rule,
vacmap:326; a minimap2 `.mmi` of that name is read too) and kept in HBM. Reads never become Python objects: the library's FASTX
reader (vm_fastx_read) fills blobs, `vacmap_amd.pipeline` cuts windows of `--window-batches` x `--batch-reads` reads into length-binned
batches, several batches are in flight, each worker thread aligns its batch on the GPU (vm_align_batch) and turns the records into SAM
text with `-t` host threads (vm_sam_emit, the C++ twin of vacmap_amd/sam.py) while the other batches align; every window is written
in input order. Like the reference's worker (:24116-24134) a read whose path or whose emission raises is skipped, and a read without
records produces no line.
With N ranks, rank 0 builds the index and broadcasts it over RCCL (vacmap_amd/dist.py), batch i of a window goes to rank i mod N, and
rank 0 gathers and writes the lines. `-mode asm` (assembly contigs; contig c -> rank c mod N): the contigs go through vm_align_batch with VM_MODE_ASM in
groups, in input order, and their lines come from the native emitter in its asm form (vm_sam_opts.asm_mode = iterator_get_bam_dict_str,
mammap_asm.py:22757; vacmap_amd/sam.py holds the same emitter in Python); `-workdir` is accepted and created like the reference's, but nothing is spilled into it.
"""
import argparse, gzip, os, shutil, struct, subprocess, sys, threading, time, queue

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
                name, com = _split_header(head)
                yield name, ''.join(seqs), None, (com if want_comment and com else None)
            elif line[0] == '@':
                head = line[1:]
                seq = f.readline().strip(); f.readline(); qual = f.readline().strip()
                name, com = _split_header(head)
                yield name, seq, qual, (com if want_comment and com else None)
                line = f.readline()
            else:
                raise ValueError('not FASTA/FASTQ: %r' % line[:40])


def _split_header(head):
    """name | comment at the first blank or tab (kseq's rule)"""
    for i, ch in enumerate(head):
        if ch == ' ' or ch == '\t':
            return head[:i], head[i + 1:]
    return head, ''


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


def _bam_chunks(path, n_max):
    """BAM records as the blob chunks the FASTX reader yields (names, upper-cased sequences, qualities, no comments)"""
    import numpy as np
    cur = []

    def pack(rows):
        out = {}
        for key, col in (('names', 0), ('seqs', 1), ('quals', 2)):
            bs = [(r[col] or '').encode() for r in rows]
            out[key] = np.frombuffer(b''.join(bs), dtype=np.uint8)
            out[key + '_off'] = np.concatenate([[0], np.cumsum([len(b) for b in bs])]).astype(np.int64)
        out['comments'] = np.zeros(0, np.uint8); out['comments_off'] = np.zeros(len(rows) + 1, np.int64)
        return out
    for name, seq, qual, _ in read_bam(path):
        cur.append((name, seq.upper(), qual))
        if len(cur) >= n_max:
            yield pack(cur); cur = []
    if cur:
        yield pack(cur)


def _is_plain_fastx(path):
    """an uncompressed FASTA / FASTQ file (byte ranges of it can be parsed independently)"""
    if path.endswith('.bam') or not os.path.isfile(path):
        return False
    with open(path, 'rb') as f:
        return f.read(2) != b'\x1f\x8b'


_HASH_POW = None


def name_hashes(blob, off):
    """64-bit hashes of the names in a blob (uint8 array + offsets), the same in every process (Python's hash() of bytes is salted per
    process): a position-weighted polynomial modulo 2^64 with a splitmix64 finish, in NumPy over the whole blob. Range mode compares the
    ranks' sets at the end of the run to give the reference's GLOBAL de-duplication by name (vacmap:457-487)."""
    global _HASH_POW
    import numpy as np
    off = np.asarray(off, dtype=np.int64)
    n = len(off) - 1
    if n <= 0:
        return np.zeros(0, np.uint64)
    lens = np.diff(off)
    mx = int(lens.max()) if n else 0
    if _HASH_POW is None or len(_HASH_POW) < mx + 1:
        pw = np.empty(max(mx + 1, 256), np.uint64); pw[0] = 1
        with np.errstate(over='ignore'):
            for i in range(1, len(pw)):
                pw[i] = pw[i - 1] * np.uint64(0x9E3779B97F4A7C15)
        _HASH_POW = pw
    b = np.asarray(blob, dtype=np.uint8)[off[0]:off[-1]]
    pos = np.arange(len(b), dtype=np.int64) - np.repeat(off[:-1] - off[0], lens)
    with np.errstate(over='ignore'):
        contrib = (b.astype(np.uint64) + np.uint64(1)) * _HASH_POW[pos]
        h = np.zeros(n, np.uint64)
        nz = lens > 0
        if len(contrib):
            st = (off[:-1] - off[0])[nz]
            h[nz] = np.add.reduceat(contrib, st)
        h = h + lens.astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93)
        h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9); h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB); h ^= h >> np.uint64(31)
    return h


def cross_rank_duplicates(hashes_per_rank, files_per_rank):
    """{rank: name hashes to leave out of that rank's part}: of the occurrences of a name over all ranks' ranges, every one but the FIRST IN INPUT ORDER
    (lowest input-file number, then lowest rank = lowest byte range of that file); a rank holds a name at most once (its own windows de-duplicate)."""
    import numpy as np
    hs = np.concatenate([np.asarray(h, dtype=np.uint64) for h in hashes_per_rank]) if hashes_per_rank else np.zeros(0, np.uint64)
    fl = np.concatenate([np.asarray(f, dtype=np.int32) for f in files_per_rank]) if files_per_rank else np.zeros(0, np.int32)
    rk = np.concatenate([np.full(len(h), r, np.int32) for r, h in enumerate(hashes_per_rank)]) if hashes_per_rank else np.zeros(0, np.int32)
    order = np.lexsort((rk, fl, hs)); hs, rk = hs[order], rk[order]
    dup = np.zeros(len(hs), bool)
    if len(hs) > 1:
        dup[1:] = hs[1:] == hs[:-1]
    return {int(r): hs[dup & (rk == r)] for r in np.unique(rk[dup])}


def _span_hashes(buf, starts, lens):
    """name_hashes for names that lie at arbitrary places of one buffer (uint8 array): the spans are gathered into one blob first"""
    import numpy as np
    starts = np.asarray(starts, dtype=np.int64); lens = np.asarray(lens, dtype=np.int64)
    off = np.zeros(len(lens) + 1, np.int64); np.cumsum(lens, out=off[1:])
    idx = np.repeat(starts - off[:-1], lens) + np.arange(int(off[-1]), dtype=np.int64)
    return name_hashes(buf[idx], off)


def _filter_part(f, o, bad, block=64 << 20):
    """copy SAM part `f` to `o` without the alignment lines whose read name hashes into `bad` (sorted uint64 array); the names of a block of
    lines are hashed together in NumPy (one name_hashes call per LINE took minutes on a large part). Returns (hashes left out, lines left out)."""
    import numpy as np
    gone, gone_lines, tail = set(), 0, b''
    while True:
        blk = f.read(block)
        data = tail + blk
        if not data:
            break
        if blk:
            cut = data.rfind(b'\n') + 1
            if cut == 0:                                 # no complete line yet
                tail = data
                continue
            data, tail = data[:cut], data[cut:]
        else:
            tail = b''
            if not data.endswith(b'\n'):
                data += b'\n'
        a = np.frombuffer(data, dtype=np.uint8)
        ends = np.flatnonzero(a == 10)
        starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.int64)
        tabs = np.flatnonzero(a == 9)
        ti = np.searchsorted(tabs, starts)
        nm_end = np.where(ti < len(tabs), tabs[np.minimum(ti, max(len(tabs) - 1, 0))] if len(tabs) else ends, ends)
        nm_end = np.minimum(nm_end, ends)
        body = a[starts] != 64                           # '@': header line
        h = np.zeros(len(starts), np.uint64)
        if body.any():
            h[body] = _span_hashes(a, starts[body], (nm_end - starts)[body])
        out_ = body & np.isin(h, bad)
        if out_.any():
            gone.update(int(x) for x in np.unique(h[out_])); gone_lines += int(out_.sum())
            keep = np.flatnonzero(~out_)
            # runs of kept lines are written as slices
            if len(keep):
                brk = np.flatnonzero(np.diff(keep) != 1)
                rs = np.concatenate([[0], brk + 1]); re_ = np.concatenate([brk, [len(keep) - 1]])
                for x, y in zip(rs, re_):
                    o.write(data[int(starts[keep[x]]):int(ends[keep[y]]) + 1])
        else:
            o.write(data)
        if not blk:
            break
    return gone, gone_lines


def _concat_parts(dst, parts, drop=None):
    """the ranks' SAM parts, in rank order, into one file (in-kernel copies; the parts are removed). drop[r]: name hashes whose lines are
    left out of part r (a read name that occurs EARLIER IN THE INPUT — an earlier file, or a lower byte range of the same file — inside another
    rank's range: the reference keeps the first occurrence only, vacmap:457-487) — that part is then copied block by block through a filter;
    returns the (reads, lines) left out."""
    import numpy as np
    gone_reads, gone_lines = set(), 0
    with open(dst, 'wb', buffering=0) as o:             # unbuffered: sendfile on the descriptor and write() through the object must not interleave out of order
        for r, pth in enumerate(parts):
            with open(pth, 'rb') as f:
                if drop and r in drop and len(drop[r]):
                    g, gl = _filter_part(f, o, np.sort(np.asarray(drop[r], dtype=np.uint64)))
                    gone_reads |= set((r, x) for x in g); gone_lines += gl
                else:
                    size = os.fstat(f.fileno()).st_size; off = 0
                    while off < size:
                        try:
                            n = os.sendfile(o.fileno(), f.fileno(), off, min(size - off, 1 << 30))
                        except OSError:
                            n = 0
                        if n <= 0:                      # (a file system without sendfile between regular files: plain copy)
                            f.seek(off)
                            for blk in iter(lambda: f.read(16 << 20), b''):
                                o.write(blk)
                            break
                        off += n
            os.remove(pth)
    return len(gone_reads), gone_lines


RG_ARGS = (('rg-id', 'ID'), ('rg-sm', 'SM'), ('rg-lb', 'LB'), ('rg-pl', 'PL'), ('rg-ds', 'DS'), ('rg-dt', 'DT'), ('rg-pu', 'PU'), ('rg-pi', 'PI'),
           ('rg-pg', 'PG'), ('rg-cn', 'CN'), ('rg-fo', 'FO'), ('rg-ks', 'KS'), ('rg-pm', 'PM'), ('rg-bc', 'BC'))       # vacmap:45-60


def build_parser():
    p = argparse.ArgumentParser(prog='vacmapx', description='MI355X-native VACmap path: seed, non-linear chain, extend; SAM output')
    p.add_argument('-ref', required=True); p.add_argument('-read', required=True, nargs='+', action='append')
    p.add_argument('-mode', required=True, choices=['H', 'L', 'S', 'R', 'asm']); p.add_argument('-workdir')
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
    p.add_argument('--window-batches', type=int, default=4); p.add_argument('--inflight', type=int, default=0, help='batches in flight on the GPU; 0 (default): five, then as many more (up to eight) as the HBM has room for after the sizing run')
    # N ranks (torchrun): 'range' = every rank parses its own byte range of each plain FASTA / FASTQ input and writes its own part of the SAM file
    # (<out>.partNNN, concatenated by rank 0 at the end unless --parts); 'batch' = every rank parses everything and keeps every N-th batch, rank 0
    # gathers the text (compressed / BAM input, stdout or BAM output); 'auto' picks 'range' whenever input and output allow it
    p.add_argument('--shard', choices=['auto', 'range', 'batch'], default='auto'); p.add_argument('--parts', action='store_true')
    p.add_argument('--parse-threads', type=int, default=0, help='parser threads per rank over record-aligned slices of a plain input (0: max(1, min(4, t / 4)))')
    p.add_argument('--debug', action='store_true', help='log every read the aligner skipped with its status (vacmap:127; mammap_clrnano.py:24120-24123)')
    return p


def _open_output(path):
    """'-' / .sam: text; .bam / .sorted.bam: a `samtools view -b` / `samtools sort --write-index` pipe (output_functions.py:200-208)"""
    if path == '-':
        return sys.stdout.buffer, None
    if path.endswith('.sam'):
        return open(path, 'w+b'), None            # (read + write: the writer maps the file's end to copy a window's lines in by several threads)
    if not shutil.which('samtools'):
        sys.exit('writing %s needs the samtools binary on PATH (the reference pipes SAM text into it too); write .sam instead' % path)
    cmd = ['samtools', 'sort', '-@', '8', '--write-index', '-o', path, '-'] if path.endswith('sorted.bam') else ['samtools', 'view', '-b', '-@', '8', '-o', path, '-']
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, bufsize=1 << 20)
    return proc.stdin, proc


last_timing = {}          # wall seconds of the last main() call by phase (tools/driver_bench.py reads it)


def _run_asm(args, lib, ctx, index, prm, rg, mark, out, proc, world, rank, text_group, t_start):
    """-mode asm (src/vacmap/vacmap:245-255, :394-411; worker mammap_asm.py:23462-23511): every input sequence is an assembly contig. --eqx is
    forced and maxdivergence set to 1 by vm_params_default(VM_MODE_ASM); contigs are aligned in groups (the long ones of a group side by side on
    the GPU) and their SAM lines written in input order. A contig the reference would skip (raised) is logged and skipped.
    With N ranks every rank parses the input, contig c of it goes to rank c mod N (contigs are independent: no data-path collective), and
    rank 0 gathers each group's lines and writes them in input order."""
    import numpy as np
    from .lib import Fastx, SamOpts, align_batch_raw, sam_emit
    if not args.workdir:
        sys.exit('workdir not provided! -workdir /path/to/workdir')                      # vacmap:247-249
    os.makedirs(args.workdir, exist_ok=True)
    # the fork hard-codes these, whatever -c / -maxdivergence say: check_num = -1 (mammap_asm.py:23206), maxdivergence = 1.0 (:23483), --eqx
    prm.eqx = 1; prm.check_num = -1; prm.maxdivergence = 1.0
    opts = SamOpts(int(bool(args.MD)), int(args.cs != 'long'), int(bool(args.L)), int(bool(mark)), int(bool(args.H)), int(bool(args.fakecigar)), rg['ID'].encode(), 1)
    seen = set(); n_contigs = n_lines = n_skipped = 0

    def blob(parts):
        off = np.zeros(len(parts) + 1, np.int64)
        np.cumsum([len(x) for x in parts], out=off[1:])
        return np.frombuffer(b''.join(parts) or b'\0', np.uint8), off

    def flush(group):
        """group: (input index, name, sequence, quality, comment — bytes) of consecutive input contigs, the same list on every rank"""
        nonlocal n_lines, n_skipped
        share = [g for g in group if g[0] % world == rank]
        done = {}                                                                           # input index -> text of its lines, None = skipped
        if share:
            nb, no = blob([g[1] for g in share]); sb, so = blob([g[2] for g in share])
            qb, qo = blob([g[3] for g in share]); cb, co = blob([g[4] for g in share])
            raw = align_batch_raw(ctx, index, prm, sb, so)
            # the asm emitter (iterator_get_bam_dict_str, mammap_asm.py:22757) in the native emitter: vm_sam_opts.asm_mode
            text, toff, _, ns = sam_emit(lib, index, opts, nb, no, sb, so, raw, quals=qb, qual_off=qo, comments=cb if args.copycomments else None,
                                         com_off=co if args.copycomments else None, nthreads=max(1, args.t))
            for x, g in enumerate(share):
                if raw.status[x] != 0:                                                      # the worker's except (:23493-23498)
                    sys.stderr.write('%s is not aligned.\n' % g[1].decode()); done[g[0]] = None
                else:
                    done[g[0]] = text[int(toff[x]):int(toff[x + 1])].tobytes()
            raw.close()
        if world > 1:
            from .dist import gather_lines
            parts = gather_lines(done, dst=0, group=text_group)
            if rank != 0:
                return
            done = {}
            for d in parts:
                done.update(d)
        for gi in sorted(done):
            if done[gi] is None:
                n_skipped += 1
                continue
            out.write(done[gi])
            n_lines += done[gi].count(b'\n')

    group, gbases = [], 0
    for grp in args.read:
        for path in grp:
            # (.bam input like every other mode: vacmap:452-470)
            chunks = _bam_chunks(path, 64) if path.endswith('.bam') else iter(lambda rd=Fastx(path, lib=lib): rd.read(64), None)
            for ch in chunks:
                nb, no, sb, so = ch['names'].tobytes(), ch['names_off'], ch['seqs'].tobytes(), ch['seqs_off']
                qb, qo, cb, co = ch['quals'].tobytes(), ch['quals_off'], ch['comments'].tobytes(), ch['comments_off']
                for i in range(len(so) - 1):
                    nm = nb[no[i]:no[i + 1]]
                    if nm in seen:
                        continue
                    seen.add(nm)
                    keep = n_contigs % world == rank                                       # the other ranks' contigs only hold their place
                    group.append((n_contigs, nm, sb[so[i]:so[i + 1]] if keep else b'', (b'' if args.Q or not keep else qb[qo[i]:qo[i + 1]]),
                                  cb[co[i]:co[i + 1]] if keep else b''))
                    n_contigs += 1
                    gbases += so[i + 1] - so[i]
                    if len(group) >= 64 * world or gbases >= 400_000_000 * world:
                        flush(group); group, gbases = [], 0
    if group:
        flush(group)
    if rank == 0:
        if proc is not None:
            out.close(); proc.wait()
        elif args.o != '-':
            out.close()
        else:
            out.flush()
        tt = max(time.time() - t_start, 0.001)
        sys.stderr.write('vacmapx: %d contigs, %d SAM lines, %d contigs skipped, %.1f s\n' % (n_contigs, n_lines, n_skipped, tt))
    return 0


def _keep_heap_pages():
    """A window of input is ~1 GB of blobs, a batch's SAM text ~125 MB, all short-lived: glibc serves blocks that large from fresh mmap()s and
    unmaps them on free, so every byte the host pipeline writes lands on a page that is faulted in and zeroed first — measured on the FASTQ
    parser alone: 1.25 -> 2.85 GB/s once the pages are reused. One arena, no mmap for malloc, no trimming: freed blocks stay in the heap and
    the next window / batch reuses them. (VMX_DRIVER_MALLOPT=0 leaves the allocator alone.)"""
    if os.environ.get('VMX_DRIVER_MALLOPT', '1') == '0':
        return
    try:
        import ctypes
        libc = ctypes.CDLL(None)
        libc.mallopt(-8, 1)                     # M_ARENA_MAX: worker threads allocate from the main heap too
        libc.mallopt(-4, 0)                     # M_MMAP_MAX: no direct mmap for large requests
        libc.mallopt(-1, 2 ** 31 - 1)           # M_TRIM_THRESHOLD: do not hand freed memory back
        libc.mallopt(-2, 256 << 20)             # M_TOP_PAD: grow the heap in large steps
    except Exception:
        pass


def main(argv=None, comm=None):
    """comm: an initialised torch.distributed module (tests); under torchrun (WORLD_SIZE > 1) the process group is created here"""
    t_start = time.time()
    args, _unknown = build_parser().parse_known_args(argv)          # unknown flags are ignored like the reference's parse_known_args (vacmap:152)
    if comm is None:                            # (a process of its own, not a test harness that shares the interpreter)
        _keep_heap_pages()
    if args.o != '-' and not (args.o.endswith('.sam') or args.o.endswith('.bam')):
        sys.exit("Output path must end with .sam, .bam, .sorted.bam, or be '-' for stdout.")
    world, rank, local_rank = 1, 0, 0
    own_group = False
    # VMX_FORCE_DIST=1: the N-rank start-up at world 1 too (process group over nccl = RCCL, gloo text group, index through a replica built from the
    # broadcast metadata): what an 8-GPU run executes, testable on one GPU (tests/test_gpu_dist.py)
    force_dist = comm is None and os.environ.get('VMX_FORCE_DIST') == '1'
    if comm is None and (int(os.environ.get('WORLD_SIZE', '1')) > 1 or force_dist):
        import torch, torch.distributed as comm
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
        torch.cuda.set_device(local_rank)
        comm.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank), rank=int(os.environ.get('RANK', '0')), world_size=int(os.environ.get('WORLD_SIZE', '1')))
        own_group = True
    text_group = None
    if own_group:
        # SAM text is host data gathered from a writer THREAD: a gloo (CPU) group suits it — an object gather over the nccl group would
        # stage its byte tensors on the calling thread's current device (cuda:0 in a new thread), not the rank's — and its timeout turns
        # a peer that died into an error instead of a hang
        import datetime
        text_group = comm.new_group(backend='gloo', timeout=datetime.timedelta(seconds=int(os.environ.get('VMX_GATHER_TIMEOUT', '1800'))))
    if comm is not None:
        world, rank = comm.get_world_size(), comm.get_rank()
    if rank == 0 and args.o != '-' and os.path.exists(args.o) and not args.force:
        sys.exit('%s exists (use --force)' % args.o)
    if comm is None and world == 1 and __name__ == '__main__':
        os.environ.setdefault('VACMAPX_SKIP_TORCH', '1')        # one GPU from the command line: nothing here needs torch (its import is 1.5-2 s)
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
    if world > 1 or force_dist:
        import torch
        from .dist import broadcast_index
        dev = torch.device('cuda', device) if torch.cuda.is_available() else torch.device('cpu')
        index, t_bc = broadcast_index(ctx, index, src=0, device=dev, self_replica=(world == 1))
        if force_dist and rank == 0:
            sys.stderr.write('vacmapx: VMX_FORCE_DIST: process group %s, world %d, index through broadcast_index in %.2f s\n' % (comm.get_backend(), world, t_bc))
    prm = lib.params(args.mode)                     # mode defaults (vacmap:257-296), then the explicit options
    prm.check_num = args.c; prm.global_maxdiff = args.globalmaxdiff; prm.local_maxdiff = args.localmaxdiff
    prm.eqx = 1 if args.eqx else 0; prm.hardclip = 1 if args.H else 0
    if args.nodiscard: prm.nodiscard = 1
    if args.maxdivergence is not None: prm.maxdivergence = args.maxdivergence
    if args.globalpenalty is not None: prm.global_skipcost = args.globalpenalty
    if args.localpenalty is not None: prm.local_skipcost = args.localpenalty
    names = index.names
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
    mark = args.markunbalancetra or args.mode in ('H', 'L')          # mode defaults of vacmap:286-296 (False for asm unless asked)
    from .lib import SamOpts, Fastx, PinnedPool, align_batch_raw, sam_emit, blob_gather, blob_gather_parts, blob_write_parts
    opts = SamOpts(int(bool(args.MD)), int(args.cs != 'long'), int(bool(args.L)), int(bool(mark)), int(bool(args.H)), int(bool(args.fakecigar)), rg['ID'].encode())
    plain = all(_is_plain_fastx(pth) for grp in args.read for pth in grp)
    range_mode = world > 1 and args.mode != 'asm' and args.shard != 'batch' and plain and args.o.endswith('.sam')
    if world > 1 and args.shard == 'range' and not range_mode:
        sys.exit('--shard range needs uncompressed FASTA / FASTQ input and a .sam output path')
    if range_mode:
        # every rank writes <out>.partNNN and rank 0 joins them: they must land in one directory. Ranks on several hosts only see each other's
        # parts on a shared file system — nothing here can check that, so `auto` falls back to the batch scheme (text gathered over gloo) and an
        # explicit `--shard range` goes on with a warning
        import socket
        hosts = [None] * world
        comm.all_gather_object(hosts, socket.gethostname(), group=text_group)
        if len(set(hosts)) > 1:
            if args.shard == 'auto':
                range_mode = False
                if rank == 0:
                    sys.stderr.write('vacmapx: ranks on %d hosts: --shard auto uses the batch scheme (SAM text gathered by rank 0); --shard range needs %s.partNNN on a shared file system\n' % (len(set(hosts)), args.o))
            elif rank == 0:
                sys.stderr.write('vacmapx: ranks on %d hosts with --shard range: %s.partNNN of every rank must be visible to rank 0 (shared file system)\n' % (len(set(hosts)), args.o))
    part_path = '%s.part%03d' % (args.o, rank) if range_mode else None
    out, proc = (None, None)
    if range_mode and rank != 0:
        out = open(part_path, 'w+b')
    if rank == 0:
        out, proc = (open(part_path, 'w+b'), None) if range_mode else _open_output(args.o)
        for ln in sam.header_lines([(n, ln_) for n, ln_ in zip(names, index.lens)], ' '.join(sys.argv if argv is None else ['vacmapx'] + list(argv)), rg):
            out.write(ln.encode() + b'\n')
    if args.mode == 'asm':
        rc = _run_asm(args, lib, ctx, index, prm, rg, mark, out, proc, world, rank, text_group, t_start)
        ctx.close()
        if own_group:                                   # no rank leaves while rank 0 still gathers and writes
            comm.barrier(); comm.destroy_process_group()
        return rc
    # Ramped start (VMX_DRIVER_RAMP=1; off by default): the stream starts on ONE context as soon as its pools are sized, the others are sized in the background and
    # join one by one (Pipeline.run_stream, ramp). Measured on the 1.64 M-read input: the first batch starts 1.6 s after the loop begins instead of 5.9 s, but
    # hipMalloc under load is slower than on an idle device and slows the running batches down — the seventh context joins after ~1 M reads, and the loop
    # takes the same 12.8 s (`profiles/r05_zz_driver_long_ramped_start.json`). Default: every context is sized before the first batch runs.
    ramp_on = os.environ.get('VMX_DRIVER_RAMP', '0') == '1' and os.environ.get('VMX_NO_WARM') != '1'
    pipe = pipeline.Pipeline(index, prm, device=device, inflight=1 if ramp_on else (args.inflight or 5), first_ctx=ctx)
    if os.environ.get('VMX_SPIN_SYNC') != '1':
        for cx in pipe.ctxs:
            cx.set_blocking_sync(True)                    # the emitters need the cores the waiting aligner threads would spin on
    # host threads (-t in all): `inflight` of them feed the GPU (gather a batch's reads, vm_align_batch) and mostly wait for it; the SAM
    # text of finished batches is produced by a pool of emit_jobs concurrent vm_sam_emit calls of emit_threads threads each, so that the
    # GPU never waits for text and the text never waits for the GPU
    emit_total = int(os.environ.get('VMX_EMIT_THREADS', '0')) or args.t
    emit_jobs = max(1, min(4, emit_total // 4))
    emit_threads = max(1, emit_total // emit_jobs)
    counts = {'reads': 0, 'lines': 0, 'skipped': 0}
    win_reads = max(1, args.batch_reads * args.window_batches)
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    parse_threads = args.parse_threads or int(os.environ.get('VMX_PARSE_THREADS', '0')) or max(1, min(4, args.t // 4))

    def chunks_of(path):
        """blob chunks of one input in file order. A plain FASTA / FASTQ file is cut into record-aligned slices (vm_fastx_open_range): in range
        mode the rank takes its own byte range of the file, and `parse_threads` threads parse slices ahead of the consumer (one thread
        parses ~3 GB/s: a third of what one GPU aligns)"""
        if path.endswith('.bam'):
            yield from _bam_chunks(path, win_reads)
            return
        if not _is_plain_fastx(path):
            rd = Fastx(path, lib=lib)
            yield from iter(lambda: rd.read(win_reads), None)
            return
        size = os.path.getsize(path)
        lo, hi = (size * rank // world, size * (rank + 1) // world) if range_mode else (0, size)
        target = max(1 << 20, int(float(os.environ.get('VMX_SLICE_MB', '1024')) * (1 << 20)))
        ns = max(1, -(-(hi - lo) // target))
        cuts = [lo + (hi - lo) * i // ns for i in range(ns + 1)]

        def parse(a, b):
            rd = Fastx(path, lib=lib, byte_range=(a, b))
            try:
                return list(iter(lambda: rd.read(win_reads), None))
            finally:
                rd.close()
        with ThreadPoolExecutor(max_workers=parse_threads) as pool:
            futs, nxt = [], 0
            while nxt < ns or futs:
                while nxt < ns and len(futs) < parse_threads + 1:
                    futs.append(pool.submit(parse, cuts[nxt], cuts[nxt + 1])); nxt += 1
                yield from futs.pop(0).result()

    def windows():
        """input records in arrival order as blobs (names, upper-cased sequences, qualities, comments), de-duplicated by name
        (vacmap:457,475,487; in range mode inside the rank's own part of the input), one window of at most win_reads reads at a time"""
        seen = set()
        file_no = -1
        for group in args.read:
            for path in group:
                file_no += 1
                for ch in chunks_of(path):
                    n = len(ch['seqs_off']) - 1
                    nb, no = ch['names'].tobytes(), ch['names_off']
                    keep = []
                    for i in range(n):
                        nm = nb[no[i]:no[i + 1]]
                        if nm in seen:
                            continue
                        seen.add(nm); keep.append(i)
                    if len(keep) < n:
                        ix = np.asarray(keep, dtype=np.int64)
                        for key in ('names', 'seqs', 'quals', 'comments'):
                            ch[key], ch[key + '_off'] = blob_gather(lib, ch[key], ch[key + '_off'], ix)
                    if range_mode and len(ch['names_off']) > 1:
                        rank_hashes.append(name_hashes(ch['names'], ch['names_off']))      # compared across the ranks at the end of the run
                        rank_hash_file.append(np.full(len(ch['names_off']) - 1, file_no, np.int32))  # ... in INPUT order: (file, byte range = rank)
                    if args.Q:
                        ch['quals_off'] = np.zeros(len(ch['seqs_off']), np.int64)
                    if not args.copycomments:
                        ch['comments_off'] = np.zeros(len(ch['seqs_off']), np.int64)
                    if len(ch['seqs_off']) > 1:
                        yield ch

    rank_hashes = []
    rank_hash_file = []
    prog = {'t0': time.time(), 't': time.time(), 'n': 0, 'next': 100000}

    def progress(count):
        """the reference's progress line, every 100 000 sequences (vacmap:498-514)"""
        if rank != 0 or count < prog['next']:
            return
        now = time.time()
        dt, tt = max(now - prog['t'], 0.001), max(now - prog['t0'], 0.001)
        sys.stderr.write('%d / sec in the last %d minutes, %d / sec AVG. %d sequences processed.\n'
                         % (round((count - prog['n']) / dt), max(round(dt / 60), 1), round(count / tt), count))
        prog['t'], prog['n'] = now, count
        prog['next'] = (count // 100000 + 1) * 100000

    tm = {'setup': time.time() - t_start, 'wait_input': 0.0, 'assemble_write': 0.0, 'job_gather': 0.0, 'job_align': 0.0, 'job_emit': 0.0}
    tml = threading.Lock()
    errs = []
    n_slots = max(2, int(os.environ.get('VMX_DRIVER_WINDOWS', '10')))      # (1.6 M-read run, profiles/r05_driver_long_*: 3 windows of 8 batches 145 k reads/s, 6 x 8: 180 k, 4 x 16: 195 k, 10 x 4: 205 k)
    slots = threading.Semaphore(n_slots)                # windows in memory at a time (input blobs + SAM text)
    oq = queue.Queue()                                  # windows in input order -> writer
    emit_pool = ThreadPoolExecutor(max_workers=emit_jobs)

    class Window:
        def __init__(self, wnd, plan):
            self.wnd, self.plan = wnd, plan
            self.has_q = bool(wnd['quals_off'][-1]); self.has_c = bool(wnd['comments_off'][-1])
            self.futs = [None] * len(plan)
            self.left = len(plan)
            self.ready = threading.Event()              # every batch of the window has been aligned and handed to the emit pool
            if not plan:
                self.ready.set()

    pending = []                                        # windows taken from the reader ahead of the workers (the warm-up below)

    def job_source():
        """(window, batch index) in schedule order; a window is planned when the first worker reaches it"""
        while not errs:
            while not slots.acquire(timeout=0.2):       # (never parked for good: a failure elsewhere must end the run, not hang it)
                if errs:
                    oq.put(None)
                    return
            t0 = time.time()
            wnd = pending.pop(0) if pending else wq.get()
            with tml:
                tm['wait_input'] += time.time() - t0
            if isinstance(wnd, BaseException):
                errs.append(wnd); wnd = None
            if wnd is None:
                slots.release()
                break
            counts['reads'] += len(wnd['seqs_off']) - 1
            progress(counts['reads'])
            plan = pipeline.plan_batches(np.diff(wnd['seqs_off']), args.batch_reads, args.window_batches)
            if not range_mode:
                plan = [plan[i] for i in range(rank, len(plan), world)]        # static sharding: batch i -> rank i mod N (range mode: the rank's own reads)
            w = Window(wnd, plan)
            oq.put(w)
            for i in range(len(plan)):
                yield w, i
        oq.put(None)

    # the batch's reads are gathered into page-locked memory: vm_align_batch's upload becomes a DMA the aligner thread does not wait for
    # (VMX_DRIVER_PINNED=0: pageable numpy arrays, the runtime stages the copy on the calling thread)
    pinned = PinnedPool(lib, device) if os.environ.get('VMX_DRIVER_PINNED', '1') != '0' else None

    def emit(w, i, ix, sb, so, raw):
        t0 = time.time()
        try:
            if os.environ.get('VMX_SKIP_EMIT') == '1':          # diagnostic: aligners alone
                raw.close()
                return ix, np.zeros(0, np.uint8), np.zeros(len(ix) + 1, np.int64), 0, 0
            wnd = w.wnd
            nb, no = blob_gather(lib, wnd['names'], wnd['names_off'], ix)
            qb, qo = blob_gather(lib, wnd['quals'], wnd['quals_off'], ix) if w.has_q else (None, None)
            cb, co = blob_gather(lib, wnd['comments'], wnd['comments_off'], ix) if w.has_c else (None, None)
            text, toff, nl, ns = sam_emit(lib, index, opts, nb, no, sb, so, raw, quals=qb, qual_off=qo, comments=cb, com_off=co, nthreads=emit_threads)
            if args.debug and ns:                       # the reference's per-read failure log (--debug, mammap_clrnano.py:24120-24123)
                nbb = nb.tobytes() if hasattr(nb, 'tobytes') else bytes(nb)
                for x in np.nonzero(np.asarray(raw.status) != 0)[0]:
                    sys.stderr.write('vacmapx --debug: read %s skipped, status %d\n' % (nbb[int(no[x]):int(no[x + 1])].decode(errors='replace'), int(raw.status[x])))
            raw.close()
            with tml:
                tm['job_emit'] += time.time() - t0
            return ix, text, toff, nl, ns
        finally:
            if pinned is not None:
                pinned.release(sb)

    # Feeders (VMX_DRIVER_FEEDERS, default 2; 0 = the round-4 form: gather and upload on the aligner thread, inside vm_align_batch): threads with a context of
    # their own take the next batch of the schedule, gather its reads into page-locked memory and stream them into HBM (vm_reads_reupload into a few reusable
    # slots) AHEAD of the aligning contexts, which then run vm_align_resident — the aligner thread's context no longer idles through the gather, and the copy of
    # batch i + 1 runs under the kernels of batch i (bench.py's host-input pass: Pipeline.run_host_blobs(prefetch=True))
    n_feed = int(os.environ.get('VMX_DRIVER_FEEDERS', '2')) if pinned is not None else 0
    feed = {'free': queue.Queue(), 'ready': queue.Queue(), 'ctxs': [], 'slots': [], 'lock': threading.Lock()}

    def feeder(fcx, src):
        try:
            while not errs:
                with feed['lock']:
                    job = next(src, None)
                if job is None:
                    break
                w, i = job
                ix = w.plan[i]
                t0 = time.time()
                sb, so = blob_gather(lib, w.wnd['seqs'], w.wnd['seqs_off'], ix, alloc=pinned.get)
                sl = feed['free'].get()
                while sl is None:                        # (a None is only queued to wake a feeder up on an error)
                    if errs:
                        return
                    sl = feed['free'].get()
                sl.reupload(sb, so, ctx=fcx)
                with tml:
                    tm['job_gather'] += time.time() - t0
                feed['ready'].put((w, i, ix, sb, so, sl))
        except BaseException as e:
            errs.append(e)
        finally:
            feed['ready'].put(None)

    def fed_jobs(n_feeders):
        """what the feeders prepared, in completion order, until all of them are done"""
        done = 0
        while done < n_feeders:
            item = feed['ready'].get()
            if item is None:
                done += 1
                continue
            yield item

    def align(job, cx):
        """one batch: gather its reads from the window, align (GPU), hand the records to the emit pool; the library calls release the GIL"""
        if len(job) == 6:                                # prepared by a feeder: reads already in HBM
            w, i, ix, sb, so, sl = job
            t0 = t1 = time.time()
            raw = sl.align_raw(index, prm, ctx=cx)
            feed['free'].put(sl)
        else:
            w, i = job
            ix = w.plan[i]
            t0 = time.time()
            sb, so = blob_gather(lib, w.wnd['seqs'], w.wnd['seqs_off'], ix, alloc=pinned.get if pinned is not None else None)
            t1 = time.time()
            raw = align_batch_raw(cx, index, prm, sb, so)
        t2 = time.time()
        w.futs[i] = emit_pool.submit(emit, w, i, ix, sb, so, raw)
        with tml:
            tm['job_gather'] += t1 - t0; tm['job_align'] += t2 - t1; tm['device_s'] = tm.get('device_s', 0.0) + raw.stats['ms_total'] * 1e-3
            if os.environ.get('VMX_DRIVER_TIMING') == '2':
                sys.stderr.write('batch t=%.3f reads %d bases %d align %.3f device %.3f\n' % (t1 - t_loop, len(ix), int(so[-1]), t2 - t1, raw.stats['ms_total'] * 1e-3))
            w.left -= 1
            if w.left == 0:
                w.ready.set()

    out_fd = None
    if (rank == 0 or range_mode) and os.environ.get('VMX_DRIVER_WRITEV', '1') != '0':
        try:
            out_fd = out.fileno()
        except Exception:
            out_fd = None

    mmap_out = os.environ.get('VMX_DRIVER_MMAP_OUT', '0') == '1'      # (measured on the 1.6 M-read run: 7.4 s of writev against 24.6 s through the mapping — page faults of a fresh file do not scale; off)
    write_threads = max(1, int(os.environ.get('VMX_WRITE_THREADS', '0')) or min(6, max(2, args.t // 3)))

    def writer():
        """a window's lines in input order (one more gather over the concatenated batch texts) while later windows align and emit"""
        try:
            while True:
                w = oq.get()
                if w is None:
                    return
                while not w.ready.wait(0.2):
                    if errs:
                        return
                done = [f.result() for f in w.futs]
                nl, ns = sum(r[3] for r in done), sum(r[4] for r in done)
                parts = [(ix, text, toff) for ix, text, toff, _, _ in done]
                t0 = time.time()
                if world > 1 and not range_mode:
                    from .dist import gather_lines
                    allp = gather_lines((parts, nl, ns), dst=0, group=text_group)
                    if rank == 0:
                        parts = [p for rp in allp for p in rp[0]]; nl = sum(rp[1] for rp in allp); ns = sum(rp[2] for rp in allp)
                    else:
                        parts = []
                counts['lines'] += nl; counts['skipped'] += ns
                if parts:
                    if out_fd is not None:           # straight from the batches' texts to the file, no assembled copy of the window: into the file's own pages
                        out.flush()                  # by a few threads when it is a regular file (mmap), else one writev stream
                        pos = None
                        if mmap_out:
                            try:
                                pos = os.lseek(out_fd, 0, os.SEEK_CUR)
                            except OSError:
                                pos = None
                        blob_write_parts(lib, out_fd, [p[1] for p in parts], [p[2] for p in parts], [p[0] for p in parts], file_off=pos, nthreads=write_threads)
                    else:
                        txt = blob_gather_parts(lib, [p[1] for p in parts], [p[2] for p in parts], [p[0] for p in parts])
                        out.write(memoryview(txt))
                w.wnd = None; w.futs = None
                slots.release()
                with tml:
                    tm['assemble_write'] += time.time() - t0
        except BaseException as e:
            errs.append(e)
        finally:
            for _ in range(n_slots):                    # whatever ended the writer, nobody stays parked on a window slot
                slots.release()

    wq = queue.Queue(maxsize=2)

    def reader():
        """input parsing runs ahead of the aligners (the FASTX reader releases the GIL)"""
        try:
            for wnd in windows():
                wq.put(wnd)
            wq.put(None)
        except BaseException as e:
            wq.put(e)

    threading.Thread(target=reader, daemon=True).start()
    wt = threading.Thread(target=writer)
    wt.start()
    t_loop = time.time()
    # Size every context's grow-only work pools ONCE, before the stream starts: each context aligns the longest batch of the first window
    # (its result is dropped). Without this a context meets its first long-read batch windows later, outgrows the pools its earlier,
    # shorter batches sized, and pays for ~50 GB of re-allocation next to two other contexts' pools — seconds with the GPU idle
    # (rocprofv3 trace of a 131 k-read run: no kernel resident 73 % of the time, 2 s batches; `profiles/r03_l_*`).
    sb0_keep = None                                     # the sizing batch: also sizes the feeders' upload slots
    if os.environ.get('VMX_NO_WARM') != '1':
        first = wq.get()
        tm['first_window'] = time.time() - t_loop
        pending.append(first)
        if first is not None and not isinstance(first, BaseException):
            plan0 = pipeline.plan_batches(np.diff(first['seqs_off']), args.batch_reads, args.window_batches)
            if not range_mode:
                plan0 = [plan0[i] for i in range(rank, len(plan0), world)]
            if plan0:
                ix0 = max(plan0, key=lambda ix: int(np.diff(first['seqs_off'])[ix].sum()))
                sb0, so0 = blob_gather(lib, first['seqs'], first['seqs_off'], ix0)
                sb0_keep = (sb0, so0)

                # (one context after the other: hipMalloc serialises anyway, and a context that finds no memory left for its pools — a first window of
                # very long reads — is given up with the ones after it instead of failing the run: Pipeline.warm)
                try:
                    n_oom = pipe.warm(run=lambda cx: align_batch_raw(cx, index, prm, sb0, so0).close(), keep=2)
                    if n_oom and rank == 0:
                        sys.stderr.write('vacmapx: %d of %d batches in flight given up: no HBM left for their work pools\n' % (n_oom, n_oom + pipe.inflight))
                except BaseException as e:
                    errs.append(e)
        dropped = pipe.trim_to_memory(float(os.environ.get('VMX_MIN_FREE_GB', '10')))
        if dropped and rank == 0:
            sys.stderr.write('vacmapx: %d of %d batches in flight given up to keep HBM head-room\n' % (dropped, dropped + pipe.inflight))
        if args.inflight == 0 and not ramp_on and not dropped and sb0_keep is not None and not errs:
            # the scheduler's rule (bench.py runs the same): more batches in flight while another context's pools + head-room fit the HBM
            try:
                added = pipe.grow_to_memory(run=lambda cx: align_batch_raw(cx, index, prm, sb0_keep[0], sb0_keep[1]).close(), max_inflight=8)
                if added and os.environ.get('VMX_SPIN_SYNC') != '1':
                    for cx in pipe.ctxs:
                        cx.set_blocking_sync(True)
            except BaseException as e:
                errs.append(e)
        tm['warm'] = time.time() - t_loop
    fth = []
    ramp = None
    if ramp_on and sb0_keep is not None and not errs:
        spin = os.environ.get('VMX_SPIN_SYNC') == '1'
        ramp = dict(run=lambda cx: align_batch_raw(cx, index, prm, sb0_keep[0], sb0_keep[1]).close(), target=args.inflight, max_inflight=8,
                    on_ctx=None if spin else (lambda cx: cx.set_blocking_sync(True)))
    try:
        if n_feed > 0 and sb0_keep is not None:
            from .lib import Context as _Ctx, ResidentReads as _RR
            src = job_source()
            for f in range(n_feed):
                fcx = _Ctx(device, lib=lib)
                if os.environ.get('VMX_SPIN_SYNC') != '1':
                    fcx.set_blocking_sync(True)
                feed['ctxs'].append(fcx)
            for _ in range((max(args.inflight or 8, pipe.inflight) if ramp_on else pipe.inflight) + n_feed + 1):
                sl = _RR(feed['ctxs'][0], concat=sb0_keep[0], offsets=sb0_keep[1]); feed['slots'].append(sl); feed['free'].put(sl)
            fth = [threading.Thread(target=feeder, args=(fcx, src)) for fcx in feed['ctxs']]
            for t_ in fth:
                t_.start()
            pipe.run_stream(fed_jobs(n_feed), align, errs, ramp=ramp)
        else:
            pipe.run_stream(job_source(), align, errs, ramp=ramp)
        tm['aligners_done'] = time.time() - t_loop; tm['contexts'] = float(pipe.inflight)
    finally:
        if errs:
            oq.put(None)
            for _ in fth:
                feed['free'].put(None)
        for t_ in fth:
            t_.join()
        for sl in feed['slots']:
            sl.close()
        for fcx in feed['ctxs']:
            fcx.close()
        wt.join()
        emit_pool.shutdown(wait=True)
    if errs:
        raise errs[0]
    tm['loop'] = time.time() - t_loop
    last_timing.clear(); last_timing.update(tm); last_timing['reads'] = counts['reads']
    if os.environ.get('VMX_DRIVER_TIMING'):
        sys.stderr.write('vacmapx timing (s): %s\n' % ' '.join('%s=%.2f' % kv for kv in tm.items()))
    pipe.close()
    if pinned is not None:
        pinned.close()
    if range_mode:
        # every rank wrote its own part; the counts travel to rank 0 (three integers), which joins the parts in rank order (the reference's output order
        # is not the input's either: mammap_clrnano.py:24147-24150)
        out.close()
        from .dist import gather_lines
        # The reference drops a read name it has seen before, anywhere in the input (vacmap:457-487); a rank only sees its own byte ranges, so the
        # ranks' name hashes (8 bytes per read) travel with the counts and rank 0 leaves the later occurrences out while it joins the parts:
        # a name that also occurs in a lower rank's ranges (of any input file) goes. No cross-rank duplicate — the rule — costs one sort.
        # Round 6: "earlier" is the INPUT's order, not the rank's — every hash travels with the number of the input file it came from; inside a file
        # the ranks' byte ranges ascend with the rank. With two files, a name in file 1 inside rank 3's range and again in file 2 inside rank 0's keeps the
        # file-1 record (sorted by hash, then file, then rank: every occurrence but the first goes), as the reference's single pass over the files does.
        myh = np.concatenate(rank_hashes) if rank_hashes else np.zeros(0, np.uint64)
        myf = np.concatenate(rank_hash_file) if rank_hash_file else np.zeros(0, np.int32)
        allc = gather_lines((counts['reads'], counts['lines'], counts['skipped'], myh, myf), dst=0, group=text_group)
        if rank == 0:
            counts['reads'], counts['lines'], counts['skipped'] = (sum(c[i] for c in allc) for i in range(3))
            drop = cross_rank_duplicates([c[3] for c in allc], [c[4] for c in allc])
            n_dup = sum(len(v) for v in drop.values())
            if drop and args.parts:
                sys.stderr.write('vacmapx: %d read names occur in more than one rank\'s part (--parts keeps the parts as written: later occurrences are NOT removed)\n' % n_dup)
            if not args.parts:
                tj = time.time()
                gr, gl = _concat_parts(args.o, ['%s.part%03d' % (args.o, r) for r in range(world)], drop)
                counts['reads'] -= gr; counts['lines'] -= gl
                if gr:
                    sys.stderr.write('vacmapx: %d reads whose name occurred earlier in the input (another rank\'s range) were left out, %d SAM lines\n' % (gr, gl))
                last_timing['concat_parts'] = time.time() - tj
    if rank == 0:
        if proc is not None:
            out.close()
            rc = proc.wait()
            if rc != 0:
                sys.stderr.write('Error: samtools exited with code %d\n' % rc)
        elif range_mode:
            pass
        elif args.o != '-':
            out.close()
        else:
            out.flush()
        tt = max(time.time() - prog['t0'], 0.001)     # vacmap:535-541
        sys.stderr.write('User time (h:m:s): %d:%d:%d %d / sec AVG. %d sequences processed.\n' % (tt // 3600, (tt % 3600) // 60, tt % 60, round(counts['reads'] / tt), counts['reads']))
        sys.stderr.write('vacmapx: %d reads, %d SAM lines, %d reads skipped\n' % (counts['reads'], counts['lines'], counts['skipped']))
    if own_group:
        comm.barrier(); comm.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
