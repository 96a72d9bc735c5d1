"""Seeded synthetic references and long reads (SURVEY §8(d) "Data rule"): everything large is generated on the box.

  make_reference(lengths, seed)                 i.i.d. uniform ACGT contigs
  sample_reads(contigs, n, ..., seed)           ONT-shape (gamma lengths, 10 % error 4:3:3 sub:del:ins) or
                                                HiFi-shape (normal lengths, 0.5 % error) reads, 50/50 strand
  implant_svs(seq, ops)                         donor sequence with simple / nested SVs (vacsim grammar subset:
                                                DEL, INS, INV, DUP[:times], TRA-like cut&paste)
Pure NumPy; byte arrays (uint8 ASCII) in, byte arrays out.
"""
import numpy as np

_ACGT = np.frombuffer(b'ACGT', dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
_COMP[:] = ord('N')
for a, b in zip(b'ACGTacgt', b'TGCATGCA'):
    _COMP[a] = b


def revcomp(a):
    return _COMP[np.asarray(a, dtype=np.uint8)][::-1].copy()


def make_reference(lengths, seed=1):
    rng = np.random.default_rng(seed)
    return [_ACGT[rng.integers(0, 4, size=int(n), dtype=np.uint8)] for n in lengths]


# GRCh38 primary assembly, chr1..22, X, Y (bases): the proportions of the hg38-size synthetic reference
HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
                133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
                156040895, 57227415]
HG38_NAMES = ['chr%d' % i for i in range(1, 23)] + ['chrX', 'chrY']


def hg38_like_lengths(total=3_100_000_000):
    """24 contig lengths with hg38's chromosome proportions, summing to `total`"""
    tot = float(sum(HG38_LENGTHS))
    ln = [int(round(x * total / tot)) for x in HG38_LENGTHS]
    ln[0] += int(total) - sum(ln)
    return ln


def shared_reference(lengths, seed, rank, barrier, threads=None, tag='0', shm_dir='/dev/shm'):
    """make_reference_fast for N processes of one host: rank 0 generates the contigs once (with `threads` threads) into a file under /dev/shm, every
    rank maps it (shared pages) — eight ranks regenerating 3.1 Gb on cores / 8 threads each was the longest part of an 8-rank start. `barrier()` is the
    caller's collective barrier (torch.distributed.barrier). Returns read-only uint8 views, one per contig."""
    import os
    path = os.path.join(shm_dir, 'vacmapx_ref_seed%d_%s.u8' % (int(seed), tag))
    if rank == 0:
        cs = make_reference_fast(lengths, seed=seed, threads=threads)
        mm = np.lib.format.open_memmap(path + '.tmp', mode='w+', dtype=np.uint8, shape=(int(sum(int(x) for x in lengths)),))
        o = 0
        for c in cs:
            mm[o:o + len(c)] = c; o += len(c)
        mm.flush(); del mm, cs
        os.replace(path + '.tmp', path)
    barrier()
    whole = np.load(path, mmap_mode='r')
    offs = np.concatenate([[0], np.cumsum([int(x) for x in lengths])])
    contigs = [whole[offs[i]:offs[i + 1]] for i in range(len(lengths))]
    barrier()
    if rank == 0:
        os.unlink(path)                          # (the mappings keep the pages until every rank is done)
    return contigs


_LUT4 = None


def make_reference_fast(lengths, seed=3, threads=None):
    """i.i.d. uniform ACGT contigs for LARGE references: one random byte yields four bases through a 256-entry table of packed
    ASCII quadruples (seconds for 3.1 Gb instead of minutes). Contig i is a function of (seed, i) only."""
    global _LUT4
    from concurrent.futures import ThreadPoolExecutor
    import os
    if _LUT4 is None:
        v = np.arange(256, dtype=np.uint32)
        _LUT4 = (_ACGT[v & 3].astype(np.uint32) | (_ACGT[(v >> 2) & 3].astype(np.uint32) << 8) |
                 (_ACGT[(v >> 4) & 3].astype(np.uint32) << 16) | (_ACGT[(v >> 6) & 3].astype(np.uint32) << 24))

    def one(i):
        n = int(lengths[i]); m = (n + 3) // 4
        rng = np.random.default_rng([int(seed), i])
        r = rng.integers(0, 1 << 32, size=(m + 3) // 4, dtype=np.uint32).view(np.uint8)[:m]
        return _LUT4[r].view(np.uint8)[:n]
    nt = threads or min(len(lengths), len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 8, 16)
    with ThreadPoolExecutor(max_workers=max(1, nt)) as ex:
        return list(ex.map(one, range(len(lengths))))


def mutate(seq, err, rng, ratio=(4, 3, 3)):
    """i.i.d. per-base errors split sub:del:ins = ratio; insertions add a random base before the position."""
    n = len(seq)
    if err <= 0 or n == 0:
        return seq.copy()
    tot = float(sum(ratio))
    u = rng.random(n)
    ps, pd = err * ratio[0] / tot, err * ratio[1] / tot
    is_sub = u < ps
    is_del = (u >= ps) & (u < ps + pd)
    is_ins = (u >= ps + pd) & (u < err)
    out = seq.copy()
    # substitutions: shift to a different base
    idx = np.nonzero(is_sub)[0]
    if len(idx):
        code = np.searchsorted(_ACGT, out[idx])  # ACGT sorted ascending in ASCII: A C G T
        code = (code + rng.integers(1, 4, size=len(idx))) % 4
        out[idx] = _ACGT[code]
    keep = ~is_del
    rep = keep.astype(np.int64) + is_ins.astype(np.int64)
    res = np.repeat(out, rep)
    # positions of inserted bases: the first copy of every is_ins position that is also kept, or the only copy if deleted
    ins_idx = np.nonzero(is_ins)[0]
    if len(ins_idx):
        starts = np.cumsum(rep) - rep
        res[starts[ins_idx]] = _ACGT[rng.integers(0, 4, size=len(ins_idx))]
    return res


def sample_reads(contigs, n, mean_len=15000, err=0.10, seed=2, shape='ont', min_len=1000, max_len=100000, sd=2000):
    """returns list of (name, uint8 array, truth dict). shape 'ont': Gamma(2, mean/2); 'hifi': Normal(mean, sd)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    out = []
    for i in range(n):
        if shape == 'ont':
            L = int(np.clip(rng.gamma(2.0, mean_len / 2.0), min_len, max_len))
        else:
            L = int(max(rng.normal(mean_len, sd), min_len))
        while True:
            g = int(rng.integers(0, cum[-1]))
            c = int(np.searchsorted(cum, g, side='right') - 1)
            st = g - int(cum[c])
            if st + L <= lens[c]:
                break
            if lens[c] < L:
                L = int(lens[c]); st = 0
                break
        frag = contigs[c][st:st + L]
        strand = int(rng.integers(0, 2))
        if strand:
            frag = revcomp(frag)
        rd = mutate(frag, err, rng)
        out.append(('read%d' % i, rd, {'contig': c, 'start': st, 'end': st + L, 'strand': '-' if strand else '+'}))
    return out


def implant_svs(seq, ops):
    """ops: list of (kind, pos, length[, extra]) applied right-to-left on the ORIGINAL coordinates:
    ('DEL', p, n) ('INS', p, n, seed) ('INV', p, n) ('DUP', p, n, times) ('INVDUP', p, n)"""
    seq = np.asarray(seq, dtype=np.uint8)
    for op in sorted(ops, key=lambda o: -o[1]):
        kind, p, n = op[0], int(op[1]), int(op[2])
        if kind == 'DEL':
            seq = np.concatenate([seq[:p], seq[p + n:]])
        elif kind == 'INS':
            rng = np.random.default_rng(op[3] if len(op) > 3 else 0)
            seq = np.concatenate([seq[:p], _ACGT[rng.integers(0, 4, size=n)], seq[p:]])
        elif kind == 'INV':
            seq = np.concatenate([seq[:p], revcomp(seq[p:p + n]), seq[p + n:]])
        elif kind == 'DUP':
            t = int(op[3]) if len(op) > 3 else 1
            seq = np.concatenate([seq[:p + n]] + [seq[p:p + n]] * t + [seq[p + n:]])
        elif kind == 'INVDUP':
            seq = np.concatenate([seq[:p + n], revcomp(seq[p:p + n]), seq[p + n:]])
        else:
            raise ValueError(kind)
    return seq


def tostr(a):
    return np.asarray(a, dtype=np.uint8).tobytes().decode()


def sample_reads_concat(contigs, n, mean_len=15000, err=0.10, seed=2, shape='ont', min_len=1000, max_len=100000, sd=2000, around=None):
    """Vectorised variant of sample_reads for large batches: returns (uint8 concatenation, int64 offsets[n+1], truth arrays).
    Same read model (Gamma(2, mean/2) or Normal lengths, uniform start, 50/50 strand, i.i.d. errors 4:3:3) but all reads of
    the batch are mutated in one pass over the concatenated fragments. around = (contig indices, positions): every read is drawn
    ACROSS one of these places, picked at random, which lies 20-80 % into the read (reads over the implanted SVs of a donor genome)."""
    rng = np.random.default_rng(seed)
    lens_c = np.array([len(c) for c in contigs], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens_c)])
    if shape == 'ont':
        L = np.clip(rng.gamma(2.0, mean_len / 2.0, size=n), min_len, max_len).astype(np.int64)
    else:
        L = np.maximum(rng.normal(mean_len, sd, size=n), min_len).astype(np.int64)
    ci = np.zeros(n, np.int64); st = np.zeros(n, np.int64)
    if around is not None:
        pick = rng.integers(0, len(around[0]), size=n)
        ci = np.asarray(around[0], dtype=np.int64)[pick]
        L = np.minimum(L, lens_c[ci])
        st = np.asarray(around[1], dtype=np.int64)[pick] - (L * rng.uniform(0.2, 0.8, size=n)).astype(np.int64)
        st = np.clip(st, 0, lens_c[ci] - L)
    for i in range(n if around is None else 0):
        while True:
            g = int(rng.integers(0, cum[-1]))
            c = int(np.searchsorted(cum, g, side='right') - 1)
            s = g - int(cum[c])
            if lens_c[c] < L[i]:
                L[i] = lens_c[c]; s = 0
            if s + L[i] <= lens_c[c]:
                break
        ci[i] = c; st[i] = s
    strand = rng.integers(0, 2, size=n)
    foff = np.concatenate([[0], np.cumsum(L)])
    frag = np.empty(int(foff[-1]), np.uint8)
    for i in range(n):
        f = contigs[ci[i]][st[i]:st[i] + L[i]]
        frag[foff[i]:foff[i + 1]] = revcomp(f) if strand[i] else f
    tot = len(frag)
    u = rng.random(tot)
    ps, pd = err * 0.4, err * 0.3
    is_sub = u < ps
    is_del = (u >= ps) & (u < ps + pd)
    is_ins = (u >= ps + pd) & (u < err)
    idx = np.nonzero(is_sub)[0]
    if len(idx):
        code = np.searchsorted(_ACGT, frag[idx])
        frag[idx] = _ACGT[(code + rng.integers(1, 4, size=len(idx))) % 4]
    rep = (~is_del).astype(np.int64) + is_ins.astype(np.int64)
    out = np.repeat(frag, rep)
    ins_idx = np.nonzero(is_ins)[0]
    if len(ins_idx):
        starts = np.cumsum(rep) - rep
        out[starts[ins_idx]] = _ACGT[rng.integers(0, 4, size=len(ins_idx))]
    crep = np.concatenate([[0], np.cumsum(rep)])
    off = crep[foff].astype(np.int64)
    return out, off, {'contig': ci, 'start': st, 'len': L, 'strand': strand}
