"""`vacmap_index`-shaped interface on top of libvacmapx (what the reference's Python imports as `mp`).

Mirrors, name for name, the surface the reference uses:
    mp.Aligner(path, w=, k=)            src/vacmap/vacmap:344          -> Aligner
    .k, .seq_offset, .seq(name)         vacmap:358-363, mammap_clrnano.py:24024
    .map(seq, check_num=, mid_occ=)     mammap_clrnano.py:23985
    mp.k_cigar(target, query, ...)      mammap_clrnano.py:21554, :2381
    edlib.align(query=, target=, task='distance')   mammap_clrnano.py:19251
plus the batched entry that replaces the whole per-read function get_readmap_DP_test (:24023):
    Aligner.align_batch(reads, option) -> per read: list of 9-tuples (readid, contig, strand, q_st, q_en, r_st, r_en, mapq, cigar)

Every call runs HIP kernels through the C-ABI; without a GPU the constructor raises (no CPU fallback).
"""
import os
from . import lib as _lib

_ctx = None


def context(device=0):
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(device)
    return _ctx


def use_context(ctx):
    """bind the module-level calls (k_cigar, edlib_align) to an existing context"""
    global _ctx
    _ctx = ctx


class Aligner:
    def __init__(self, fn_idx_in=None, w=10, k=15, device=0, ctx=None, index=None, **kw):
        self.ctx = ctx or context(device)
        if index is not None:
            self.index = index
        elif fn_idx_in is not None and fn_idx_in.endswith('.vmx'):
            self.index = _lib.Index.load(self.ctx, fn_idx_in)
        else:
            # the reference names its cached index <ref>.w<w>_k<k>.mmi (vacmap:326); ours is <ref>.w<w>_k<k>.vmx
            # (indexfile.find_index: a cached file older than the FASTA, or one the loader rejects, is rebuilt, never trusted)
            from .indexfile import find_index
            self.index = find_index(self.ctx, fn_idx_in, k, w, write=False)
        self.k = self.index.k
        self.w = self.index.w
        self.seq_offset = [(n.encode(), ln, off) for n, ln, off in zip(self.index.names, self.index.lens, self.index.offsets)]
        self._name2i = {n: i for i, n in enumerate(self.index.names)}

    def __bool__(self):
        return self.index is not None

    def seq(self, name, start=0, end=None):
        return self.index.seq(self._name2i[name], start, end)

    def map(self, seq, check_num=100, mid_occ=-1):
        rows = self.ctx.map(self.index, seq, check_num=check_num, mid_occ=mid_occ)            # vm_map, the reference's call shape
        return [tuple(int(v) for v in r) for r in rows]

    def save(self, path):
        self.index.save(path)

    # ---- the batched path
    def params(self, option):
        """vm_params from the reference's option dict `pdict` (vacmap:177-296)"""
        p = self.ctx.lib.params(option.get('mode', 'H'))
        p.check_num = int(option.get('c', 100))
        p.global_maxdiff = int(option.get('golbal_maxdiff', 50)); p.local_maxdiff = int(option.get('local_maxdiff', 30))
        p.local_kmersize = int(option.get('local_kmersize', 9))
        p.eqx = int(bool(option.get('eqx', False))); p.hardclip = int(bool(option.get('H', False)))
        if 'nodiscard' in option:
            p.nodiscard = int(bool(option['nodiscard']))
        for key, field in (('golbal_skipcost', 'global_skipcost'), ('local_skipcost', 'local_skipcost'), ('maxdivergence', 'maxdivergence')):
            if key in option:
                setattr(p, field, float(option[key]))
        return p

    def align_batch(self, reads, option):
        """reads: list of (readid, seq). returns (status per read, list per read of 9-tuples like get_onemapinfolist :20760)"""
        prm = self.params(option)
        status, recs, stats = _lib.align_batch(self.ctx, self.index, prm, [s.upper() for _, s in reads])
        out = [[] for _ in reads]
        for t in recs:
            out[t[0]].append((reads[t[0]][0], self.index.names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]))
        return status, out, stats


def k_cigar(target, query, match=2, mismatch=-4, gap_open_1=4, gap_extend_1=2, gap_open_2=24, gap_extend_2=1, bw=-1, zdropvalue=-1, eqx=False):
    return context().k_cigar(target, query, match, mismatch, gap_open_1, gap_extend_1, gap_open_2, gap_extend_2, bw, zdropvalue, eqx)


def edlib_align(query, target, task='distance', **kw):
    return {'editDistance': context().edit_distance(query, target)}
