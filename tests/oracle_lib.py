"""ctypes binding of the CPU ORACLE (oracle/liboracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, tools/harness (golden generation), __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module. The product package (vacmap_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get('VMO_LIB') or os.path.join(_ROOT, 'oracle', 'liboracle.so')     # VMO_LIB: e.g. the sanitizer build (oracle/Makefile `asan`)


class Params(C.Structure):
    _fields_ = [('mode', C.c_int32), ('check_num', C.c_int32), ('mid_occ', C.c_int32), ('global_maxdiff', C.c_int32),
                ('local_maxdiff', C.c_int32), ('local_kmersize', C.c_int32), ('eqx', C.c_int32), ('hardclip', C.c_int32),
                ('nodiscard', C.c_int32), ('reserved', C.c_int32), ('global_skipcost', C.c_double),
                ('local_skipcost', C.c_double), ('maxdivergence', C.c_double)]


class Record(C.Structure):
    _fields_ = [('read_idx', C.c_int32), ('contig', C.c_int32), ('strand', C.c_int32), ('mapq', C.c_int32),
                ('q_st', C.c_int64), ('q_en', C.c_int64), ('r_st', C.c_int64), ('r_en', C.c_int64),
                ('cigar_off', C.c_int64), ('cigar_len', C.c_int64)]


class Chains(C.Structure):
    _fields_ = [('need_reverse', C.c_int32), ('mapq', C.c_int32), ('score', C.c_double), ('n_paths', C.c_int32),
                ('fast_used', C.c_int32), ('path_off', C.POINTER(C.c_int64)), ('path_anchors', C.POINTER(C.c_int64)),
                ('n_all', C.c_int32), ('all_scores', C.POINTER(C.c_double))]


MODES = {'H': 0, 'L': 1, 'S': 2, 'R': 3, 'asm': 4}
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', os.path.join(_ROOT, 'oracle'), '-j8'])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, i32, i64, dbl, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_char_p
    P = C.POINTER
    L.vmo_params_default.argtypes = [P(Params), C.c_int]
    L.vmo_index_build_fasta.restype = vp; L.vmo_index_build_fasta.argtypes = [cp, C.c_int, C.c_int]
    L.vmo_index_build_mem.restype = vp
    L.vmo_index_build_mem.argtypes = [C.c_int, P(cp), P(cp), P(i64), C.c_int, C.c_int]
    L.vmo_index_free.argtypes = [vp]
    for f in ('vmo_index_k', 'vmo_index_w', 'vmo_index_nseq', 'vmo_index_mid_occ'):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int
    for f in ('vmo_index_n_minimizers', 'vmo_index_n_distinct'):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = i64
    L.vmo_index_seq_name.argtypes = [vp, C.c_int]; L.vmo_index_seq_name.restype = cp
    L.vmo_index_seq_len.argtypes = [vp, C.c_int]; L.vmo_index_seq_len.restype = i64
    L.vmo_index_seq_offset.argtypes = [vp, C.c_int]; L.vmo_index_seq_offset.restype = i64
    L.vmo_index_seq.argtypes = [vp, C.c_int, i64, i64, vp]; L.vmo_index_seq.restype = i64
    L.vmo_index_hashes.argtypes = [vp]; L.vmo_index_hashes.restype = P(C.c_uint64)
    L.vmo_index_positions.argtypes = [vp]; L.vmo_index_positions.restype = P(C.c_uint64)
    L.vmo_sketch.argtypes = [cp, i64, C.c_int, C.c_int, vp, vp, vp]; L.vmo_sketch.restype = i64
    L.vmo_map.argtypes = [vp, cp, i64, C.c_int, C.c_int, P(P(i64))]; L.vmo_map.restype = i64
    L.vmo_free.argtypes = [vp]
    L.vmo_stage_v4.argtypes = [vp, C.c_int, vp, i64, i64, cp, i64, P(P(i64)), P(i64), P(C.c_int)]
    L.vmo_k_cigar_global.argtypes = [cp, i64, cp, i64] + [C.c_int] * 7 + [P(vp), P(i32)]
    L.vmo_k_extend.argtypes = [cp, i64, cp, i64] + [C.c_int] * 6 + [P(i32), P(i32)]
    L.vmo_edit_distance.argtypes = [cp, i64, cp, i64]; L.vmo_edit_distance.restype = i64
    L.vmo_strand_flip.argtypes = [vp, i64, i64]
    L.vmo_chain_global_raw.argtypes = [vp, i64, C.c_int, C.c_int, dbl, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.vmo_chain_global_raw.restype = i64
    L.vmo_decode_hit.argtypes = [vp, i64, i64, C.c_int, P(Params), P(Chains)]
    L.vmo_chains_free.argtypes = [P(Chains)]
    L.vmo_local_chain.argtypes = [vp, cp, i64, C.c_int, vp, vp, P(Params), P(dbl), P(P(i64)), P(i64), P(P(i64)), P(i64), P(i32)]
    L.vmo_extend.argtypes = [vp, cp, i64, vp, i64, C.c_int, C.c_int, C.c_int, P(Params), P(P(Record)), P(i64), P(vp), P(i32)]
    L.vmo_align_read.argtypes = [vp, cp, i64, P(Params), P(P(Record)), P(i64), P(vp)]
    L.vmo_align_batch.argtypes = [vp, P(Params), i64, cp, vp, C.c_int, P(P(Record)), P(i64), P(vp), vp]
    L.vmo_align_asm.argtypes = [vp, cp, i64, P(Params), i64, i64, i64, P(P(Record)), P(i64), P(vp)]
    L.vmo_test_asm_max_factor.argtypes = [dbl]
    L.vmo_asm_trace.argtypes = [vp, cp, i64, P(Params), i64, i64, i64, C.c_int, P(P(i64)), P(i64), P(P(i64)), P(i64)]
    L.vmo_decode_hit_asm.argtypes = [vp, cp, i64, P(Params), P(Chains)]
    L.vmo_chain_linked_raw.argtypes = [vp, i64, C.c_int, C.c_int, dbl, C.c_int, C.c_int, dbl, i64, vp, vp, i64, i64, vp, vp, vp]
    L.vmo_chain_linked_raw.restype = i64
    L.vmo_table.argtypes = [C.c_int, P(vp)]; L.vmo_table.restype = i64
    L.vmo_dplog_begin.argtypes = []; L.vmo_dplog_end.restype = i64
    L.vmo_dplog_get.argtypes = [i64, P(i32), P(vp), P(i64), P(vp), P(i64)]
    L.vmo_last_error.restype = cp
    _lib = L
    return L


def params(mode='H', **kw):
    p = Params()
    lib().vmo_params_default(C.byref(p), MODES[mode])
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


class Index:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError('oracle index build failed: %s' % lib().vmo_last_error().decode())
        self.h = handle
        L = lib()
        self.k = L.vmo_index_k(handle); self.w = L.vmo_index_w(handle); self.nseq = L.vmo_index_nseq(handle)
        self.names = [L.vmo_index_seq_name(handle, i).decode() for i in range(self.nseq)]
        self.lens = [L.vmo_index_seq_len(handle, i) for i in range(self.nseq)]
        self.offsets = [L.vmo_index_seq_offset(handle, i) for i in range(self.nseq)]
        self.mid_occ = L.vmo_index_mid_occ(handle)

    @classmethod
    def from_fasta(cls, path, k=15, w=10):
        return cls(lib().vmo_index_build_fasta(_b(path), k, w))

    @classmethod
    def from_seqs(cls, names, seqs, k=15, w=10):
        n = len(names)
        na = (C.c_char_p * n)(*[_b(x) for x in names])
        keep, ptrs, lens = [], [], []
        for x in seqs:                      # uint8 NumPy arrays are handed over in place (hg38-size references)
            if isinstance(x, np.ndarray):
                a = np.ascontiguousarray(x, dtype=np.uint8); keep.append(a); ptrs.append(a.ctypes.data); lens.append(a.size)
            else:
                b = _b(x); keep.append(b); ptrs.append(C.cast(C.c_char_p(b), C.c_void_p).value); lens.append(len(b))
        sa = C.cast((C.c_void_p * n)(*ptrs), C.POINTER(C.c_char_p))
        la = (C.c_int64 * n)(*lens)
        return cls(lib().vmo_index_build_mem(n, na, sa, la, k, w))

    def seq(self, i, st=0, en=None):
        if en is None:
            en = self.lens[i]
        buf = C.create_string_buffer(max(en - st, 1))
        n = lib().vmo_index_seq(self.h, i, st, en, buf)
        return buf.raw[:n].decode()

    def n_minimizers(self):
        return lib().vmo_index_n_minimizers(self.h)

    def n_distinct(self):
        return lib().vmo_index_n_distinct(self.h)

    def positions_view(self):
        """the position column without a copy (hg38-size comparisons)"""
        n = lib().vmo_index_n_minimizers(self.h)
        return np.ctypeslib.as_array(lib().vmo_index_positions(self.h), shape=(max(n, 1),))[:n]

    def minimizers(self):
        n = lib().vmo_index_n_minimizers(self.h)
        h = np.ctypeslib.as_array(lib().vmo_index_hashes(self.h), shape=(n,)).copy()
        p = np.ctypeslib.as_array(lib().vmo_index_positions(self.h), shape=(n,)).copy()
        return h, p

    def map(self, seq, check_num=100, mid_occ=-1):
        s = _b(seq)
        out = C.POINTER(C.c_int64)()
        n = lib().vmo_map(self.h, s, len(s), check_num, mid_occ, C.byref(out))
        a = np.ctypeslib.as_array(out, shape=(max(n, 1), 4))[:n].copy()
        lib().vmo_free(out)
        return a

    def __del__(self):
        try:
            lib().vmo_index_free(self.h)
        except Exception:
            pass


def sketch(seq, k, w):
    s = _b(seq)
    n = len(s)
    h = np.zeros(max(n, 1), np.uint64); p = np.zeros(max(n, 1), np.int32); z = np.zeros(max(n, 1), np.int8)
    m = lib().vmo_sketch(s, n, k, w, h.ctypes.data, p.ctypes.data, z.ctypes.data)
    return h[:m], p[:m], z[:m]


def k_cigar_global(t, q, match=2, mismatch=-4, o1=4, e1=2, o2=24, e2=1, eqx=False):
    t, q = _b(t), _b(q)
    out = C.c_void_p(); sc = C.c_int32()
    lib().vmo_k_cigar_global(t, len(t), q, len(q), match, mismatch, o1, e1, o2, e2, int(eqx), C.byref(out), C.byref(sc))
    s = C.string_at(out).decode()
    lib().vmo_free(out)
    return s, sc.value


def k_extend(t, q, match=2, mismatch=-4, o=4, e=4, bw=100, zdrop=50):
    t, q = _b(t), _b(q)
    te = C.c_int32(); qe = C.c_int32()
    sc = lib().vmo_k_extend(t, len(t), q, len(q), match, mismatch, o, e, bw, zdrop, C.byref(te), C.byref(qe))
    return sc, te.value, qe.value


def edit_distance(q, t):
    q, t = _b(q), _b(t)
    return lib().vmo_edit_distance(q, len(q), t, len(t))


def strand_flip(anchors, readlen):
    a = np.ascontiguousarray(anchors, dtype=np.int64).copy()
    f = lib().vmo_strand_flip(a.ctypes.data, len(a), readlen)
    return bool(f), a


def chain_global_raw(anchors_sorted, kmersize, skipcost=40., maxdiff=50, maxgap=1000, which=0, mode='H'):
    a = np.ascontiguousarray(anchors_sorted, dtype=np.int64)
    n = len(a)
    S = np.zeros(n, np.float64); P = np.zeros(n, np.int64); SA = np.zeros(n, np.int64)
    g = lib().vmo_chain_global_raw(a.ctypes.data, n, MODES[mode], kmersize, skipcost, maxdiff, maxgap, which,
                                   S.ctypes.data, P.ctypes.data, SA.ctypes.data)
    return g, S, P, SA


def decode_hit(anchors, readlen, kmersize, prm):
    a = np.ascontiguousarray(anchors, dtype=np.int64)
    ch = Chains()
    rc = lib().vmo_decode_hit(a.ctypes.data, len(a), readlen, kmersize, C.byref(prm), C.byref(ch))
    res = {'rc': rc, 'need_reverse': bool(ch.need_reverse), 'mapq': ch.mapq, 'score': ch.score, 'paths': [],
           'all_scores': [], 'fast_used': bool(ch.fast_used)}
    if rc == 0 and ch.n_paths > 0:
        off = [ch.path_off[i] for i in range(ch.n_paths + 1)]
        tot = off[-1]
        pa = np.ctypeslib.as_array(ch.path_anchors, shape=(max(tot, 1), 4))[:tot].copy()
        res['paths'] = [pa[off[i]:off[i + 1]] for i in range(ch.n_paths)]
    if rc == 0:
        res['all_scores'] = [ch.all_scores[i] for i in range(ch.n_all)]
    lib().vmo_chains_free(C.byref(ch))
    return res


def _take_rows(ptr, n):
    a = np.ctypeslib.as_array(ptr, shape=(max(n, 1), 4))[:n].copy()
    lib().vmo_free(ptr)
    return a


def local_chain(index, read, paths, prm):
    """paths: list of (m,4) arrays (descending read order). returns dict(rc, score, chain(desc), raw, variant)"""
    off = np.zeros(len(paths) + 1, np.int64)
    for i, p in enumerate(paths):
        off[i + 1] = off[i] + len(p)
    pa = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64).reshape(-1, 4) for p in paths]), dtype=np.int64)
    rd = _b(read)
    sc = C.c_double(); ch = C.POINTER(C.c_int64)(); nch = C.c_int64(); rw = C.POINTER(C.c_int64)(); nrw = C.c_int64()
    var = C.c_int32()
    rc = lib().vmo_local_chain(index.h, rd, len(rd), len(paths), off.ctypes.data, pa.ctypes.data, C.byref(prm), C.byref(sc),
                               C.byref(ch), C.byref(nch), C.byref(rw), C.byref(nrw), C.byref(var))
    return {'rc': rc, 'score': sc.value, 'chain': _take_rows(ch, nch.value), 'raw': _take_rows(rw, nrw.value), 'variant': var.value}


def _take_records(recs, n, blob, names=None):
    out = []
    for i in range(n):
        r = recs[i]
        cg = C.string_at(blob.value + r.cigar_off, r.cigar_len).decode()
        out.append((r.read_idx, r.contig, '+' if r.strand == 1 else '-', r.q_st, r.q_en, r.r_st, r.r_en, r.mapq, cg))
    lib().vmo_free(recs); lib().vmo_free(blob)
    return out


def extend(index, read, chain_asc, mapq, need_reverse, nofilter, prm):
    rd = _b(read)
    ch = np.ascontiguousarray(chain_asc, dtype=np.int64)
    recs = C.POINTER(Record)(); n = C.c_int64(); blob = C.c_void_p(); filt = C.c_int32()
    rc = lib().vmo_extend(index.h, rd, len(rd), ch.ctypes.data, len(ch), mapq, int(need_reverse), int(nofilter), C.byref(prm),
                          C.byref(recs), C.byref(n), C.byref(blob), C.byref(filt))
    return rc, _take_records(recs, n.value, blob), bool(filt.value)


def align_read(index, read, prm):
    rd = _b(read)
    recs = C.POINTER(Record)(); n = C.c_int64(); blob = C.c_void_p()
    rc = lib().vmo_align_read(index.h, rd, len(rd), C.byref(prm), C.byref(recs), C.byref(n), C.byref(blob))
    return rc, _take_records(recs, n.value, blob)


def align_asm(index, contig, prm, split_len=0, batch_anchors=0, window=0):
    """-mode asm on one assembly contig (mammap_asm.py:23204); the three sizes default to the reference's 500000 / 500000 / 100000"""
    rd = _b(contig)
    recs = C.POINTER(Record)(); n = C.c_int64(); blob = C.c_void_p()
    rc = lib().vmo_align_asm(index.h, rd, len(rd), C.byref(prm), split_len, batch_anchors, window, C.byref(recs), C.byref(n), C.byref(blob))
    return rc, _take_records(recs, n.value, blob)


def asm_trace(index, contig, prm, which, split_len=0, batch_anchors=0, window=0):
    """stages of the long-contig asm path (vmo_asm_trace): (rc, rows (n,4), batch ends)"""
    rd = _b(contig)
    rows = C.POINTER(C.c_int64)(); n = C.c_int64(); off = C.POINTER(C.c_int64)(); no = C.c_int64()
    rc = lib().vmo_asm_trace(index.h, rd, len(rd), C.byref(prm), split_len, batch_anchors, window, which, C.byref(rows), C.byref(n), C.byref(off), C.byref(no))
    a = np.ctypeslib.as_array(rows, shape=(max(n.value, 1), 4))[:n.value].copy()
    o = np.ctypeslib.as_array(off, shape=(no.value,)).copy()
    lib().vmo_free(rows); lib().vmo_free(off)
    return rc, a, o


def decode_hit_asm(index, contig, prm):
    """decode_hit of the -mode asm fork (mammap_asm.py:21280) -> dict(rc, need_reverse, mapq, score, paths)"""
    rd = _b(contig)
    ch = Chains()
    rc = lib().vmo_decode_hit_asm(index.h, rd, len(rd), C.byref(prm), C.byref(ch))
    res = {'rc': rc, 'need_reverse': bool(ch.need_reverse), 'mapq': ch.mapq, 'score': ch.score, 'fast_used': bool(ch.fast_used), 'paths': []}
    if rc == 0 and ch.n_paths > 0:
        off = [ch.path_off[i] for i in range(ch.n_paths + 1)]
        pa = np.ctypeslib.as_array(ch.path_anchors, shape=(max(off[-1], 1), 4))[:off[-1]].copy()
        res['paths'] = [pa[off[i]:off[i + 1]] for i in range(ch.n_paths)]
    lib().vmo_chains_free(C.byref(ch))
    return res


def chain_linked_raw(anchors, which, kmersize, skipcost, maxdiff, maxgap, g_max_scores=0., g_max_index=0, pre_S=None, pre_P=None, prereadloc=0):
    """linked chain DPs of -mode asm (which: 0 GC-exact :21686, 1 GC-fast :21871, 2 LC :21504) -> (g_max_index, S, P, S_arg)"""
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 4)
    n = len(a)
    ps = np.ascontiguousarray(pre_S if pre_S is not None else [], dtype=np.float64)
    pp = np.ascontiguousarray(pre_P if pre_P is not None else [], dtype=np.int64)
    S = np.zeros(n, np.float64); P = np.zeros(n, np.int64); SA = np.zeros(n, np.int64)
    g = lib().vmo_chain_linked_raw(a.ctypes.data, n, which, kmersize, float(skipcost), int(maxdiff), int(maxgap), float(g_max_scores), int(g_max_index),
                                   ps.ctypes.data, pp.ctypes.data, len(ps), int(prereadloc), S.ctypes.data, P.ctypes.data, SA.ctypes.data)
    return g, S, P, SA


def align_batch(index, reads, prm, nthreads=1):
    bs = [_b(r) for r in reads]
    off = np.zeros(len(bs) + 1, np.int64)
    for i, b in enumerate(bs):
        off[i + 1] = off[i] + len(b)
    cat = b''.join(bs)
    status = np.zeros(len(bs), np.int32)
    recs = C.POINTER(Record)(); n = C.c_int64(); blob = C.c_void_p()
    lib().vmo_align_batch(index.h, C.byref(prm), len(bs), cat, off.ctypes.data, nthreads, C.byref(recs), C.byref(n), C.byref(blob),
                          status.ctypes.data)
    return status, _take_records(recs, n.value, blob)


def stage_v4(index, fn, rows, arg=0, read=b''):
    """golden V4 stage entry: rows (n, 5) int64 = (segment, q, r, s, l); returns (rc, ret, rows out)"""
    rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 5)
    out = C.POINTER(C.c_int64)(); n = C.c_int64(); ret = C.c_int()
    rd = _b(read)
    rc = lib().vmo_stage_v4(index.h, fn, rows.ctypes.data, len(rows), int(arg), rd, len(rd), C.byref(out), C.byref(n), C.byref(ret))
    a = np.ctypeslib.as_array(out, shape=(max(n.value, 1), 5))[:n.value].copy()
    lib().vmo_free(out)
    return rc, ret.value, a


def table(which):
    ptr = C.c_void_p()
    n = lib().vmo_table(which, C.byref(ptr))
    ty = C.c_float if which < 4 else C.c_double
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(n,)).copy()


def fast_counters(reset=False):
    """(GC-fast, LC-fast, LC-mm-fast) call counts of this process since the last reset"""
    out = (C.c_int64 * 3)()
    lib().vmo_fast_counters(out, 1 if reset else 0)
    return tuple(out)


def surgery_counters(reset=False):
    """(drop_misplaced removals, merges, fix_simple_inv shifts of the left / right breakpoint) of this process since the last reset"""
    out = (C.c_int64 * 4)()
    lib().vmo_surgery_counters(out, 1 if reset else 0)
    return tuple(out)


def dplog(fn):
    """run fn() with DP-call logging on this thread; returns (fn result, [(kind, target, query)])"""
    L = lib()
    L.vmo_dplog_begin()
    res = fn()
    n = L.vmo_dplog_end()
    calls = []
    for i in range(n):
        kind = C.c_int32(); t = C.c_void_p(); tl = C.c_int64(); q = C.c_void_p(); ql = C.c_int64()
        L.vmo_dplog_get(i, C.byref(kind), C.byref(t), C.byref(tl), C.byref(q), C.byref(ql))
        calls.append((kind.value, C.string_at(t.value, tl.value).decode() if tl.value else '',
                      C.string_at(q.value, ql.value).decode() if ql.value else ''))
    return res, calls
