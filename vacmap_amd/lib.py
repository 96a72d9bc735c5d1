"""ctypes binding of libvacmapx.so (include/vacmapx.h) — the reference-side binding a VACmap maintainer would add.

`load()` opens the in-tree HIP library (built by vacmap_amd/build.py for gfx950). There is no fallback: a missing library
raises, and without a GPU `Context()` raises VmxError(VM_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, 'libvacmapx.so')

VM_ERR_NO_DEVICE = -2
MODES = {'H': 0, 'L': 1, 'S': 2, 'R': 3}


class VmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('vacmapx error %d: %s' % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [('mode', C.c_int32), ('check_num', C.c_int32), ('mid_occ', C.c_int32), ('global_maxdiff', C.c_int32),
                ('local_maxdiff', C.c_int32), ('local_kmersize', C.c_int32), ('eqx', C.c_int32), ('hardclip', C.c_int32),
                ('nodiscard', C.c_int32), ('reserved', C.c_int32), ('global_skipcost', C.c_double),
                ('local_skipcost', C.c_double), ('maxdivergence', C.c_double)]


class Score(C.Structure):
    _fields_ = [('match', C.c_int32), ('mismatch', C.c_int32), ('o1', C.c_int32), ('e1', C.c_int32), ('o2', C.c_int32), ('e2', C.c_int32)]


class CigarOut(C.Structure):
    _fields_ = [('cigar', C.c_void_p), ('q_e', C.c_int32), ('t_e', C.c_int32), ('score', C.c_int32)]


class ChainsOut(C.Structure):
    _fields_ = [('need_reverse', C.POINTER(C.c_int32)), ('mapq', C.POINTER(C.c_int32)), ('score', C.POINTER(C.c_double)),
                ('fast_used', C.POINTER(C.c_int32)), ('read_path_off', C.POINTER(C.c_int64)), ('path_off', C.POINTER(C.c_int64)),
                ('path_anchors', C.POINTER(C.c_int64)), ('S', C.POINTER(C.c_double)), ('P', C.POINTER(C.c_int64)),
                ('S_arg', C.POINTER(C.c_int64)), ('gmax', C.POINTER(C.c_int64)), ('opcount', C.POINTER(C.c_int64))]


class LocalOut(C.Structure):
    _fields_ = [('status', C.POINTER(C.c_int32)), ('variant', C.POINTER(C.c_int32)), ('score', C.POINTER(C.c_double)),
                ('chain_off', C.POINTER(C.c_int64)), ('chain', C.POINTER(C.c_int64)), ('raw_off', C.POINTER(C.c_int64)),
                ('raw', C.POINTER(C.c_int64))]


class Record(C.Structure):
    _fields_ = [('read_idx', C.c_int32), ('contig', C.c_int32), ('strand', C.c_int32), ('mapq', C.c_int32),
                ('q_st', C.c_int64), ('q_en', C.c_int64), ('r_st', C.c_int64), ('r_en', C.c_int64),
                ('cigar_off', C.c_int64), ('cigar_len', C.c_int64)]


class BatchStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('n_reads', 'read_bases', 'n_minimizers', 'n_hits', 'n_anchors', 'n_local_hits', 'n_local_anchors',
                                          'n_segments', 'n_ed_problems', 'ed_cells', 'n_ext_problems', 'ext_cells', 'n_dp_problems',
                                          'dp_cells', 'n_records', 'cigar_bytes', 'aligned_bases', 'n_unmapped', 'n_failed')] + \
               [('ms_total', C.c_double), ('ms_stage', C.c_double * 16)]


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def _cat(seqs):
    bs = [_b(s) for s in seqs]
    off = np.zeros(len(bs) + 1, np.int64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    return b''.join(bs), off


class VmxLib:
    """Thin typed view of the C-ABI. `path` defaults to the in-tree libvacmapx.so."""

    def __init__(self, path=None):
        path = path or DEFAULT_SO
        if not os.path.exists(path):
            raise FileNotFoundError('%s not built: run `python -m vacmap_amd.build` (hipcc, gfx950). No fallback exists.' % path)
        L = self.L = C.CDLL(path)
        vp, i32, i64, dbl, cp, P = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_char_p, C.POINTER
        L.vm_last_error.restype = cp
        L.vm_version.restype = cp
        L.vm_free.argtypes = [vp]
        L.vm_params_default.argtypes = [P(Params), C.c_int]
        L.vm_ctx_create.argtypes = [C.c_int, P(vp)]
        L.vm_ctx_destroy.argtypes = [vp]
        L.vm_table.argtypes = [vp, C.c_int, P(vp)]; L.vm_table.restype = i64
        L.vm_edit_distance_batch.argtypes = [vp, i64, cp, vp, cp, vp, P(P(i64))]
        L.vm_edit_distance.argtypes = [vp, cp, i64, cp, i64]; L.vm_edit_distance.restype = i64
        L.vm_k_extend_batch.argtypes = [vp] + [C.c_int] * 6 + [i64, cp, vp, cp, vp, P(P(i32)), P(P(i32)), P(P(i32))]
        L.vm_k_cigar_batch.argtypes = [vp, P(Score), C.c_int, i64, cp, vp, cp, vp, P(vp), P(P(i64)), P(P(i32))]
        L.vm_k_cigar.argtypes = [vp, cp, i64, cp, i64, P(Score), C.c_int, C.c_int, C.c_int, P(CigarOut)]
        L.vm_chain_global_batch.argtypes = [vp, P(Params), C.c_int, i64, vp, vp, vp, C.c_int, P(ChainsOut)]
        L.vm_chains_out_free.argtypes = [P(ChainsOut)]

    def err(self):
        return self.L.vm_last_error().decode()

    def check(self, rc):
        if rc < 0:
            raise VmxError(rc, self.err())
        return rc

    def params(self, mode='H', **kw):
        p = Params()
        self.L.vm_params_default(C.byref(p), MODES[mode])
        for k, v in kw.items():
            setattr(p, k, v)
        return p


_default = None


def load():
    global _default
    if _default is None:
        _default = VmxLib()
    return _default


class Context:
    """one GPU + stream + work buffers (vm_ctx). Fails loudly without a device."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load()
        h = C.c_void_p()
        self.lib.check(self.lib.L.vm_ctx_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.L.vm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _take(self, ptr, n, dtype):
        a = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].astype(dtype, copy=True)
        self.lib.L.vm_free(ptr)
        return a

    def table(self, which):
        p = C.c_void_p()
        n = self.lib.check(self.lib.L.vm_table(self.h, which, C.byref(p)))
        ty = C.c_float if which < 4 else C.c_double
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ty)), shape=(n,)).copy()
        self.lib.L.vm_free(p)
        return a

    # ---- DP primitives
    def edit_distance_batch(self, queries, targets):
        q, qo = _cat(queries); t, to = _cat(targets)
        out = C.POINTER(C.c_int64)()
        self.lib.check(self.lib.L.vm_edit_distance_batch(self.h, len(queries), q, qo.ctypes.data, t, to.ctypes.data, C.byref(out)))
        return self._take(out, len(queries), np.int64)

    def edit_distance(self, q, t):
        q, t = _b(q), _b(t)
        return self.lib.check(self.lib.L.vm_edit_distance(self.h, q, len(q), t, len(t)))

    def k_extend_batch(self, targets, queries, match=2, mismatch=-4, o=4, e=4, bw=100, zdrop=50):
        t, to = _cat(targets); q, qo = _cat(queries)
        te = C.POINTER(C.c_int32)(); qe = C.POINTER(C.c_int32)(); sc = C.POINTER(C.c_int32)()
        n = len(targets)
        self.lib.check(self.lib.L.vm_k_extend_batch(self.h, match, mismatch, o, e, bw, zdrop, n, t, to.ctypes.data, q, qo.ctypes.data,
                                                    C.byref(te), C.byref(qe), C.byref(sc)))
        return self._take(sc, n, np.int32), self._take(te, n, np.int32), self._take(qe, n, np.int32)

    def k_cigar_batch(self, targets, queries, match=2, mismatch=-4, o1=4, e1=2, o2=24, e2=1, eqx=False):
        t, to = _cat(targets); q, qo = _cat(queries)
        n = len(targets)
        sc = Score(match, mismatch, o1, e1, o2, e2)
        cg = C.c_void_p(); co = C.POINTER(C.c_int64)(); ss = C.POINTER(C.c_int32)()
        self.lib.check(self.lib.L.vm_k_cigar_batch(self.h, C.byref(sc), int(eqx), n, t, to.ctypes.data, q, qo.ctypes.data,
                                                   C.byref(cg), C.byref(co), C.byref(ss)))
        off = self._take(co, n + 1, np.int64)
        blob = C.string_at(cg.value, int(off[-1])) if off[-1] else b''
        self.lib.L.vm_free(cg)
        cigars = [blob[off[i]:off[i + 1] - 1].decode() for i in range(n)]
        return cigars, self._take(ss, n, np.int32)

    def k_cigar(self, target, query, match=2, mismatch=-4, gap_open_1=4, gap_extend_1=2, gap_open_2=24, gap_extend_2=1,
                bw=-1, zdropvalue=-1, eqx=False):
        """mp.k_cigar's argument list and return tuple (cigar, zdropcode, q_e, t_e, del, ins) (mammap_clrnano.py:21554, :2381)"""
        t, q = _b(target), _b(query)
        sc = Score(match, mismatch, gap_open_1, gap_extend_1, gap_open_2, gap_extend_2)
        out = CigarOut()
        self.lib.check(self.lib.L.vm_k_cigar(self.h, t, len(t), q, len(q), C.byref(sc), bw, zdropvalue, int(eqx), C.byref(out)))
        cg = C.string_at(out.cigar).decode() if out.cigar else ''
        self.lib.L.vm_free(out.cigar)
        return cg, 0, out.q_e, out.t_e, 0, 0

    # ---- global chain stage
    def chain_global_batch(self, prm, kmersize, anchors_list, readlens, want_raw=False):
        n = len(anchors_list)
        aoff = np.zeros(n + 1, np.int64)
        for i, a in enumerate(anchors_list):
            aoff[i + 1] = aoff[i] + len(a)
        rows = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1, 4) for a in anchors_list]) if n else np.zeros((0, 4), np.int64))
        rl = np.ascontiguousarray(readlens, dtype=np.int64)
        out = ChainsOut()
        self.lib.check(self.lib.L.vm_chain_global_batch(self.h, C.byref(prm), kmersize, n, rows.ctypes.data, aoff.ctypes.data, rl.ctypes.data,
                                                        int(want_raw), C.byref(out)))
        res = []
        rpo = np.ctypeslib.as_array(out.read_path_off, shape=(n + 1,)).copy()
        npaths = int(rpo[-1])
        po = np.ctypeslib.as_array(out.path_off, shape=(npaths + 1,)).copy()
        nrows = int(po[-1])
        pa = np.ctypeslib.as_array(out.path_anchors, shape=(max(nrows, 1), 4))[:nrows].copy()
        tot = int(aoff[-1])
        for r in range(n):
            d = {'need_reverse': bool(out.need_reverse[r]), 'mapq': int(out.mapq[r]), 'score': float(out.score[r]),
                 'fast_used': bool(out.fast_used[r]), 'gmax': int(out.gmax[r]), 'opcount': int(out.opcount[r]),
                 'paths': [pa[po[p]:po[p + 1]] for p in range(rpo[r], rpo[r + 1])]}
            if want_raw:
                a, b = int(aoff[r]), int(aoff[r + 1])
                d['S'] = np.ctypeslib.as_array(out.S, shape=(max(tot, 1),))[a:b].copy()
                d['P'] = np.ctypeslib.as_array(out.P, shape=(max(tot, 1),))[a:b].copy()
                d['S_arg'] = np.ctypeslib.as_array(out.S_arg, shape=(max(tot, 1),))[a:b].copy()
            res.append(d)
        self.lib.L.vm_chains_out_free(C.byref(out))
        return res
