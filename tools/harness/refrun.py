"""Run the reference's own per-read path (get_readmap_DP_test, mammap_*.py) in the build container.

The reference's native dependencies are absent (vacmap_index, edlib): this module provides a
`vacmap_index`-shaped shim (Aligner / map / k_cigar / seq / seq_offset / k) and an `edlib.align`
implementation, both backed by the build's CPU oracle through ctypes (tests/oracle_lib.py). The composite
"reference Python + oracle primitives" is the operational reference.

Numba semantics: under numba every float argument is float64; under plain NumPy (NEP 50) a Python float is
"weak" and `40.0 + np.float32(x)` would round to float32. Option floats are therefore passed as np.float64 so that
the stubbed (plain CPython) run computes exactly what the numba-compiled reference computes.
"""
import os, sys, types
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, os.path.join(_ROOT, 'tests'))
sys.path.insert(0, _HERE)
import oracle_lib as O   # noqa: E402
import refload           # noqa: E402

DPLOG = None  # set to a list to record (kind, target, query) of every DP call the reference makes


class Aligner:
    """vacmap_index.Aligner shape (src/vacmap/vacmap:344-367; mammap_clrnano.py:23985, :24024, :24098)."""

    def __init__(self, path=None, w=10, k=15, oracle_index=None):
        self._ix = oracle_index if oracle_index is not None else O.Index.from_fasta(path, k=k, w=w)
        self.k = self._ix.k
        self.w = self._ix.w
        self.seq_offset = [(n.encode(), ln, off) for n, ln, off in zip(self._ix.names, self._ix.lens, self._ix.offsets)]
        self._name2i = {n: i for i, n in enumerate(self._ix.names)}

    def __bool__(self):
        return True

    def seq(self, name, start=0, end=None):
        return self._ix.seq(self._name2i[name], start, end)

    def map(self, seq, check_num=100, mid_occ=-1):
        a = self._ix.map(seq, check_num=check_num, mid_occ=mid_occ)
        return [tuple(int(v) for v in row) for row in a]


def k_cigar(target, query, match=2, mismatch=-4, gap_open_1=4, gap_extend_1=2, gap_open_2=24, gap_extend_2=1,
            bw=-1, zdropvalue=-1, eqx=False):
    """vacmap_index.k_cigar shape (mammap_clrnano.py:21554, :2381) -> (cigar, zdropcode, q_e, t_e, del, ins)"""
    if zdropvalue < 0:
        if DPLOG is not None:
            DPLOG.append((0, target, query))
        cg, sc = O.k_cigar_global(target, query, match, mismatch, gap_open_1, gap_extend_1, gap_open_2, gap_extend_2, eqx)
        return cg, 0, len(query), len(target), 0, 0
    assert gap_open_1 == gap_open_2 and gap_extend_1 == gap_extend_2
    if DPLOG is not None:
        DPLOG.append((1, target, query))
    sc, t_e, q_e = O.k_extend(target, query, match, mismatch, gap_open_1, gap_extend_1, bw, zdropvalue)
    return '', 0, q_e, t_e, 0, 0


def _edit(query, target):
    if DPLOG is not None:
        DPLOG.append((2, target, query))
    return O.edit_distance(query, target)


_vmi = types.ModuleType('vacmap_index')
_vmi.Aligner = Aligner
_vmi.k_cigar = k_cigar


def load(mode='H'):
    m = refload.load(mode, _vmi)
    import edlib
    edlib.set_impl(_edit)
    return m


def make_option(mode='H', **kw):
    """pdict as built by src/vacmap/vacmap:177-296 (floats as np.float64, see module docstring)."""
    if mode == 'L':
        ls, gs, dv = 59., 40., 0.1
    elif mode == 'H':
        ls, gs, dv = 40., 40., 0.2
    else:
        ls, gs, dv = 30., 30., 0.5
    o = {'golbal_skipcost': np.float64(gs), 'golbal_maxdiff': 50, 'local_skipcost': np.float64(ls), 'local_maxdiff': 30,
         'maxdivergence': np.float64(dv), 'nodiscard': mode not in ('L', 'H'), 'markunbalancetra': mode in ('L', 'H'),
         'H': False, 'c': 100, 'eqx': False, 'fakecigar': False, 'md': False, 'shortcs': True, 'cigar2cg': False,
         'rg-id': '1', 'debug': False, 'k': '15', 'w': '10', 'mode': mode, 'local_kmersize': 9, 'copycomments': False,
         'Q': False}
    o.update(kw)
    return o


class RefContext:
    def __init__(self, mode, aligner, option=None):
        self.m = load(mode)
        self.mode = mode
        self.al = aligner
        self.option = option or make_option(mode)
        from numba.typed import Dict, List
        self.contig2start = Dict(); self.contig2seq = Dict(); self.index2contig = List(); self.contig2iloc = {}
        for i, item in enumerate(aligner.seq_offset):
            name = item[0].decode()
            self.contig2start[name] = item[2]
            self.contig2seq[name] = aligner.seq(name).upper()
            self.index2contig.append(name)
            self.contig2iloc[name] = i

    def align(self, readid, seq):
        """returns (status, onemapinfolist) — status -1 when the reference raises (read skipped, :24116-24125)"""
        try:
            one, (al, raw), tra, fr = self.m.get_readmap_DP_test(
                readid, seq.upper(), self.contig2start, self.contig2seq, self.al, self.index2contig, self.option,
                hastra=False, redo_ratio=5, eqx=self.option['eqx'], check_num=self.option['c'])
            return 0, one
        except Exception as e:  # noqa
            self.last_exc = e
            return -1, []


def read_fasta(path):
    name, seqs, out = None, [], []
    for ln in open(path):
        ln = ln.rstrip()
        if ln.startswith('>'):
            if name is not None:
                out.append((name, ''.join(seqs)))
            name, seqs = ln[1:].split()[0], []
        elif ln:
            seqs.append(ln)
    if name is not None:
        out.append((name, ''.join(seqs)))
    return out


if __name__ == '__main__':
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/testdata/reference.fasta'
    rd = sys.argv[2] if len(sys.argv) > 2 else '/root/reference/testdata/read.fasta'
    mode = sys.argv[3] if len(sys.argv) > 3 else 'H'
    al = Aligner(ref, w=10, k=15)
    ctx = RefContext(mode, al)
    prm = O.params(mode)
    for name, seq in read_fasta(rd):
        st, one = ctx.align(name, seq)
        print('REF   ', st, [(t[1], t[2], t[3], t[4], t[5], t[6], t[7], len(t[8])) for t in one])
        if st < 0:
            import traceback; traceback.print_exception(type(ctx.last_exc), ctx.last_exc, ctx.last_exc.__traceback__)
        st2, recs = O.align_read(al._ix, seq, prm)
        print('ORACLE', st2, [(al._ix.names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], len(t[8])) for t in recs])
        same = st == st2 and len(one) == len(recs) and all(
            (a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]) == (al._ix.names[b[1]], b[2], b[3], b[4], b[5], b[6], b[7], b[8])
            for a, b in zip(one, recs))
        print('IDENTICAL' if same else 'DIFFERENT')
