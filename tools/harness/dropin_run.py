#!/usr/bin/env python3
"""Drop-in experiment (build container only): the REFERENCE's own Python (get_readmap_DP_test, imported in place from /root/reference)
running on top of `vacmap_amd.aligner` — the `vacmap_index`-shaped shim over libvacmapx's C-ABI (vm_map, vm_k_cigar, vm_edit_distance) —
instead of on the oracle's primitives. Here the library is the product's kernel sources compiled against the CPU fiber emulator
(tests/emu; there is no GPU in this container); on a GPU box the same shim binds the real libvacmapx.so, but the reference cannot
travel there.

    python tools/harness/dropin_run.py [case ...]          default: case A = the reference's testdata pair (README.md:124)

For every read of the golden case: the 9-tuples the reference produces through the shim must equal the golden V6 records (which were
produced by the reference on top of the oracle). Nothing of the reference is copied or shipped.
"""
import json, os, sys, types
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests')); sys.path.insert(0, _HERE)


def run(cases=('A',), max_reads=None, log=sys.stdout):
    import emu_lib
    import refload
    import refrun
    from vacmap_amd import aligner as AL
    from vacmap_amd.lib import Index
    ctx = emu_lib.context()
    AL.use_context(ctx)
    calls = {'map': 0, 'k_cigar_global': 0, 'k_cigar_zdrop': 0, 'edlib': 0}

    # the module the reference imports as `vacmap_index` (alias mp): the shim's names, with call counters
    vmi = types.ModuleType('vacmap_index')

    class CountingAligner(AL.Aligner):
        def map(self, seq, check_num=100, mid_occ=-1):
            calls['map'] += 1
            return AL.Aligner.map(self, seq, check_num=check_num, mid_occ=mid_occ)

    def k_cigar(target, query, *a, **kw):
        zd = kw.get('zdropvalue', a[7] if len(a) > 7 else -1)
        calls['k_cigar_global' if zd < 0 else 'k_cigar_zdrop'] += 1
        return AL.k_cigar(target, query, *a, **kw)

    def edit(query, target):
        calls['edlib'] += 1
        return AL.edlib_align(query=query, target=target, task='distance')['editDistance']
    vmi.Aligner = CountingAligner; vmi.k_cigar = k_cigar
    meta = json.load(open(os.path.join(_ROOT, 'tests', 'golden', 'cases.json')))
    arrays = np.load(os.path.join(_ROOT, 'tests', 'golden', 'cases.npz'))
    total = same = 0
    for cid in cases:
        c = meta[cid]
        contigs = [arrays['%s_contig%d' % (cid, i)].tobytes().decode() for i in range(len(c['names']))]
        gi = Index.from_seqs(ctx, c['names'], contigs, k=c['k'], w=c['w'])
        al = CountingAligner(index=gi, ctx=ctx)
        m = refload.load(c['mode'], vmi)
        import edlib
        edlib.set_impl(edit)
        rc = refrun.RefContext.__new__(refrun.RefContext)
        rc.m = m; rc.mode = c['mode']; rc.al = al; rc.option = refrun.make_option(c['mode'])
        from numba.typed import Dict, List
        rc.contig2start = Dict(); rc.contig2seq = Dict(); rc.index2contig = List(); rc.contig2iloc = {}
        for i, item in enumerate(al.seq_offset):
            name = item[0].decode()
            rc.contig2start[name] = item[2]; rc.contig2seq[name] = al.seq(name).upper(); rc.index2contig.append(name); rc.contig2iloc[name] = i
        for ri, r in enumerate(c['reads'][:max_reads]):
            seq = arrays['%s_r%d_seq' % (cid, ri)].tobytes().decode()
            st, one = rc.align(r['name'], seq)
            got = [[t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), t[8]] for t in one]
            ok = st == r['v6_status'] and got == r['v6_records']
            total += 1; same += int(ok)
            log.write('case %s read %s: status %d, %d records %s  %s\n' % (cid, r['name'], st, len(got), [(g[0], g[1], g[2], g[3]) for g in got], 'IDENTICAL' if ok else 'DIFFERENT'))
    log.write('reads %d, identical to the golden records %d; calls through the C-ABI: %s\n' % (total, same, calls))
    return total, same, calls


if __name__ == '__main__':
    t, s, _ = run(tuple(sys.argv[1:]) or ('A',))
    sys.exit(0 if t == s else 1)
