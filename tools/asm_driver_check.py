#!/usr/bin/env python3
"""-mode asm end to end on the GPU box at a size the unit tests do not reach: a synthetic assembly (one contig long enough for more than 32 767
CIGAR operators, i.e. the CG-tag branch of --L) goes through `python -m vacmap_amd.driver -mode asm` (native emitter, vm_sam_opts.asm_mode) with and
without --L / --MD; the lines must equal those of vacmap_amd.sam.sam_lines(asm=True) on the ORACLE's records.
    python tools/asm_driver_check.py [--ref-mb 60] [--max-mb 8]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--ref-mb', type=int, default=60); ap.add_argument('--max-mb', type=float, default=8.0)
    a = ap.parse_args()
    from vacmap_amd import synth, sam
    import oracle_lib as O
    L = a.ref_mb * 1_000_000
    ref = synth.make_reference_fast([L // 2, L // 2], seed=21)
    rng = np.random.default_rng(22)
    contigs = []
    for i, (ci, ln) in enumerate(((0, int(a.max_mb * 1e6)), (1, 700_000), (0, 150_000), (1, 1_200_000))):
        st = int(rng.integers(0, len(ref[ci]) - ln - 1))
        ops = [(('INV', 'DEL', 'DUP')[k % 3], p, 1500) if k % 3 != 2 else ('DUP', p, 1500, 2) for k, p in enumerate(range(60_000, ln - 60_000, 300_000))]
        if i == 0:
            ops = []                              # no SV: one record over the whole contig, > 32 767 operators -> the CG-tag branch under --L
        seq = synth.mutate(synth.implant_svs(ref[ci][st:st + ln], ops), 0.004, rng)
        contigs.append(synth.tostr(synth.revcomp(seq) if i % 2 else seq))
    names = ['chrA', 'chrB']; refs = [synth.tostr(r) for r in ref]
    tmp = '/tmp/vmx_asm_check'; os.makedirs(tmp, exist_ok=True)
    with open(tmp + '/ref.fa', 'w') as f:
        for n, r in zip(names, refs):
            f.write('>%s\n%s\n' % (n, r))
    with open(tmp + '/asm.fa', 'w') as f:
        for i, c in enumerate(contigs):
            f.write('>ctg%d some comment\n%s\n' % (i, c))
    oi = O.Index.from_seqs(names, refs, k=15, w=10); op = O.params('asm')
    t = time.time(); orec = [O.align_asm(oi, c, op)[1] for c in contigs]; t_or = time.time() - t
    out = {'contigs': len(contigs), 'bases': sum(len(c) for c in contigs), 'oracle_s': round(t_or, 1), 'checks': []}
    for flags, kw in (([], {}), (['--L'], {'cigar2cg': True}), (['--MD'], {'md': True})):
        t = time.time()
        pr = subprocess.run([sys.executable, '-m', 'vacmap_amd.driver', '-ref', tmp + '/ref.fa', '-read', tmp + '/asm.fa', '-mode', 'asm', '-workdir', tmp + '/wd',
                             '-o', tmp + '/out.sam', '--nowriteindex', '--force'] + flags, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
        dt = time.time() - t
        got = [l.rstrip('\n') for l in open(tmp + '/out.sam') if not l.startswith('@')]
        want = []
        for i, c in enumerate(contigs):
            mine = [('ctg%d' % i, names[r[1]]) + tuple(r[2:]) for r in orec[i]]
            want += sam.sam_lines(mine, c, None, lambda cn, x, y: refs[names.index(cn)][x:y], rg_id='1', asm=True, markunbalancetra=False, **kw)
        nops = max(sum(ch.isalpha() or ch == '=' for ch in l.split('\t')[5]) for l in want) if want else 0
        cg = sum('\tCG:Z:' in l for l in got)
        out['checks'].append({'flags': flags, 'rc': pr.returncode, 'lines': len(got), 'identical': got == want, 'driver_s': round(dt, 1), 'max_cigar_ops_in_column': nops, 'lines_with_CG': cg})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
