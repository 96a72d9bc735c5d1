#!/usr/bin/env python3
"""-mode asm at a larger scale on the GPU box: a synthetic assembly (contigs of 0.2 - 6 Mb with SVs, both strands, 0.3 % divergence) against a
synthetic reference, every contig through vm_align_batch (VM_MODE_ASM: the per-read function below 500 kb, the batch-linked path above) and
through the CPU oracle; records must be identical. Prints one JSON line with the times.
    python tools/asm_bench.py [--ref-mb 60] [--contigs 8] [--max-mb 6] [--seed 1]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref-mb', type=int, default=60); ap.add_argument('--contigs', type=int, default=8); ap.add_argument('--max-mb', type=float, default=6.0)
    ap.add_argument('--seed', type=int, default=1); ap.add_argument('--no-oracle', action='store_true')
    a = ap.parse_args()
    from vacmap_amd import synth
    from vacmap_amd.driver import _keep_heap_pages
    _keep_heap_pages()                          # the driver's allocator setting (VMX_DRIVER_MALLOPT=0: off)
    from vacmap_amd.lib import Context, Index, align_batch
    import oracle_lib as O
    ctx = Context(0)
    L = a.ref_mb * 1_000_000
    ref = synth.make_reference_fast([L // 2, L // 2], seed=a.seed)
    rng = np.random.default_rng(a.seed + 1)
    contigs = []
    for i in range(a.contigs):
        ln = int(rng.uniform(0.2, a.max_mb) * 1_000_000) if i else int(a.max_mb * 1_000_000)
        ci = i % 2
        st = int(rng.integers(0, len(ref[ci]) - ln - 1))
        piece = ref[ci][st:st + ln]
        ops = []
        p = 50_000
        while p < ln - 60_000:
            kind = ('INV', 'DEL', 'DUP', 'INS')[int(rng.integers(0, 4))]
            sz = int(rng.integers(300, 4000))
            ops.append((kind, p, sz) if kind in ('INV', 'DEL') else ((kind, p, sz, 2) if kind == 'DUP' else (kind, p, sz, int(rng.integers(1, 100)))))
            p += int(rng.integers(150_000, 400_000))
        donor = synth.implant_svs(piece, ops)
        seq = synth.mutate(donor, 0.003, rng)
        if i % 3 == 2:
            seq = synth.revcomp(seq)
        contigs.append(synth.tostr(seq))
    names = ['chrA', 'chrB']
    refs = [synth.tostr(r) for r in ref]
    t = time.time(); gi = Index.from_seqs(ctx, names, refs, k=15, w=10); t_index = time.time() - t
    prm = ctx.lib.params('asm')
    t = time.time(); status, recs, stats = align_batch(ctx, gi, prm, contigs); t_dev = time.time() - t
    out = {'contigs': len(contigs), 'contig_bases': sum(len(c) for c in contigs), 'longest': max(len(c) for c in contigs), 'ref_bases': L,
           'device_s': round(t_dev, 2), 'index_build_s': round(t_index, 2), 'records': len(recs), 'status': [int(s) for s in status],
           'device_Mbp_per_s': round(sum(len(c) for c in contigs) / t_dev / 1e6, 2)}
    if not a.no_oracle:
        oi = O.Index.from_seqs(names, refs, k=15, w=10)
        oprm = O.params('asm')
        t = time.time(); same = 0
        for x, c in enumerate(contigs):
            ost, orecs = O.align_asm(oi, c, oprm)
            mine = [r[1:] for r in recs if r[0] == x]
            ok = (ost == 0) == (status[x] == 0) and mine == [r[1:] for r in orecs]
            same += ok
        out['oracle_s_1_thread'] = round(time.time() - t, 2); out['contigs_identical_to_oracle'] = same
    print(json.dumps(out))


if __name__ == '__main__':
    main()
