#!/usr/bin/env python3
"""Search (build container only) for inputs that drive the reference's fix_simple_inv (mammap_clrnano.py:24226-24312) into its
`refen_0 < refst_1` branch (the left flank of a simple inversion ends before the inverted segment's reference start while the right
flank starts the same number of bases early). Prints the seeds / parameters that reach it; tools/harness/gen_golden.py case I uses them."""
import sys, os
import numpy as np
_HERE = os.path.dirname(os.path.abspath(__file__)); _ROOT = os.path.dirname(os.path.dirname(_HERE))
sys.path.insert(0, _ROOT); sys.path.insert(0, _HERE)
import refrun
from refrun import O
from vacmap_amd import synth


def probe(m, al, ctx):
    hits = []
    orig = m.fix_simple_inv

    def w(alignment_list, contig2start, contig2seq, testseq):
        al_ = alignment_list
        for iloc in range(max(0, len(al_) - 2)):
            if al_[iloc][0][2] == al_[iloc + 2][0][2] and al_[iloc][0][2] != al_[iloc + 1][0][2] and al_[iloc][0][2] == 1:
                c = m.pos2contig(al_[iloc][0][1], contig2start); b = contig2start[c]
                refen_0 = al_[iloc][-1][1] + al_[iloc][-1][3] - b; readen_0 = al_[iloc][-1][0] + al_[iloc][-1][3]
                refst_1 = al_[iloc + 1][-1][1] - b; readst_1 = al_[iloc + 1][0][0]
                refen_1 = al_[iloc + 1][0][1] + al_[iloc + 1][0][3] - b; readen_1 = al_[iloc + 1][-1][0] + al_[iloc + 1][-1][3]
                refst_2 = al_[iloc + 2][0][1] - b; readst_2 = al_[iloc + 2][0][0]
                if refst_2 - refen_0 == refen_1 - refst_1 and readst_1 - readen_0 + readst_2 - readen_1 == 0 and refst_1 - refen_0 != 0 and refst_1 - refen_0 + refst_2 - refen_1 == 0:
                    tr = contig2seq[c][refen_0:refst_1] if refen_0 < refst_1 else None
                    tq = testseq[readen_0:readen_0 - refen_0 + refst_1] if refen_0 < refst_1 else None
                    hits.append(('LT' if refen_0 < refst_1 else 'GT', refst_1 - refen_0, tr == tq if tr is not None else None))
        return orig(alignment_list, contig2start, contig2seq, testseq)
    m.fix_simple_inv = w
    return hits, orig


def main():
    ref = synth.make_reference([200000], seed=151)[0]
    found = []
    trial = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 460):
        rng = np.random.default_rng(seed)
        r = ref.copy()
        p = 60000 + int(rng.integers(0, 50000)); ml = int(rng.integers(900, 3000)); b = int(rng.integers(5, 30))
        kind = int(rng.integers(0, 4)); b = b + 8 if kind == 3 else b
        if kind == 0:      # arm before the left breakpoint mirrors the end of the inverted segment
            r[p - b:p] = synth.revcomp(r[p + ml - b:p + ml])
        elif kind == 1:    # arm after the right breakpoint mirrors the start of the inverted segment
            r[p + ml:p + ml + b] = synth.revcomp(r[p:p + b])
        elif kind == 2:    # both
            r[p - b:p] = synth.revcomp(r[p + ml - b:p + ml]); r[p + ml:p + ml + b] = synth.revcomp(r[p:p + b])
        else:              # the arm before the left breakpoint mirrors the bases AFTER the inverted segment: the minus-strand segment can claim it
            r[p - b:p] = synth.revcomp(r[p + ml:p + ml + b])
        don = np.concatenate([r[:p], synth.revcomp(r[p:p + ml]), r[p + ml:]])
        err = float(rng.choice([0.0, 0.003, 0.01]))
        read = synth.tostr(synth.mutate(don[p - 3500:p + ml + 3500], err, rng))
        ix = O.Index.from_seqs(['chrA'], [synth.tostr(r)], k=15, w=10)
        al = refrun.Aligner(oracle_index=ix)
        ctx = refrun.RefContext('H', al)
        hits, orig = probe(ctx.m, al, ctx)
        st, one = ctx.align('t', read)
        ctx.m.fix_simple_inv = orig
        print(seed, 'kind', kind, 'b', b, 'ml', ml, 'err', err, 'status', st, 'nrec', len(one), hits, flush=True)
        if any(h[0] == 'LT' for h in hits):
            found.append((seed, kind, b, ml, err, hits))
    print('FOUND', found)


if __name__ == '__main__':
    main()
