// vmx_select.h — chain peeling, primary/MAPQ and secondary selection (serial per read; host+device).
// Follows hit2work_1 (/root/reference/src/vacmap/mammap_clrnano.py:23588-23707, nested select_secondary_alignment
// :23505-23538) and decode_hit (:23981-24020). np.argsort is taken as stable.
// Mode deltas: accept threshold 60 (H) / 40 (L,S,R) (:23650, mammap_ccs.py:23649); secondary min span 50 / 100 (R).
#ifndef VMX_SELECT_H
#define VMX_SELECT_H
#include "vmx_kernels.h"
#include <math.h>

#ifndef __host__
#define __host__
#define __device__
#endif

struct vmx_select_out { int mapq; double score; int n_paths; };

__host__ __device__ inline int64_t vmx_select_scratch_bytes(int64_t n) { return 8 * n + 4 * (8 * n + 12) + n + 64; }

// |A ∩ B| of two strictly descending int lists
__host__ __device__ inline int vmx_desc_intersect(const int* a, int na, const int* b, int nb) {
    int i = 0, j = 0, c = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) { ++c; ++i; ++j; }
        else if (a[i] > b[j]) ++i;
        else ++j;
    }
    return c;
}

// always inlined on the device: the kernel calls it on LDS-staged and on HBM arrays, and each inlined copy gets plain ds_ / global_
// accesses where a shared out-of-line body would have to use flat ones
#ifdef __HIPCC__
#define VMX_SELECT_INLINE __attribute__((always_inline)) inline
#else
#define VMX_SELECT_INLINE inline
#endif
// The function is cut into the pieces k_chain_select (k_chain.hip) runs: the serial peel on lane 0 over LDS-staged S / P / S_arg / used,
// a wave-parallel gather of the chain nodes' read positions, the serial ranking (order, read bins, primaries, MAPQ, secondaries) on
// small arrays, and a wave-parallel copy of the selected paths. Scratch of a read in HBM (vmx_select_scratch_bytes):
// cscore[n] f64 | cidx[n] | coff[n+1] | order[n] | bins[n] | boff[n+1] | prim[n] | sec[n] | cq[n] (+ used[n] bytes for the HBM tier).
struct vmx_select_scr { double* cscore; int *cidx, *coff, *order, *bins, *boff, *prim, *sec, *cq; unsigned char* used; };
__host__ __device__ inline vmx_select_scr vmx_select_scratch(char* scratch, int n) {
    vmx_select_scr r;
    r.cscore = (double*)scratch; r.cidx = (int*)(r.cscore + n); r.coff = r.cidx + n; r.order = r.coff + n + 1; r.bins = r.order + n;
    r.boff = r.bins + n; r.prim = r.boff + n + 1; r.sec = r.prim + n; r.cq = r.sec + n; r.used = (unsigned char*)(r.cq + n + 2);
    return r;
}

// hit2work_1 :23588-23640: best chain first, then every other chain end in descending-S order (S_arg), a walk stops at the first anchor
// an earlier chain used and its score is S[end] - S[that anchor]; chains scoring <= 40 are dropped. Returns the number of chains kept,
// or -1 when the read is unmapped (best chain <= 40, or not above the mode's accept threshold :23650). `used`: n zeroed flags.
__host__ __device__ VMX_SELECT_INLINE int vmx_select_peel(int n, const double* S, const int32_t* P, const int32_t* SA, int gmax, int mode, unsigned char* used,
                                                          double* cscore, int* coff, int* cidx) {
    const double accept = (mode == 0) ? 60.0 : 40.0;
    int nch = 0, w = 0;
    bool hit = false;
    const double scores = S[gmax];
    {
        int take = gmax; used[take] = 1; double score = S[take]; int start = w;
        while (true) { cidx[w++] = take; if (P[take] == VMX_NOPRE) break; take = P[take]; used[take] = 1; }
        if (score > 40) { hit = true; cscore[nch] = score; coff[nch] = start; ++nch; } else w = start;
    }
    const double max_scores = scores > 0 ? scores : 0;
    if (!hit) return -1;                                 // hit == False -> unmapped whatever follows
    for (int x = n - 1; x >= 0; --x) {
        int take = SA[x];
        if (used[take]) continue;
        int start = w; used[take] = 1; double score = S[take];
        while (true) {
            cidx[w++] = take;
            if (P[take] == VMX_NOPRE) break;
            take = P[take];
            if (used[take]) { score = score - S[take]; break; }
            used[take] = 1;
        }
        if (score > 40) { cscore[nch] = score; coff[nch] = start; ++nch; } else w = start;
    }
    coff[nch] = w;
    if (!(max_scores > accept)) return -1;
    return nch;
}

// :23650-23707 + select_secondary_alignment :23505-23538 on the kept chains: cq[t] = read position of chain node t (A[cidx[t]].q).
// Sg / cidx: S and the chain node list in HBM (read a few times by the secondary test). Returns the number of secondaries (sec[]).
// mode 4 (-mode asm, mammap_asm.py:18551-18602 + decode_hit :21280-21348): the chains keep their score order (no "best chain first" swap), the
// primary is order[0] (*out_prim), there are no secondaries; when MAPQ is 0 and the primary's group holds a second chain within 0.1 % of its
// score the reference picks the least divergent of them with edlib (:21302-21326) — that choice is not made here: returns -2, the kernel
// marks the contig (n_paths = -7) and the host makes the choice with the device's edit-distance kernels (vmx_asm_resolve_ties, vmx_asm.hip).
__host__ __device__ VMX_SELECT_INLINE int vmx_select_rank(int nch, int mode, const double* cscore, const int* coff, const int* cq, const double* Sg, const int* cidx,
                                                          int* order, int* bins, int* boff, int* prim, int* sec, int* out_mapq, int* out_prim, bool have_order_bins = false) {
    const int sec_min_span = (mode == 3) ? 100 : 50;
    const bool asmv = mode == 4;
    // order = argsort(scores)[::-1] (stable): descending score, equal scores in descending index
    // (have_order_bins: k_chain_select has filled order / bins / boff with the whole wavefront — vmx_select_order_bins_wave, k_chain.hip)
    if (!have_order_bins) for (int c = 0; c < nch; ++c) {
        int pos = 0;
        while (pos < c && cscore[order[pos]] > cscore[c]) ++pos;
        for (int t = c; t > pos; --t) order[t] = order[t - 1];
        order[pos] = c;
    }
    if (!asmv && order[0] != 0) { for (int i = 0; i < nch; ++i) if (order[i] == 0) { order[i] = order[0]; order[0] = 0; break; } }
    const int pc0 = order[0];
    *out_prim = pc0;
    // read-position bins (//100) per chain, unique, descending
    if (!have_order_bins) {
        int bw = 0;
        for (int c = 0; c < nch; ++c) {
            boff[c] = bw; int last = -1;
            for (int t = coff[c]; t < coff[c + 1]; ++t) { int b = cq[t] / 100; if (b != last) { bins[bw++] = b; last = b; } }
        }
        boff[nch] = bw;
    }
    int np = 0; prim[np++] = order[0];
    double f2 = 0.0;
    for (int oi = 1; oi < nch; ++oi) {
        int c = order[oi]; int lc = boff[c + 1] - boff[c];
        double maxov = 0.0; int prefer = 0;
        for (int p = 0; p < np; ++p) {
            int pc = prim[p]; int lp = boff[pc + 1] - boff[pc];
            int inter = vmx_desc_intersect(bins + boff[c], lc, bins + boff[pc], lp);
            double ov = (double)inter / (double)(lc < lp ? lc : lp);
            if (ov > maxov) { maxov = ov; prefer = p; }
        }
        if (maxov < 0.5) prim[np++] = c;
        else if (prefer == 0) { f2 = cscore[c]; break; }   // = primary_scores_List[0][1]; later members never matter
    }
    {
        const double f1 = cscore[pc0];
        const double mlen = (double)(coff[pc0 + 1] - coff[pc0]);
        double v = 40 * (1 - f2 / f1);
        double mm = mlen / 10; if (mm > 1.0) mm = 1.0;
        v = v * mm;
        v = v * log(f1);
        long long iv = (long long)v;
        *out_mapq = (int)(iv < 60 ? iv : 60);
        if (asmv) return (*out_mapq == 0 && f2 != 0.0 && !(f2 / f1 < 0.999)) ? -2 : 0;
    }
    int nsec = 0;
    if (nch > 1) {
        const int b0 = coff[0], b1 = coff[1];   // best path, descending q
        for (int oi = 1; oi < nch; ++oi) {
            int c = order[oi];
            int en = cq[coff[c]], st = cq[coff[c + 1] - 1];
            if (en - st < sec_min_span) continue;
            double v_en = 0.0, v_st = 0.0;
            for (int pass = 0; pass < 2; ++pass) {   // loc2score[x] = S of the best-path anchor with the largest q <= x (0 if none)
                int x = pass == 0 ? en : st;
                int lo = b0, hi = b1;   // first t in [b0,b1) with q <= x (q descending)
                while (lo < hi) { int mid = (lo + hi) >> 1; if (cq[mid] <= x) hi = mid; else lo = mid + 1; }
                double v = lo < b1 ? Sg[cidx[lo]] : 0.0;
                if (pass == 0) v_en = v; else v_st = v;
            }
            double f1s = v_en - v_st; if (f1s < 1.0) f1s = 1.0;
            double f2s = cscore[c];
            double df = f1s - f2s; if (df < 0) df = -df;
            if (f2s / f1s > 0.9 || df < 40) {
                bool skip = false;
                for (int s2 = 0; s2 < nsec; ++s2) {
                    int pc = sec[s2];
                    int pen = cq[coff[pc]], pst = cq[coff[pc + 1] - 1];
                    int lo = pst > st ? pst : st, hi = en < pen ? en : pen;
                    int ov = hi - lo; if (ov < 0) ov = 0;
                    if (((double)ov / (double)(en - st)) > 0.5) { skip = true; break; }
                }
                if (!skip) sec[nsec++] = c;
            }
        }
    }
    return nsec;
}

#endif
