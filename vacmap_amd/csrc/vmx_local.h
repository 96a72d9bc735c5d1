// vmx_local.h — serial per-read pieces of the local stage (host+device): guide-chain preparation (L1) and the proximity
// filter of L2. Follows get_localmap_multi_all_forDP_inv_guide_list (/root/reference/src/vacmap/mammap_clrnano.py:28479-28589):
// merge_chain :28529-28569, drop_somechains :28482-28528, sort by 1/len :28574, chain budget :28576-28582.
#ifndef VMX_LOCAL_H
#define VMX_LOCAL_H
#include "vmx_kernels.h"
#ifndef __host__
#define __host__
#define __device__
#endif

#define VM_READ_CAPACITY_DEV (-20)
#define VM_READ_RAISED_DEV (-10)
#define VM_READ_BANDFALL_DEV (-23)   // k_local_seed_band hands the read to the general kernel k_local_seed (internal: never returned by vm_align_batch)
#define VMX_PREP_WS 9        // ints of scratch per path of a read that vmx_local_prep needs (no limit on the number of chains: mode S re-seeds them all)

// :23231 acceptance of a table hit at `refloc` given the two closest guide anchors
__host__ __device__ inline bool vmx_local_accept(long long refloc, long long ref1, long long ref2, long long interval, long long readgap) {
    long long refgap = refloc - ref1; if (refgap < 0) refgap = -refgap;
    long long diff = readgap - refgap; if (diff < 0) diff = -diff;
    return (diff < 500) || (ref1 + interval >= refloc && ref1 - interval <= refloc) || (ref2 + interval >= refloc && ref2 - interval <= refloc);
}

// L1. paths: return_path_list of decode_hit (primary first), each in descending read order, concatenated in `rows` with lengths `len`.
// out: the guide chains that will be re-seeded (in processing order), concatenated in out_rows (capacity = total anchors), lengths out_len;
// *n_used = how many are re-seeded (<= 5 H, <= 3 L, all S), *n_total = len(list after drop_somechains) (> 1 selects LC-mm).
// ws: VMX_PREP_WS * np ints of scratch. segs != nullptr: the rows are NOT copied; instead the (source start, length) pieces that make out_rows, in
// order, go to ws[7 np ..] / ws[8 np ..] and their number to *segs (k_local_prep copies them with the whole wavefront).
__host__ __device__ inline void vmx_local_prep(const vmx_anchor* rows, const int32_t* len, int np, int mode, vmx_anchor* out_rows,
                                               int32_t* out_len, int32_t* n_used, int32_t* n_total, int32_t* ws, int32_t* segs = nullptr) {
    int32_t* const seg_src = ws + 7 * np; int32_t* const seg_len = ws + 8 * np; int nseg = 0;
    if (mode == 3 || mode == 4) {
        // mode R (mammap_noprefercloser.py:23902-23914): every chain is re-seeded, in the order given, no merge / drop / cap
        // (-mode asm, mammap_asm.py:19714-19719: decode_hit returns the primary path only, and that one is re-seeded)
        int w = 0;
        for (int p = 0; p < np; ++p) {
            if (segs) { seg_src[nseg] = w; seg_len[nseg] = len[p]; ++nseg; w += len[p]; }
            else for (int t = 0; t < len[p]; ++t) { out_rows[w] = rows[w]; ++w; }
            out_len[p] = len[p];
        }
        *n_used = np; *n_total = np;
        if (segs) *segs = nseg;
        return;
    }
    int32_t* const start = ws;
    { int o = 0; for (int p = 0; p < np; ++p) { start[p] = o; o += len[p]; } }
    // a chain = ordered list of original paths (concatenation); next[] links, head/tail per chain
    int32_t* const head = ws + np; int32_t* const tail = ws + 2 * np; int32_t* const nxt = ws + 3 * np; int32_t* const clen = ws + 4 * np;
    int32_t* const chains = ws + 5 * np; int nc = 0;
    for (int p = 1; p < np; ++p) { head[p] = p; tail[p] = p; nxt[p] = -1; clen[p] = len[p]; chains[nc++] = p; }
    // chains.sort(key = start read position) stable
    for (int i = 1; i < nc; ++i) {
        int c = chains[i]; int key = rows[start[tail[c]] + len[tail[c]] - 1].q; int j = i - 1;
        while (j >= 0 && rows[start[tail[chains[j]]] + len[tail[chains[j]]] - 1].q > key) { chains[j + 1] = chains[j]; --j; }
        chains[j + 1] = c;
    }
    for (int iloc = 0; iloc + 1 < nc; ++iloc) {
        int jloc = iloc + 1;
        while (jloc < nc) {
            int ci = chains[iloc], cj = chains[jloc];
            const vmx_anchor ie = rows[start[head[ci]]];                               // chains[iloc][0]: end anchor (highest q)
            const vmx_anchor js = rows[start[tail[cj]] + len[tail[cj]] - 1];            // chains[jloc][-1]: start anchor (lowest q)
            bool merged = false;
            if (ie.q + ie.l <= js.q && ie.s == js.s) {
                long long readgap = (long long)js.q - ie.q - ie.l, refgap;
                if (ie.s == 1) refgap = (long long)js.r - ie.r - ie.l; else refgap = (long long)ie.r - js.r - js.l;
                long long d = readgap - refgap; if (d < 0) d = -d;
                if (d < 500) {
                    // chains[iloc] = concatenate((chains[jloc], chains[iloc]))
                    nxt[tail[cj]] = head[ci]; head[ci] = head[cj]; clen[ci] += clen[cj];
                    for (int t = jloc; t + 1 < nc; ++t) chains[t] = chains[t + 1];
                    --nc; merged = true;
                }
            }
            if (!merged) ++jloc;
        }
    }
    // chains.sort(key = len) stable
    for (int i = 1; i < nc; ++i) { int c = chains[i]; int j = i - 1; while (j >= 0 && clen[chains[j]] > clen[c]) { chains[j + 1] = chains[j]; --j; } chains[j + 1] = c; }
    // list = [primary] + chains ; primary is "chain 0"
    head[0] = 0; tail[0] = 0; nxt[0] = -1; clen[0] = len[0];
    int32_t* const lst = ws + 6 * np; int nl = 0;
    lst[nl++] = 0;
    // drop_somechains
    for (int ci = 0; ci < nc; ++ci) {
        const int c = chains[ci];
        double sc0 = 0, sc1 = 0, cc0 = 0, cc1 = 0;
        long long distance = 0x7fffffffffffffffLL;
        const vmx_anchor cfirst = rows[start[head[c]]];
        const vmx_anchor clast = rows[start[tail[c]] + len[tail[c]] - 1];
        // iterate the chain with a cursor (descending q)
        int cp = head[c], ci2 = 0;   // current path / index within it
        for (int t = 0; t < len[0]; ++t) {
            const vmx_anchor item = rows[start[0] + t];
            if (item.q >= clast.q && item.q <= cfirst.q) { if (item.s == 1) sc0 += 1; else sc1 += 1; }
            while (rows[start[cp] + ci2].q > item.q) {
                bool has_next = (ci2 + 1 < len[cp]) || (nxt[cp] >= 0);
                if (has_next) { if (ci2 + 1 < len[cp]) ++ci2; else { cp = nxt[cp]; ci2 = 0; } } else break;
            }
            long long d = (long long)item.r - rows[start[cp] + ci2].r; if (d < 0) d = -d;
            if (d < distance) distance = d;
        }
        for (int p = head[c]; p >= 0; p = nxt[p]) for (int t = 0; t < len[p]; ++t) { if (rows[start[p] + t].s == 1) cc0 += 1; else cc1 += 1; }
        bool keep;
        if (sc0 > sc1 && cc0 > cc1) keep = true; else if (sc0 < sc1 && cc0 < cc1) keep = true; else keep = false;
        if ((!keep && distance < 500) || (cfirst.q - clast.q) < 100) continue;
        lst[nl++] = c;
    }
    // sort by 1/len ascending = len descending, stable
    for (int i = 1; i < nl; ++i) { int c = lst[i]; int j = i - 1; while (j >= 0 && clen[lst[j]] < clen[c]) { lst[j + 1] = lst[j]; --j; } lst[j + 1] = c; }
    int budget = mode == 1 ? 3 : (mode == 2 ? nl : 5);   // :28581 / mammap_ccs.py:28581 / S: unlimited
    int used = nl < budget ? nl : budget;
    int w = 0;
    for (int gi = 0; gi < used; ++gi) {
        int c = lst[gi]; int n = 0;
        for (int p = head[c]; p >= 0; p = nxt[p]) {
            if (segs) { seg_src[nseg] = start[p]; seg_len[nseg] = len[p]; ++nseg; n += len[p]; w += len[p]; }
            else for (int t = 0; t < len[p]; ++t) { out_rows[w++] = rows[start[p] + t]; ++n; }
        }
        out_len[gi] = n;
    }
    *n_used = used; *n_total = nl;
    if (segs) *segs = nseg;
}

#endif
