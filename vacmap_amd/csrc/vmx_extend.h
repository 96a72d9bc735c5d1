// vmx_extend.h — serial per-read "segment surgery" of the extend stage (host+device, one thread per read).
// Array-based restatement of the reference's list manipulations in /root/reference/src/vacmap/mammap_clrnano.py:
//   rebuild_chain_break :23437-23484 (E1)    get_query_target_for_cigar :5802-5818     extend_edge_test :2302-2525 (E3, problem setup/apply)
//   drop_misplaced_alignment_test :726-787   getdupiloc_numba :16680-16734             merge_conjacent_alignment :16736-16780
//   fix_simple_inv :24226-24312              split_alignment_test :21505-21617 (E5 checkpoints)   get_onemapinfolist :20731-20838 (E6)
//   pairedindel :5604-5650
// The DP itself (edit distance, x-drop extension, gap fill) runs in the wave-parallel kernels of k_dp.hip on strings gathered
// by k_gather from these descriptors.
#ifndef VMX_EXTEND_H
#define VMX_EXTEND_H
#include "vmx_kernels.h"
#include "vmx_local.h"
// a pool of the EXTEND stage was too small for this read: the batch is run again with larger pools (vmx_align.hip, align_device); a read that still
// carries the code when the retries are used up is reported as VM_READ_CAPACITY like a local-stage one
#define VMX_EXT_CAPACITY_DEV (-24)
// a segment of this read reached the last (exact, unbanded) tier of the divergence filter in a batch that does not launch it: the read is run again alone with the
// exact tier (vmx_align.hip, align_device); never leaves the library
#define VMX_EXT_NEED_EXACT_DEV (-25)
static_assert(VMX_EXT_NEED_EXACT_DEV != VM_READ_FASTPATH_DEV && VMX_EXT_NEED_EXACT_DEV != VM_READ_CAPACITY_DEV && VMX_EXT_NEED_EXACT_DEV != VM_READ_RAISED_DEV && VMX_EXT_NEED_EXACT_DEV != -22, "distinct device codes");
static_assert(VMX_EXT_CAPACITY_DEV != VM_READ_FASTPATH_DEV && VMX_EXT_CAPACITY_DEV != VM_READ_CAPACITY_DEV && VMX_EXT_CAPACITY_DEV != VM_READ_RAISED_DEV && VMX_EXT_CAPACITY_DEV != -22,
              "the internal per-read device codes must be distinct (E.status is seeded from the local stage's status)");
#ifndef __host__
#define __host__
#define __device__
#endif

// string descriptor: bases [start, start+len) of the oriented read (src 0) or of the reference (src 1, global coords)
// op: 0 plain, 1 reversed, 2 complemented (same order), 3 reverse-complemented
struct vmx_sdesc { int64_t start; int32_t len; int8_t src; int8_t op; int16_t pad; };
struct vmx_pair_desc { vmx_sdesc t, q; };

struct vmx_ref_view {
    const uint8_t* codes; const int64_t* coff; int nseq;
    // the contig of the last look-up: consecutive anchors of a read sit on one contig, so nearly every vmx_p2c is answered by two compares
    // on registers instead of a five-step bisection of dependent loads (the segment walks make two look-ups per anchor)
    mutable int cc = -1; mutable long long clo = 1, chi = 0;
};

// per-read segment list: segment s = A[st[s] .. en[s]) ; every segment keeps one spare slot before and after it
struct vmx_segs { vmx_anchor* A; int32_t* st; int32_t* en; int32_t nseg; int32_t capA; int32_t capS; };

// pos2contig (:51-59): the last contig whose start is <= pos (0 when pos lies before the first). The reference scans the contig
// starts linearly; a bisection gives the same index with 5 instead of 24 dependent loads on an hg38-size contig table.
__host__ __device__ inline int vmx_p2c(const vmx_ref_view& R, long long pos) {
    if (pos >= R.clo && pos < R.chi) return R.cc;
    int lo = 0, hi = R.nseq;                      // invariant: coff[lo] <= pos (or lo == 0), coff[hi] > pos (or hi == nseq)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (R.coff[mid] <= pos) lo = mid; else hi = mid; }
    // remember the interval on which this answer holds: [coff[lo], coff[lo + 1]) — for lo = 0 everything below coff[1], for the last contig everything above
    R.cc = lo; R.clo = lo == 0 ? -(1LL << 62) : R.coff[lo]; R.chi = lo + 1 >= R.nseq ? (1LL << 62) : R.coff[lo + 1];
    return lo;
}
__host__ __device__ inline long long vmx_clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (v > hi ? hi : v); }
__host__ __device__ inline vmx_anchor vmx_mk(long long q, long long r, int s, int l) { vmx_anchor a; a.q = (int32_t)q; a.r = r; a.s = (int16_t)s; a.l = (int16_t)l; return a; }
#define VMX_PF 8                 // anchors fetched ahead by the serial segment walks
#define SEG_FIRST(S, s) ((S).A[(S).st[s]])
#define SEG_LAST(S, s) ((S).A[(S).en[s] - 1])
#define SEG_LEN(S, s) ((S).en[s] - (S).st[s])

// remove segment s (list.pop(s)); anchors stay where they are
__host__ __device__ inline void vmx_seg_erase(vmx_segs& S, int s) {
    for (int t = s; t + 1 < S.nseg; ++t) { S.st[t] = S.st[t + 1]; S.en[t] = S.en[t + 1]; }
    --S.nseg;
}

// E1 :23437-23484. chain in ASCENDING read order. returns 0, VM_READ_RAISED_DEV (IndexError at :23480) or VMX_EXT_CAPACITY_DEV
// asmv: the -mode asm fork (mammap_asm.py:13256-13295) joins only when refgap >= 0 (no tolerance for a 20-base back-step)
__host__ __device__ inline int vmx_rebuild_chain_break(const vmx_anchor* chain_desc, int n, const vmx_ref_view& R, int large_cost, int small_alignment, vmx_segs& S, bool asmv = false) {
    S.nseg = 0;
    int w = 1;                                   // write cursor in A (slot 0 = spare before the first segment)
    vmx_anchor pre = chain_desc[n - 1];
    if (S.capS < 1 || S.capA < 4) return VMX_EXT_CAPACITY_DEV;
    S.st[0] = w; S.A[w++] = pre; S.en[0] = w; S.nseg = 1;
    // the chain is walked serially (every step depends on the anchor kept before it), but the LOADS do not: VMX_PF anchors are fetched
    // ahead so that their HBM latencies overlap instead of adding up (one lane per read: nothing else hides them)
    for (int x0 = 1; x0 < n; x0 += VMX_PF) {
      vmx_anchor pf[VMX_PF];
#pragma unroll
      for (int j = 0; j < VMX_PF; ++j) pf[j] = chain_desc[x0 + j < n ? n - 1 - (x0 + j) : 0];
#pragma unroll
      for (int j = 0; j < VMX_PF; ++j) {
        if (x0 + j >= n) break;
        const vmx_anchor now = pf[j];
        if (pre.s == now.s) {
            long long readgap = (long long)now.q - pre.q - pre.l, refgap;
            if (pre.s == 1) refgap = (long long)now.r - pre.r - pre.l; else refgap = (long long)pre.r - now.r - now.l;
            long long d = readgap - refgap; if (d < 0) d = -d;
            if (d <= large_cost && refgap >= (asmv ? 0 : -20) && readgap < 100) {
                if (vmx_p2c(R, pre.r) == vmx_p2c(R, now.r)) {
                    if (refgap >= 0) { if (w + 2 > S.capA) return VMX_EXT_CAPACITY_DEV; S.A[w++] = now; S.en[S.nseg - 1] = w; pre = now; continue; }
                    else { if (readgap <= 20) continue; if (w + 2 > S.capA) return VMX_EXT_CAPACITY_DEV; S.A[w++] = now; S.en[S.nseg - 1] = w; pre = now; continue; }
                }
            }
        }
        if (SEG_LEN(S, S.nseg - 1) == 1) { w = S.st[S.nseg - 1]; --S.nseg; }
        if (S.nseg > 0) {
            const int s = S.nseg - 1;
            if ((SEG_LAST(S, s).q + SEG_LAST(S, s).l - SEG_FIRST(S, s).q) < small_alignment) { w = S.st[s]; --S.nseg; }
        }
        // new segment: spare slot after the previous one and before this one
        if (S.nseg > 0) w = S.en[S.nseg - 1] + 2; else w = 1;
        if (S.nseg + 1 > S.capS || w + 2 > S.capA) return VMX_EXT_CAPACITY_DEV;
        S.st[S.nseg] = w; S.A[w++] = now; S.en[S.nseg] = w; ++S.nseg;
        pre = now;
      }
    }
    if (SEG_LEN(S, S.nseg - 1) == 1) --S.nseg;
    if (S.nseg == 0) return VM_READ_RAISED_DEV;
    { const int s = S.nseg - 1; if ((SEG_LAST(S, s).q + SEG_LAST(S, s).l - SEG_FIRST(S, s).q) < small_alignment) --S.nseg; }
    return 0;
}

// get_query_target_for_cigar :5802-5818 as descriptors (python slices clipped to the sequence)
__host__ __device__ inline void vmx_qt_for_cigar(const vmx_anchor& pre, const vmx_anchor& now, long long L, const vmx_ref_view& R, vmx_pair_desc* d) {
    if (pre.s == 1) {
        int c = vmx_p2c(R, pre.r); long long bias = R.coff[c], clen = R.coff[c + 1] - bias;
        long long a = vmx_clampll(pre.q, 0, L), b = vmx_clampll(now.q, 0, L);
        d->q.src = 0; d->q.op = 0; d->q.start = a; d->q.len = (int32_t)(b > a ? b - a : 0);
        long long ta = vmx_clampll(pre.r - bias, 0, clen), tb = vmx_clampll(now.r - bias, 0, clen);
        d->t.src = 1; d->t.op = 0; d->t.start = bias + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);
    } else {
        int c = vmx_p2c(R, now.r); long long bias = R.coff[c], clen = R.coff[c + 1] - bias;
        // rc[L - now.q : L - pre.q] == revcomp(read[pre.q : now.q])
        long long a = vmx_clampll(pre.q, 0, L), b = vmx_clampll(now.q, 0, L);
        d->q.src = 0; d->q.op = 3; d->q.start = a; d->q.len = (int32_t)(b > a ? b - a : 0);
        long long ta = vmx_clampll(now.r + now.l - bias, 0, clen), tb = vmx_clampll(pre.r + pre.l - bias, 0, clen);
        d->t.src = 1; d->t.op = 0; d->t.start = bias + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);
    }
}

// ---- E3 extend_edge_test :2302-2525, split in problem setup and result application (san = 1).
// side 1 = right ends (independent of each other), side 0 = left ends (need the already extended right end of the previous segment).
// returns 1 and fills d when a DP call is made; 0 when the reference makes none (it may still rewrite the end anchor, done here).
__host__ __device__ inline int vmx_ext_setup(vmx_segs& S, int idx, int side, long long L, const vmx_ref_view& R, vmx_pair_desc* d) {
    const long long max_extend_size = 20000;
    if (side == 0) {
        vmx_anchor& first = SEG_FIRST(S, idx);
        if (first.q > 0) {
            long long looksize;
            if (idx == 0) looksize = first.q; else looksize = (long long)first.q - (SEG_LAST(S, idx - 1).q + SEG_LAST(S, idx - 1).l);
            const vmx_anchor pre = first;
            int c = vmx_p2c(R, pre.r); long long cst = R.coff[c], clen = R.coff[c + 1] - cst;
            if (pre.s == 1) {
                long long target_st = pre.r, query_st = pre.q;
                if (target_st - cst < looksize) looksize = target_st - cst;
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize == 0) return 0;
                long long qa = query_st - looksize; if (qa < 0) qa = 0;
                long long qlen = query_st > qa ? query_st - qa : 0;               // looksize < 0 -> empty slices
                d->q.src = 0; d->q.op = 1; d->q.start = qa; d->q.len = (int32_t)qlen;
                long long ta = vmx_clampll(target_st - cst - qlen, 0, clen), tb = vmx_clampll(target_st - cst, 0, clen);
                d->t.src = 1; d->t.op = 1; d->t.start = cst + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);
                return 1;
            } else {
                long long target_en = pre.r + pre.l, query_st = pre.q;
                long long lim = cst + clen - (target_en - 1); if (lim < looksize) looksize = lim;
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize == 0) return 0;
                long long qa = query_st - looksize; if (qa < 0) qa = 0;
                long long qlen = query_st > qa ? query_st - qa : 0;
                d->q.src = 0; d->q.op = 1; d->q.start = qa; d->q.len = (int32_t)qlen;
                long long ta = vmx_clampll(target_en - cst, 0, clen), tb = vmx_clampll(target_en + qlen - cst, 0, clen);
                d->t.src = 1; d->t.op = 2; d->t.start = cst + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);   // reversed(revcomp(x)) = complement(x)
                return 1;
            }
        } else {
            const vmx_anchor t = first;
            if (t.s == 1) first = vmx_mk(t.q, t.r, 1, 0); else first = vmx_mk(t.q, t.r + t.l, -1, 0);
            return 0;
        }
    } else {
        vmx_anchor& last = SEG_LAST(S, idx);
        if ((long long)last.q + last.l < L) {
            long long looksize;
            if (idx + 1 == S.nseg) looksize = L - ((long long)last.q + last.l); else looksize = (long long)SEG_FIRST(S, idx + 1).q - ((long long)last.q + last.l);
            const vmx_anchor pre = S.A[S.en[idx] - 2];
            const vmx_anchor now = last;
            int c = vmx_p2c(R, pre.r); long long cst = R.coff[c], clen = R.coff[c + 1] - cst;
            if (pre.s == 1) {
                long long target_en = now.r + now.l, query_en = (long long)now.q + now.l;
                long long lim = cst + clen - (target_en - 1); if (lim < looksize) looksize = lim;
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize == 0) return 0;
                long long qb = vmx_clampll(query_en + looksize, 0, L); long long qlen = qb > query_en ? qb - query_en : 0;
                d->q.src = 0; d->q.op = 0; d->q.start = query_en; d->q.len = (int32_t)qlen;
                long long ta = vmx_clampll(target_en - cst, 0, clen), tb = vmx_clampll(target_en + qlen - cst, 0, clen);
                d->t.src = 1; d->t.op = 0; d->t.start = cst + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);
                return 1;
            } else {
                long long target_st = now.r, query_en = (long long)now.q + now.l;
                if (target_st - cst < looksize) looksize = target_st - cst;
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize == 0) return 0;
                long long qb = vmx_clampll(query_en + looksize, 0, L); long long qlen = qb > query_en ? qb - query_en : 0;
                d->q.src = 0; d->q.op = 0; d->q.start = query_en; d->q.len = (int32_t)qlen;
                long long ta = vmx_clampll(target_st - cst - qlen, 0, clen), tb = vmx_clampll(target_st - cst, 0, clen);
                d->t.src = 1; d->t.op = 3; d->t.start = cst + ta; d->t.len = (int32_t)(tb > ta ? tb - ta : 0);
                return 1;
            }
        } else {
            const vmx_anchor t = last;
            if (t.s == 1) last = vmx_mk((long long)t.q + t.l, t.r + t.l, 1, 0); else last = vmx_mk((long long)t.q + t.l, t.r, -1, 0);
            return 0;
        }
    }
}
__host__ __device__ inline void vmx_ext_apply(vmx_segs& S, int idx, int side, int t_e, int q_e) {
    if (side == 0) {
        const vmx_anchor pre = SEG_FIRST(S, idx);
        if (pre.s == 1) SEG_FIRST(S, idx) = vmx_mk((long long)pre.q - q_e, pre.r - t_e, 1, 0);
        else SEG_FIRST(S, idx) = vmx_mk((long long)pre.q - q_e, pre.r + pre.l + t_e, -1, 0);
    } else {
        const vmx_anchor pre = S.A[S.en[idx] - 2];   // strand of the segment (:2454)
        const vmx_anchor now = SEG_LAST(S, idx);
        if (pre.s == 1) SEG_LAST(S, idx) = vmx_mk((long long)now.q + now.l + q_e, now.r + now.l + t_e, 1, 0);
        else SEG_LAST(S, idx) = vmx_mk((long long)now.q + now.l + q_e, now.r - t_e, -1, 0);
    }
}

// :726-787
__host__ __device__ inline bool vmx_drop_misplaced(vmx_segs& S, int iloc) {
    if (SEG_FIRST(S, iloc).s == SEG_FIRST(S, iloc + 1).s && SEG_FIRST(S, iloc).s == SEG_FIRST(S, iloc + 2).s) {
        long long mid_size = (long long)SEG_LAST(S, iloc + 1).q + SEG_LAST(S, iloc + 1).l - SEG_FIRST(S, iloc + 1).q;
        if (mid_size > 1000) return false;
        vmx_anchor pre = SEG_LAST(S, iloc), now = SEG_FIRST(S, iloc + 1);
        long long readgap = (long long)now.q - pre.q - pre.l, refgap;
        if (pre.s == 1) refgap = (long long)now.r - pre.r - pre.l; else refgap = (long long)pre.r - now.r - now.l;
        long long ar = refgap < 0 ? -refgap : refgap;
        if (ar < 100000) {
            int DEL = 0, INS = 0;
            if ((readgap - refgap) < -30) DEL += 1; else if ((readgap - refgap) > 30) INS += 1; else return false;
            long long gap_1 = readgap - refgap; if (gap_1 < 0) gap_1 = -gap_1;
            pre = SEG_LAST(S, iloc + 1); now = SEG_FIRST(S, iloc + 2);
            readgap = (long long)now.q - pre.q - pre.l;
            if (pre.s == 1) refgap = (long long)now.r - pre.r - pre.l; else refgap = (long long)pre.r - now.r - now.l;
            ar = refgap < 0 ? -refgap : refgap;
            if (ar < 100000) {
                if ((readgap - refgap) < -30) DEL += 1; else if ((readgap - refgap) > 30) INS += 1; else return false;
                long long gap_2 = readgap - refgap; if (gap_2 < 0) gap_2 = -gap_2;
                long long gm = gap_1 > gap_2 ? gap_1 : gap_2;
                if (DEL == 1 && INS == 1 && (mid_size < 500 || ((double)gm / (double)mid_size) > 0.5)) { vmx_seg_erase(S, iloc + 1); return true; }
            }
        }
    }
    return false;
}

// :16680-16734 + :16736-16780 (Q7 preserved: the strand field is added at :16705). dup: scratch of capS ints
__host__ __device__ inline void vmx_merge_conjacent(vmx_segs& S, const vmx_ref_view& R, int32_t* dup) {
    if (S.nseg < 2) return;
    int ndup = 0;
    {
        int iloc = 0;
        while (iloc + 1 < S.nseg) {
            long long readpos_1 = (long long)SEG_LAST(S, iloc).q + SEG_LAST(S, iloc).l;
            long long refpos_1; int strand_1;
            if (SEG_LAST(S, iloc).s == 1) { refpos_1 = SEG_LAST(S, iloc).r + SEG_LAST(S, iloc).l; strand_1 = 1; } else { refpos_1 = SEG_LAST(S, iloc).r; strand_1 = -1; }
            int jloc = iloc; bool hit = false; long long dupsize = 0, readpos_2 = 0; int new_iloc = 0;
            while (jloc + 1 < S.nseg) {
                jloc += 1;
                long long refpos_2; int strand_2;
                if (SEG_LAST(S, jloc).s == 1) { refpos_2 = SEG_FIRST(S, jloc).r; strand_2 = 1; } else { refpos_2 = SEG_FIRST(S, jloc).r + SEG_FIRST(S, jloc).s; strand_2 = -1; }
                if (strand_1 != strand_2) continue;
                if (strand_1 == 1) { if ((refpos_2 - refpos_1) < 50) { new_iloc = jloc; dupsize = refpos_2 - refpos_1; readpos_2 = SEG_FIRST(S, jloc).q; hit = true; } }
                else { if ((refpos_1 - refpos_2) < 50) { new_iloc = jloc; dupsize = refpos_1 - refpos_2; readpos_2 = SEG_FIRST(S, jloc).q; hit = true; } }
            }
            if (hit) {
                long long readgap = readpos_2 - readpos_1;
                if (((iloc + 1) < new_iloc) || (((dupsize - readgap) < -30) && (readgap < 30))) for (int s = iloc; s < new_iloc; ++s) dup[ndup++] = s;
                iloc = new_iloc;
            } else iloc += 1;
        }
    }
    int iloc = 0;
    while (iloc + 1 < S.nseg) {
        bool isdup = false; for (int t = 0; t < ndup; ++t) if (dup[t] == iloc) { isdup = true; break; }
        if (isdup) { iloc += 1; continue; }
        const vmx_anchor pre = SEG_LAST(S, iloc), now = SEG_FIRST(S, iloc + 1);
        if (pre.s != now.s || vmx_p2c(R, pre.r) != vmx_p2c(R, now.r)) { iloc += 1; continue; }
        long long readgap = (long long)now.q - pre.q - pre.l, refgap;
        if (pre.s == 1) refgap = (long long)now.r - pre.r - pre.l; else refgap = (long long)pre.r - now.r - now.l;
        if (refgap < 0) { iloc += 1; continue; }
        long long mn = readgap < refgap ? readgap : refgap; long long d = readgap - refgap; if (d < 0) d = -d;
        if (mn < 50 && d < 10000) {
            // List_merge: append the anchors of segment iloc+1 right behind segment iloc (moves down; source is always ahead of the destination)
            int wpos = S.en[iloc];
            for (int t = S.st[iloc + 1]; t < S.en[iloc + 1]; ++t) S.A[wpos++] = S.A[t];
            S.en[iloc] = wpos;
            vmx_seg_erase(S, iloc + 1);
        } else iloc += 1;
    }
}

__host__ __device__ inline uint8_t vmx_read_base(const uint8_t* rd, long long L, long long i) { return (i >= 0 && i < L) ? rd[i] : 4; }

// :24226-24312. returns 0 or VM_READ_RAISED_DEV (assert / IndexError)
// rmode: mode R keeps an older body (mammap_noprefercloser.py:17155-17200) whose `refen_0 > refst_1` branch changes nothing
__host__ __device__ inline int vmx_fix_simple_inv(vmx_segs& S, const vmx_ref_view& R, const uint8_t* rd, long long L, bool rmode) {
    if (S.nseg <= 2) return 0;
    for (int iloc = 0; iloc + 2 < S.nseg; ++iloc) {
        if (!(SEG_FIRST(S, iloc).s == SEG_FIRST(S, iloc + 2).s && SEG_FIRST(S, iloc).s != SEG_FIRST(S, iloc + 1).s)) continue;
        if (SEG_FIRST(S, iloc).s != 1) continue;
        int c = vmx_p2c(R, SEG_FIRST(S, iloc).r); long long bias = R.coff[c], clen = R.coff[c + 1] - bias;
        const uint8_t* cs = R.codes + bias;
        long long refen_0 = SEG_LAST(S, iloc).r + SEG_LAST(S, iloc).l - bias, readen_0 = (long long)SEG_LAST(S, iloc).q + SEG_LAST(S, iloc).l;
        long long refst_1 = SEG_LAST(S, iloc + 1).r - bias, readst_1 = SEG_FIRST(S, iloc + 1).q;
        long long refen_1 = SEG_FIRST(S, iloc + 1).r + SEG_FIRST(S, iloc + 1).l - bias, readen_1 = (long long)SEG_LAST(S, iloc + 1).q + SEG_LAST(S, iloc + 1).l;
        long long refst_2 = SEG_FIRST(S, iloc + 2).r - bias, readst_2 = SEG_FIRST(S, iloc + 2).q;
        if (!(refst_2 - refen_0 == refen_1 - refst_1 && readst_1 - readen_0 + readst_2 - readen_1 == 0)) continue;
        if (!(refst_1 - refen_0 != 0 && refst_1 - refen_0 + refst_2 - refen_1 == 0)) continue;
        if (refen_0 > refst_1) {
            if (rmode) continue;
            // tempref = revcomp(ref[refen_1 : refen_1 + refen_0 - refst_1]) ; tempquery = read[readen_0 - refen_0 + refst_1 : readen_0]
            long long n = refen_0 - refst_1;
            long long ta = vmx_clampll(refen_1, 0, clen), tb = vmx_clampll(refen_1 + n, 0, clen);
            long long qa = readen_0 - n, qb = readen_0;
            if (qa < 0) { qa += L; if (qa < 0) qa = 0; }                        // python negative index
            qa = vmx_clampll(qa, 0, L); qb = vmx_clampll(qb, 0, L);
            long long tlen = tb > ta ? tb - ta : 0, qlen = qb > qa ? qb - qa : 0;
            bool eq = tlen == qlen;
            for (long long x = 0; eq && x < tlen; ++x) {
                uint8_t rc = cs[tb - 1 - x]; rc = rc < 4 ? 3 - rc : 4;          // get_rc maps everything else to 'N'
                uint8_t qc = rd[qa + x];
                if (rc != qc) eq = false;                                      // 'N' == 'N' compares equal like the strings do
            }
            if (eq) {
                long long b = refen_0 - refst_1;
                SEG_FIRST(S, iloc + 2) = vmx_mk(readst_2 - b, refst_2 - b + bias, 1, 0);
                const vmx_anchor ins = vmx_mk(readst_2 - b, refen_0 + bias, -1, 0);
                while (true) {
                    if (SEG_LEN(S, iloc + 1) == 0) return VM_READ_RAISED_DEV;
                    if (ins.q <= (SEG_LAST(S, iloc + 1).q + SEG_LAST(S, iloc + 1).l)) --S.en[iloc + 1]; else break;
                }
                S.A[S.en[iloc + 1]++] = ins;                                   // spare slot after the segment guarantees room
            }
        } else {
            long long n = refst_1 - refen_0;
            long long ta = vmx_clampll(refen_0, 0, clen), tb = vmx_clampll(refst_1, 0, clen);
            long long qa = vmx_clampll(readen_0, 0, L), qb = vmx_clampll(readen_0 + n, 0, L);
            long long tlen = tb > ta ? tb - ta : 0, qlen = qb > qa ? qb - qa : 0;
            bool eq = tlen == qlen;
            for (long long x = 0; eq && x < tlen; ++x) if (cs[ta + x] != rd[qa + x]) eq = false;
            if (eq) {
                SEG_LAST(S, iloc) = vmx_mk(readen_0 - refen_0 + refst_1, refst_1 + bias, 1, 0);
                const vmx_anchor ins = vmx_mk(readen_0 - refen_0 + refst_1, refen_1 + refen_0 - refst_1 + bias, -1, 0);
                while (true) {
                    if (SEG_LEN(S, iloc + 1) == 0) return VM_READ_RAISED_DEV;
                    if (ins.q >= SEG_FIRST(S, iloc + 1).q) ++S.st[iloc + 1]; else break;
                }
                S.A[--S.st[iloc + 1]] = ins;                                   // spare slot before the segment
            }
        }
    }
    return 0;
}

// E5 checkpoints :21505-21617. Emits the DP problems of segment s (in the order the reference computes them) into out[];
// converts the end anchors to zero length like the reference. returns the number of problems, or a negative status.
// out == nullptr: count only (the end-anchor conversions are idempotent, so a counting call followed by an emitting call is safe)
// asmv: mammap_asm.py:22197-22316 — the short-anchor / short-gap skip applies only while max(readgap, refgap) < 2000
__host__ __device__ inline int vmx_split_alignment(vmx_segs& S, int s, long long L, const vmx_ref_view& R, vmx_pair_desc* out, int cap, bool asmv = false) {
    const long long min_gap_forcigar = 200;
    int np = 0;
    const int st = S.st[s], en = S.en[s];
    vmx_pair_desc tmp;
    if (S.A[st].s == 1) {
        vmx_anchor& last = S.A[en - 1];
        if (last.l != 0) last = vmx_mk((long long)last.q + last.l, last.r + last.l, 1, 0);
        vmx_anchor pre = S.A[st];
        for (int i0 = st + 1; i0 < en; i0 += VMX_PF) {
          vmx_anchor pf[VMX_PF];                             // loads fetched ahead of the serial walk (see vmx_rebuild_chain_break)
#pragma unroll
          for (int j = 0; j < VMX_PF; ++j) pf[j] = S.A[i0 + j < en ? i0 + j : en - 1];
#pragma unroll
          for (int j = 0; j < VMX_PF; ++j) {
            const int i = i0 + j;
            if (i >= en) break;
            const vmx_anchor now = pf[j];
            long long readgap = (long long)now.q - pre.q - pre.l, refgap = (long long)now.r - pre.r - pre.l;
            long long mn = readgap < refgap ? readgap : refgap;
            const long long mx = readgap < refgap ? refgap : readgap;
            if ((!asmv || mx < 2000) && (now.l < 19 || mn < min_gap_forcigar) && i + 1 != en) continue;
            if (out && np >= cap) return VMX_EXT_CAPACITY_DEV;
            vmx_pair_desc* d = out ? &out[np] : &tmp;
            vmx_qt_for_cigar(pre, now, L, R, d);
            if (d->t.len <= 0 || d->q.len <= 0) return VM_READ_RAISED_DEV;    // "Failed to compute CIGAR" :21562
            ++np; pre = now;
          }
        }
    } else {
        if (S.A[st].l != 0) S.A[st] = vmx_mk(S.A[st].q, S.A[st].r + S.A[st].l, -1, 0);
        if (S.A[en - 1].l != 0) S.A[en - 1] = vmx_mk((long long)S.A[en - 1].q + S.A[en - 1].l, S.A[en - 1].r, -1, 0);
        vmx_anchor pre = S.A[en - 1];                      // alignment[::-1]
        for (int i0 = en - 2; i0 >= st; i0 -= VMX_PF) {
          vmx_anchor pf[VMX_PF];
#pragma unroll
          for (int j = 0; j < VMX_PF; ++j) pf[j] = S.A[i0 - j >= st ? i0 - j : st];
#pragma unroll
          for (int j = 0; j < VMX_PF; ++j) {
            const int i = i0 - j;
            if (i < st) break;
            const vmx_anchor now = pf[j];
            long long readgap = (long long)pre.q - now.q - now.l, refgap = (long long)now.r - pre.r - pre.l;
            long long mn = readgap < refgap ? readgap : refgap;
            const long long mx = readgap < refgap ? refgap : readgap;
            if ((!asmv || mx < 2000) && (now.l < 19 || mn < min_gap_forcigar) && i != st) continue;
            if (out && np >= cap) return VMX_EXT_CAPACITY_DEV;
            vmx_pair_desc* d = out ? &out[np] : &tmp;
            vmx_qt_for_cigar(now, pre, L, R, d);
            if (d->t.len <= 0 || d->q.len <= 0) return VM_READ_RAISED_DEV;
            ++np; pre = now;
          }
        }
    }
    if (np == 0) return VM_READ_RAISED_DEV;                // cigarlist[-1] == [] :21566
    return np;
}

__host__ __device__ inline int vmx_put_int(char* o, long long v) {
    char tmp[24]; int n = 0;
    if (v < 0) v = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) o[i] = tmp[n - 1 - i];
    return n;
}

// len(Cigar(s)): M, I, S, =, X
__host__ __device__ inline long long vmx_cigar_qlen(const char* c, int n) {
    long long num = 0, tot = 0;
    for (int i = 0; i < n; ++i) {
        char ch = c[i];
        if (ch >= '0' && ch <= '9') num = num * 10 + (ch - '0');
        else { if (ch == 'M' || ch == 'I' || ch == 'S' || ch == '=' || ch == 'X') tot += num; num = 0; }
    }
    return tot;
}

// pairedindel :5604-5650 over the CIGARs of one read. sizes: scratch for up to cap doubles
__host__ __device__ inline bool vmx_pairedindel(const char* blob, const int64_t* off, const int32_t* len, int nrec, double indelsize, double* sizes, int cap) {
    int n = 0;
    for (int r = 0; r < nrec; ++r) {
        const char* c = blob + off[r]; double number = 0.;
        for (int i = 0; i < len[r]; ++i) {
            int item = c[i] - '0';
            if (item < 10) number = number * 10. + item;
            else {
                char ch = c[i];
                if (ch != 'I' && ch != 'S' && ch != 'H' && ch != 'P') { if (ch == 'D' && number > indelsize && n < cap) sizes[n++] = number; number = 0.; }
                else { if (ch == 'I' && number > indelsize && n < cap) sizes[n++] = number; number = 0.; }
            }
        }
    }
    for (int i = 1; i < n; ++i) { double v = sizes[i]; int j = i - 1; while (j >= 0 && sizes[j] > v) { sizes[j + 1] = sizes[j]; --j; } sizes[j + 1] = v; }
    double pre = 0;
    for (int i = 0; i < n; ++i) {
        double now = sizes[i];
        double mn = pre < now ? pre : now, mx = pre < now ? now : pre;
        if ((mn / mx) > 0.7) return true;      // clustersize becomes 2 > 1
        pre = now;
    }
    return false;
}

#endif
