// k_ext.hip — kernels of the extend stage that surround the DP kernels of k_dp.hip (SURVEY §8(a) rows E1, E4, E6 + problem plumbing).
//
//   k_ext_phase   one THREAD per read: runs one phase of extend_func (/root/reference/src/vacmap/mammap_clrnano.py:19238-19303) on the
//                 read's segment list (vmx_extend.h): applies the results of the previous DP round and emits the string
//                 descriptors of the next one. Phases: 0 rebuild + divergence problems, 1 filter + right-end extensions,
//                 2 left-end extensions, 3 drop_misplaced (+ right ends again), 4 left ends again, 5 merge / fix_simple_inv /
//                 checkpoints -> gap-fill problems, 6 records (+ pairedindel -> redo with nofilter, :24079-24080).
//   k_desc_lens / k_gather   materialise the (target, query) strings of all problems of a round into code pools
//                 (plain / reversed / complemented / reverse-complemented views of the read or the reference).
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_extend.h"
#include "vmx_ext_state.h"

__device__ __forceinline__ vmx_segs vmx_read_segs(const vmx_ext_args& A, int r, bool snap) {
    vmx_segs S;
    const int64_t c0 = A.coff3[r];            // 3*chain_len+8 slots per read
    const int64_t s0 = A.soff[r];             // chain_len+2 per read
    S.A = (snap ? A.segA_snap : A.segA) + c0; S.st = (snap ? A.st_snap : A.st) + s0; S.en = (snap ? A.en_snap : A.en) + s0;
    S.capA = (int)(A.coff3[r + 1] - c0); S.capS = (int)(A.soff[r + 1] - s0);
    S.nseg = 0;
    return S;
}

// allocate n contiguous problem slots of the current round
// (round 6: the slots' owner — prob_read[slot] = read — is written here by the allocating lane, or by the caller's whole wavefront with `own` false;
//  k_prob_owner, a launch per round, did that before)
__device__ __forceinline__ int vmx_alloc_probs(const vmx_ext_args& A, int n, int r, bool own = true) {
    if (n <= 0) return 0;
    // one atomic add (a compare-and-swap loop on this single word serialises thousands of lanes: measured 14 -> 100 ms per batch); a
    // request that does not fit is rolled back, so that once the kernel has finished the published count is the sum of the granted
    // requests and never passes the capacity — the kernels that re-read it (descriptor lengths, DP sizes, gather, queue order) stay
    // inside the pools. The read that does not fit is reported (VM_READ_CAPACITY); the batch goes on.
    const int b = atomicAdd(A.round_count, n);
    if ((long long)b + n > A.round_cap) { atomicAdd(A.round_count, -n); return -1; }
    if (own) for (int i = 0; i < n; ++i) A.prob_read[b + i] = r;
    return b;
}

// ---- wave-cooperative forms of the segment walks (k_ext_phase with spread 64: one wavefront per read) ----------------------------------
// The walks of vmx_extend.h are chains of dependent loads on one lane: a 35 kb read's 3000 anchors cost its lane 3000 steps (375 with the
// eight-anchor prefetch), and the lane's wave waits on HBM for every one. Here the 64 lanes look at 64 consecutive anchors at once
// against the same `pre` (the last anchor kept) and a ballot finds the next one kept: one step per checkpoint, not per anchor.
// Every lane holds the same scalars; lane 0 stores. Lane 0's stores are followed by a wave exchange before another lane reads them.
__device__ __forceinline__ vmx_anchor vmx_anchor_from_lane(const vmx_anchor& a, int src) {
    vmx_anchor o; o.q = __shfl(a.q, src); const int ls = __shfl((int)(uint16_t)a.l | ((int)a.s << 16), src); o.l = (int16_t)(ls & 0xffff); o.s = (int16_t)(ls >> 16);
    o.r = __shfl((long long)a.r, src); return o;
}
// vmx_split_alignment (E5 checkpoints :21505-21617) on a wavefront; same results, same return codes
__device__ __forceinline__ int vmx_split_alignment_w(vmx_segs& S, int s, long long L, const vmx_ref_view& R, vmx_pair_desc* out, int cap, bool asmv, int lane) {
    const long long min_gap_forcigar = 200;
    int np = 0;
    const int st = S.st[s], en = S.en[s];
    const bool fwd = S.A[st].s == 1;
    if (lane == 0) {
        if (fwd) { vmx_anchor& last = S.A[en - 1]; if (last.l != 0) last = vmx_mk((long long)last.q + last.l, last.r + last.l, 1, 0); }
        else {
            if (S.A[st].l != 0) S.A[st] = vmx_mk(S.A[st].q, S.A[st].r + S.A[st].l, -1, 0);
            if (S.A[en - 1].l != 0) S.A[en - 1] = vmx_mk((long long)S.A[en - 1].q + S.A[en - 1].l, S.A[en - 1].r, -1, 0);
        }
    }
    __threadfence_block();
    (void)__ballot(1);                                           // (the end anchors are rewritten before any lane reads them)
    vmx_anchor pre = fwd ? S.A[st] : S.A[en - 1];
    // forward: i runs st + 1 .. en - 1 upwards; reverse (alignment[::-1]): en - 2 .. st downwards. k counts the anchors already passed.
    const int nwalk = en - st - 1;
    int k = 0;
    while (k < nwalk) {
        const int kk = k + lane < nwalk ? k + lane : nwalk - 1;
        const int i = fwd ? st + 1 + kk : en - 2 - kk;
        const vmx_anchor now = S.A[i];
        long long readgap, refgap;
        if (fwd) { readgap = (long long)now.q - pre.q - pre.l; refgap = (long long)now.r - pre.r - pre.l; }
        else { readgap = (long long)pre.q - now.q - now.l; refgap = (long long)now.r - pre.r - pre.l; }
        const long long mn = readgap < refgap ? readgap : refgap, mx = readgap < refgap ? refgap : readgap;
        const bool lastone = fwd ? (i + 1 == en) : (i == st);
        const bool skip = (!asmv || mx < 2000) && (now.l < 19 || mn < min_gap_forcigar) && !lastone;
        const unsigned long long m = __ballot(k + lane < nwalk && !skip);
        if (!m) { k += 64; continue; }
        const int e = __ffsll((unsigned long long)m) - 1;
        const vmx_anchor kept = vmx_anchor_from_lane(now, e);
        if (out && np >= cap) return VMX_EXT_CAPACITY_DEV;
        vmx_pair_desc d;
        if (fwd) vmx_qt_for_cigar(pre, kept, L, R, &d); else vmx_qt_for_cigar(kept, pre, L, R, &d);
        if (out && lane == 0) out[np] = d;
        if (d.t.len <= 0 || d.q.len <= 0) return VM_READ_RAISED_DEV;    // "Failed to compute CIGAR" :21562
        ++np; pre = kept; k += e + 1;
    }
    if (np == 0) return VM_READ_RAISED_DEV;
    return np;
}

__device__ __forceinline__ vmx_anchor vmx_anchor_up1(const vmx_anchor& a) {      // lane j <- lane j - 1 (lane 0 keeps its own)
    vmx_anchor o; o.q = __shfl_up(a.q, 1); const int ls = __shfl_up((int)(uint16_t)a.l | ((int)a.s << 16), 1); o.l = (int16_t)(ls & 0xffff); o.s = (int16_t)(ls >> 16);
    o.r = __shfl_up((long long)a.r, 1); return o;
}
// vmx_rebuild_chain_break (E1 :23437-23484) on a wavefront. Nearly every anchor of a chain joins the segment of the anchor before it, so the 64
// lanes test 64 consecutive anchors each against its predecessor; the run of joins up to the first anchor that does something else is
// appended in one step, that anchor (a skip: `pre` stays, or a break: new segment) is handled with the scalars every lane holds, and the
// lanes behind it are tested again. Same segments, same return codes as the one-lane form.
__device__ __forceinline__ int vmx_rebuild_chain_break_w(const vmx_anchor* chain_desc, int n, const vmx_ref_view& R, int large_cost, int small_alignment, vmx_segs& S, bool asmv, int lane) {
    S.nseg = 0;
    if (S.capS < 1 || S.capA < 4) return VMX_EXT_CAPACITY_DEV;
    vmx_anchor pre = chain_desc[n - 1];
    int w = 2, nseg = 1, st_cur = 1; long long segq0 = pre.q;     // the open segment: A[st_cur .. w), its first anchor's q; closed segments live in st / en
    bool cur_alive = true;
    if (lane == 0) { S.st[0] = 1; S.A[1] = pre; }
    for (int x0 = 1; x0 < n; x0 += 64) {
        const int nv = n - x0 < 64 ? n - x0 : 64;                 // anchors of this chunk
        const vmx_anchor now = chain_desc[n - 1 - (lane < nv ? x0 + lane : n - 1)];
        const vmx_anchor up = vmx_anchor_up1(now);
        int pos = 0;
        while (pos < nv) {
            const vmx_anchor pv = lane == pos ? pre : up;
            int kind = 2;                                        // 0 join, 1 skip, 2 break
            if (pv.s == now.s) {
                const long long readgap = (long long)now.q - pv.q - pv.l;
                const long long refgap = pv.s == 1 ? (long long)now.r - pv.r - pv.l : (long long)pv.r - now.r - now.l;
                long long d = readgap - refgap; if (d < 0) d = -d;
                if (d <= large_cost && refgap >= (asmv ? 0 : -20) && readgap < 100 && vmx_p2c(R, pv.r) == vmx_p2c(R, now.r)) kind = refgap >= 0 ? 0 : (readgap <= 20 ? 1 : 0);
            }
            const unsigned long long ev = __ballot(lane >= pos && lane < nv && kind != 0);
            const int e = ev ? __ffsll((unsigned long long)ev) - 1 : nv;
            const int cnt = e - pos;
            if (cnt > 0) {
                if (w + cnt + 1 > S.capA) return VMX_EXT_CAPACITY_DEV;
                if (lane >= pos && lane < e) S.A[w + lane - pos] = now;
                w += cnt; pre = vmx_anchor_from_lane(now, e - 1);
            }
            if (e >= nv) break;
            const int ekind = __shfl(kind, e);
            const vmx_anchor ea = vmx_anchor_from_lane(now, e);
            pos = e + 1;
            if (ekind == 1) continue;                            // readgap <= 20 on a back-step: the anchor is dropped, `pre` stays
            // a new segment starts at ea: close the open one (single anchors and short segments are dropped), leave a spare slot
            if (lane == 0) S.en[nseg - 1] = w;
            const int cur_en = w;
            cur_alive = true;
            if (cur_en - st_cur == 1) { w = st_cur; --nseg; cur_alive = false; }
            if (nseg > 0) {
                if (cur_alive) { if (((long long)pre.q + pre.l - segq0) < small_alignment) { w = st_cur; --nseg; cur_alive = false; } }
                else {
                    __threadfence_block(); (void)__ballot(1);
                    const int ps = S.st[nseg - 1], pe = S.en[nseg - 1]; const vmx_anchor fa = S.A[ps], la = S.A[pe - 1];
                    if (((long long)la.q + la.l - fa.q) < small_alignment) { w = ps; --nseg; }
                }
            }
            if (nseg > 0) {
                int last_en = cur_en;
                if (!cur_alive) { __threadfence_block(); (void)__ballot(1); last_en = S.en[nseg - 1]; }
                w = last_en + 2;
            } else w = 1;
            if (nseg + 1 > S.capS || w + 2 > S.capA) return VMX_EXT_CAPACITY_DEV;
            if (lane == 0) { S.st[nseg] = w; S.A[w] = ea; }
            st_cur = w; ++w; ++nseg; pre = ea; segq0 = ea.q; cur_alive = true;
        }
    }
    if (lane == 0) S.en[nseg - 1] = w;
    cur_alive = true;
    if (w - st_cur == 1) { --nseg; cur_alive = false; }
    if (nseg == 0) { S.nseg = 0; return VM_READ_RAISED_DEV; }
    if (cur_alive) { if (((long long)pre.q + pre.l - segq0) < small_alignment) --nseg; }
    else {
        __threadfence_block(); (void)__ballot(1);
        const int ps = S.st[nseg - 1], pe = S.en[nseg - 1]; const vmx_anchor fa = S.A[ps], la = S.A[pe - 1];
        if (((long long)la.q + la.l - fa.q) < small_alignment) --nseg;
    }
    S.nseg = nseg;
    return 0;
}

__global__ void k_ext_phase(vmx_ext_args A, int phase) {
    VMX_SETPRIO(3);
    // one lane per read, every A.spread-th lane of the grid: a read's walk is a chain of dependent loads and data-dependent branches, and the
    // 64 reads of a full wave execute the union of their branches in lock step
    const int sp = A.spread > 1 ? A.spread : 1;
    const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const bool wave = sp == 64 && (phase == 0 || phase == 3 || phase == 5);   // one wavefront per read: the anchor walks run on all 64 lanes, the rest on lane 0
    const int lane = gt & 63;
    if (!wave && gt % sp) return;
    const int r = gt / sp;
    if (r >= A.n_reads) return;
    vmx_ext_read& E = A.er[r];
    if (phase == 0) { E.status = A.lstatus[r]; E.nseg = 0; E.nseg_snap = 0; E.filtered = 0; E.redo = 0; E.prob_base = 0; E.prob_n = 0; E.dp_base = 0; E.dp_n = 0; E.nrec = 0; E.active = 0; E.pass = 0; E.skip_ext = 0; }
    const int cl = A.chain_len[r];
    if (phase == 0) { if (E.status == 0 && cl > 1) E.active = 1; }     // len(raw_alignment_list) <= 1 -> unmapped (:24067)
    if (!E.active || E.status != 0) return;
    if (A.redo_only && !(E.redo == 1 && E.pass == 1)) return;
    vmx_ref_view R; R.codes = A.ref; R.coff = A.coff; R.nseq = A.nseq;
    const long long L = A.roff[r + 1] - A.roff[r];
    const uint8_t* RD = A.ocodes + A.roff[r];
    vmx_segs S = vmx_read_segs(A, r, false);
    S.nseg = E.nseg;
    int32_t* segprob = A.seg_prob + A.soff[r];
    const bool nofilter = A.nodiscard || E.pass == 1;
    if (phase == 0) {
        int rc;
        if (wave) {
            (void)__ballot(1);                                   // (every lane has read the read's state before lane 0 changes it)
            rc = vmx_rebuild_chain_break_w(A.chain + A.la_off[r], cl, R, A.local_maxdiff, A.asm_long ? 30 : (A.mode == 4 ? 40 : 50), S, A.mode == 4, lane);
            __threadfence_block(); (void)__ballot(1);            // the anchors the other lanes stored, before lane 0 reads the segments' ends
            if (lane != 0) return;
        } else
        rc = vmx_rebuild_chain_break(A.chain + A.la_off[r], cl, R, A.local_maxdiff, A.asm_long ? 30 : (A.mode == 4 ? 40 : 50), S, A.mode == 4);     // small_alignment 40 in the asm fork (mammap_asm.py:22321), 30 in ass_extend_func (:23426)
        if (rc < 0) { E.status = rc; return; }
        E.nseg = S.nseg;
        if (A.asm_long) { E.prob_base = 0; E.prob_n = 0; return; }                  // ass_extend_func has no divergence filter
        int b = vmx_alloc_probs(A, S.nseg, r);
        if (b < 0) { E.status = VMX_EXT_CAPACITY_DEV; return; }
        E.prob_base = b; E.prob_n = S.nseg;
        for (int s = 0; s < S.nseg; ++s) {
            vmx_pair_desc d; vmx_qt_for_cigar(SEG_FIRST(S, s), SEG_LAST(S, s), L, R, &d);
            A.desc[b + s] = d;
        }
        return;
    }
    if (phase == 1 && !A.asm_long) {
        // divergence filter :19246-19254
        int b = E.prob_base; int w = 0;
        const int n0 = S.nseg;
        for (int s = 0; s < n0; ++s) {
            const vmx_pair_desc d = A.desc_prev[b + s];
            const long long mn = d.t.len < d.q.len ? d.t.len : d.q.len;
            if (mn == 0) { E.status = VM_READ_RAISED_DEV; return; }           // ZeroDivisionError
            const double ratio = (double)A.ed_out[b + s] / (double)mn;
            if (ratio > A.maxdivergence) continue;
            S.st[w] = S.st[s]; S.en[w] = S.en[s]; ++w;
        }
        S.nseg = w; E.nseg = w;
    }
    if (phase == 2 || (phase == 4 && E.skip_ext == 0)) {
        // apply right-end results, then set up the left ends (they see the extended right end of their left neighbour)
        for (int s = 0; s < S.nseg; ++s) if (segprob[s] >= 0) vmx_ext_apply(S, s, 1, A.ext_te[E.prob_base + segprob[s]], A.ext_qe[E.prob_base + segprob[s]]);
    }
    if (wave && phase == 3) {
        // phase 3 on a wavefront: lane 0 applies the left-end results, all lanes copy the snapshot for the nofilter redo, lane 0 goes on alone
        (void)__ballot(1);
        if (lane == 0 && E.skip_ext == 0) for (int s = 0; s < S.nseg; ++s) if (segprob[s] >= 0) vmx_ext_apply(S, s, 0, A.ext_te[E.prob_base + segprob[s]], A.ext_qe[E.prob_base + segprob[s]]);
        __threadfence_block(); (void)__ballot(1);
        if (E.pass == 0) {
            vmx_segs P = vmx_read_segs(A, r, true);
            for (int s = 0; s < S.nseg; ++s) {
                const int a = S.st[s] - 1, b = S.en[s];
                for (int t = a + lane; t <= b; t += 64) P.A[t] = S.A[t];
                if (lane == 0) { P.st[s] = S.st[s]; P.en[s] = S.en[s]; }
            }
        }
        (void)__ballot(1);
        if (lane != 0) return;
    }
    if (wave && phase == 5) {
        // phase 5 on a wavefront: lane 0 applies the left-end results and runs the segment-level steps (merge, fix_simple_inv: a few
        // segments), every lane takes part in the checkpoint walks
        int rc = 0, nseg = S.nseg;
        (void)__ballot(1);                                       // (every lane has read the read's state before lane 0 changes it)
        if (lane == 0) {
            if (E.skip_ext == 0) for (int s = 0; s < S.nseg; ++s) if (segprob[s] >= 0) vmx_ext_apply(S, s, 0, A.ext_te[E.prob_base + segprob[s]], A.ext_qe[E.prob_base + segprob[s]]);
            E.prob_n = 0;
            vmx_merge_conjacent(S, R, A.dup + A.soff[r]);
            rc = vmx_fix_simple_inv(S, R, RD, L, A.mode == 3 || A.mode == 4);
            nseg = S.nseg;
            if (rc < 0) E.status = rc; else E.nseg = nseg;
        }
        __threadfence_block();
        rc = __shfl(rc, 0); nseg = __shfl(nseg, 0);
        if (rc < 0) return;
        S.nseg = nseg;
        int total = 0;
        for (int s = 0; s < S.nseg; ++s) {
            const int np = vmx_split_alignment_w(S, s, L, R, nullptr, 0, A.mode == 4, lane);
            if (np < 0) { if (lane == 0) E.status = np; return; }
            if (lane == 0) segprob[s] = np;
            total += np;
        }
        int b = lane == 0 ? vmx_alloc_probs(A, total, r, false) : 0;
        b = __shfl(b, 0);
        if (b < 0) { if (lane == 0) E.status = VMX_EXT_CAPACITY_DEV; return; }
        for (int i = lane; i < total; i += 64) A.prob_read[b + i] = r;
        int k = 0;
        for (int s = 0; s < S.nseg; ++s) {
            const int np = vmx_split_alignment_w(S, s, L, R, A.desc + b + k, total - k, A.mode == 4, lane);
            if (np < 0) { if (lane == 0) E.status = np; return; }
            k += np;
        }
        if (lane == 0) { E.dp_base = b; E.dp_n = k; E.prob_base = b; E.prob_n = total; }
        return;
    }
    if ((phase == 3 && !wave) || phase == 5) {
        if (E.skip_ext == 0) for (int s = 0; s < S.nseg; ++s) if (segprob[s] >= 0) vmx_ext_apply(S, s, 0, A.ext_te[E.prob_base + segprob[s]], A.ext_qe[E.prob_base + segprob[s]]);
    }
    if (phase >= 1 && phase <= 5) E.prob_n = 0;      // results of the previous round are consumed; a new round may follow below
    if (phase == 3) {
        // snapshot for the nofilter redo (:24080 restarts extend_func; everything up to here is identical in both runs)
        if (E.pass == 0) {
            if (!wave) {
                vmx_segs P = vmx_read_segs(A, r, true);
                for (int s = 0; s < S.nseg; ++s) { P.st[s] = S.st[s]; P.en[s] = S.en[s]; for (int t = S.st[s] - 1; t <= S.en[s]; ++t) P.A[t] = S.A[t]; }
            }
            E.nseg_snap = S.nseg;
        }
        const int o_len = S.nseg;
        if (S.nseg > 2 && !nofilter && A.mode != 4) { int iloc = 0; while (iloc < S.nseg - 2) { if (vmx_drop_misplaced(S, iloc)) continue; else iloc += 1; } }
        E.nseg = S.nseg;
        E.skip_ext = 1;
        if (S.nseg < o_len) { E.filtered = 1; E.skip_ext = 0; }
    }
    if (phase == 1 || (phase == 3 && E.skip_ext == 0)) {
        // right ends
        int np = 0;
        for (int s = 0; s < S.nseg; ++s) segprob[s] = -1;
        // two passes: count, allocate, fill
        vmx_pair_desc tmp;
        // descriptors are cheap to recompute: first decide which segments issue a DP call (setup also rewrites end anchors when no call is made)
        int b = -1;
        for (int pass = 0; pass < 2; ++pass) {
            int k = 0;
            for (int s = 0; s < S.nseg; ++s) {
                if (pass == 0) { if (vmx_ext_setup(S, s, 1, L, R, &tmp)) { segprob[s] = k++; } }
                else if (segprob[s] >= 0) { vmx_ext_setup(S, s, 1, L, R, &tmp); A.desc[b + segprob[s]] = tmp; }
            }
            if (pass == 0) { np = k; b = vmx_alloc_probs(A, np, r); if (b < 0) { E.status = VMX_EXT_CAPACITY_DEV; return; } }
        }
        E.prob_base = b < 0 ? 0 : b; E.prob_n = np;
        return;
    }
    if (phase == 2 || (phase == 4 && E.skip_ext == 0)) {
        int np = 0; vmx_pair_desc tmp; int b = -1;
        for (int s = 0; s < S.nseg; ++s) segprob[s] = -1;
        for (int pass = 0; pass < 2; ++pass) {
            int k = 0;
            for (int s = 0; s < S.nseg; ++s) {
                if (pass == 0) { if (vmx_ext_setup(S, s, 0, L, R, &tmp)) { segprob[s] = k++; } }
                else if (segprob[s] >= 0) { vmx_ext_setup(S, s, 0, L, R, &tmp); A.desc[b + segprob[s]] = tmp; }
            }
            if (pass == 0) { np = k; b = vmx_alloc_probs(A, np, r); if (b < 0) { E.status = VMX_EXT_CAPACITY_DEV; return; } }
        }
        E.prob_base = b < 0 ? 0 : b; E.prob_n = np;
        return;
    }
    if (phase == 5) {
        vmx_merge_conjacent(S, R, A.dup + A.soff[r]);
        int rc = vmx_fix_simple_inv(S, R, RD, L, A.mode == 3 || A.mode == 4);      // the asm fork keeps mode R's body (mammap_asm.py:17158-17207)
        if (rc < 0) { E.status = rc; return; }
        E.nseg = S.nseg;
        // checkpoints -> gap-fill problems: count first, then allocate exactly that many slots
        int total = 0;
        for (int s = 0; s < S.nseg; ++s) {
            int np = vmx_split_alignment(S, s, L, R, nullptr, 0, A.mode == 4);
            if (np < 0) { E.status = np; return; }
            segprob[s] = np; total += np;
        }
        int b = vmx_alloc_probs(A, total, r);
        if (b < 0) { E.status = VMX_EXT_CAPACITY_DEV; return; }
        int k = 0;
        for (int s = 0; s < S.nseg; ++s) {
            int np = vmx_split_alignment(S, s, L, R, A.desc + b + k, total - k, A.mode == 4);
            if (np < 0) { E.status = np; return; }
            k += np;
        }
        E.dp_base = b; E.dp_n = k; E.prob_base = b; E.prob_n = total;
        return;
    }
}

// lengths of the strings of every problem of the round (for the offset scans)
__global__ void k_desc_lens(const vmx_pair_desc* __restrict__ desc, const int32_t* __restrict__ n_prob, int64_t* __restrict__ tl, int64_t* __restrict__ ql) {
    const int n = *n_prob;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) { tl[i] = desc[i].t.len; ql[i] = desc[i].q.len; }
}

__device__ __forceinline__ void vmx_gather_one(const vmx_sdesc& d, const uint8_t* rd, const uint8_t* ref, uint8_t* out) {
    const uint8_t* src = d.src == 0 ? rd : ref;
    const int n = d.len;
    for (int i = (int)(threadIdx.x & 63); i < n; i += 64) {       // one wavefront per problem
        uint8_t c;
        if (d.op == 0) c = src[d.start + i];
        else if (d.op == 1) c = src[d.start + n - 1 - i];
        else if (d.op == 2) { c = src[d.start + i]; c = c < 4 ? 3 - c : 4; }
        else { c = src[d.start + n - 1 - i]; c = c < 4 ? 3 - c : 4; }
        out[i] = c;
    }
}

// prob_read[i] = read index of problem i (to find the oriented read codes)
__global__ void __launch_bounds__(256) k_gather(const vmx_pair_desc* __restrict__ desc, const int32_t* __restrict__ n_prob, const int32_t* __restrict__ prob_read,
                                                const uint8_t* __restrict__ ocodes, const int64_t* __restrict__ roff, const uint8_t* __restrict__ ref,
                                                const int64_t* __restrict__ t_off, const int64_t* __restrict__ q_off, uint8_t* __restrict__ tpool,
                                                uint8_t* __restrict__ qpool, int64_t pool_cap, vmx_ext_read* __restrict__ er) {
    // a problem's two strings are a few hundred bytes: one wavefront each (a 256-thread workgroup per problem kept a quarter as many
    // descriptor / source loads in flight and left most of its threads without a byte to copy)
    const int n = *n_prob;
    const int wpb = (int)(blockDim.x >> 6), wv = (int)(threadIdx.x >> 6);
    for (int i = (int)blockIdx.x * wpb + wv; i < n; i += (int)gridDim.x * wpb) {
        const vmx_pair_desc d = desc[i];
        // the string pools are an estimate (6 B per read base): a problem that does not fit marks ITS READ (round 6; a batch-wide flag before) — the read is run again
        // alone with larger pools (align_device), the others keep their results
        if (t_off[i] + d.t.len > pool_cap || q_off[i] + d.q.len > pool_cap) { if ((threadIdx.x & 63) == 0) er[prob_read[i]].status = VMX_EXT_CAPACITY_DEV; continue; }
        const uint8_t* rd = ocodes + roff[prob_read[i]];
        vmx_gather_one(d.t, rd, ref, tpool + t_off[i]);
        vmx_gather_one(d.q, rd, ref, qpool + q_off[i]);
    }
}

// mark which read owns each problem slot of the round
__global__ void k_prob_owner(const vmx_ext_read* __restrict__ er, int n_reads, int use_dp, int32_t* __restrict__ prob_read) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_reads) return;
    const vmx_ext_read E = er[r];
    if (!E.active || E.status != 0) return;
    if (use_dp /* redo_only */ && !(E.redo == 1 && E.pass == 1)) return;
    for (int i = 0; i < E.prob_n; ++i) prob_read[E.prob_base + i] = r;
}

// gap-fill problem table from the descriptors + offsets; tb/bnd/run/cig sizes per problem go to size arrays for the scans
__global__ void k_dp_sizes(const vmx_pair_desc* __restrict__ desc, const int32_t* __restrict__ n_prob, int64_t* __restrict__ tb_sz, int64_t* __restrict__ bnd_sz,
                           int64_t* __restrict__ run_sz, int64_t* __restrict__ cig_sz) {
    const int n = *n_prob;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
        const long long tl = desc[i].t.len, ql = desc[i].q.len;
        tb_sz[i] = VMX_TB_BYTES_NS(tl, ql);
        bnd_sz[i] = 3 * (ql + 1); run_sz[i] = tl + ql + 2; cig_sz[i] = 2 * (tl + ql) + 16;
    }
}
__global__ void k_dp_table(const vmx_pair_desc* __restrict__ desc, const int32_t* __restrict__ n_prob, const int64_t* __restrict__ t_off, const int64_t* __restrict__ q_off,
                           const int64_t* __restrict__ tb_off, const int64_t* __restrict__ bnd_off, const int64_t* __restrict__ run_off,
                           const int64_t* __restrict__ cig_off, vmx_dp_prob* __restrict__ probs) {
    const int n = *n_prob;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
        vmx_dp_prob p; p.t_off = t_off[i]; p.q_off = q_off[i]; p.tl = desc[i].t.len; p.ql = desc[i].q.len;
        p.tb_off = tb_off[i]; p.bnd_off = bnd_off[i]; p.run_off = run_off[i]; p.cig_off = cig_off[i];
        probs[i] = p;
    }
}

// result compaction: per-read record counts / blob bytes -> (scan on the host side of the stream) -> packed arrays
__global__ void k_res_sizes(const vmx_ext_read* __restrict__ er, const vm_record* __restrict__ rec, const int64_t* __restrict__ soff, int n_reads,
                            int64_t* __restrict__ recn, int64_t* __restrict__ blobn) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_reads) return;
    const vmx_ext_read E = er[r];
    long long nr = 0, nb = 0;
    if (E.active && E.status == 0) { nr = E.nrec; for (int x = 0; x < E.nrec; ++x) nb += rec[soff[r] + x].cigar_len + 1; }
    recn[r] = nr; blobn[r] = nb;
}
__global__ void k_res_pack(const vmx_ext_read* __restrict__ er, const vm_record* __restrict__ rec, const char* __restrict__ blob, const int64_t* __restrict__ soff,
                           const int64_t* __restrict__ blob_off, int n_reads, const int64_t* __restrict__ rec_o, const int64_t* __restrict__ blob_o,
                           vm_record* __restrict__ out_rec, char* __restrict__ out_blob) {
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const vmx_ext_read E = er[r];
        if (!(E.active && E.status == 0)) continue;
        long long bo = blob_o[r];
        for (int x = 0; x < E.nrec; ++x) {
            vm_record rc = rec[soff[r] + x];
            const char* src = blob + blob_off[r] + rc.cigar_off;
            for (long long i = threadIdx.x; i <= rc.cigar_len; i += blockDim.x) out_blob[bo + i] = i < rc.cigar_len ? src[i] : 0;
            if (threadIdx.x == 0) { rc.cigar_off = bo; out_rec[rec_o[r] + x] = rc; }
            bo += rc.cigar_len + 1;
        }
    }
}

// E6 records (:20731-20838) + pairedindel (:5604). One WAVEFRONT per read: every lane follows the same (uniform) record logic on the same
// data, the bytes of the problems' CIGARs — 8 KB per 15 kb read, which one lane used to move a byte at a time — are copied by the 64 lanes
// together, and lane 0 stores the records. The reference's check that the concatenated CIGAR consumes the whole read (:20779-20786) uses the
// per-problem query lengths the trace kernel counted from the operators it emitted (cig_q) instead of parsing the text again.
__global__ void __launch_bounds__(64) k_ext_records(vmx_ext_args A, const vmx_dp_prob* __restrict__ probs, const char* __restrict__ cig_pool, const int32_t* __restrict__ cig_len,
                                                    const int32_t* __restrict__ cig_q) {
    VMX_SETPRIO(3);
    const int lane = vmx_lane();
    const int r = (int)blockIdx.x;
    if (r >= A.n_reads) return;
    vmx_ext_read& E = A.er[r];
    if (!E.active || E.status != 0) return;
    if (A.redo_only && !(E.redo == 1 && E.pass == 1)) return;
    vmx_ref_view R; R.codes = A.ref; R.coff = A.coff; R.nseq = A.nseq;
    const long long L = A.roff[r + 1] - A.roff[r];
    vmx_segs S = vmx_read_segs(A, r, false); S.nseg = E.nseg;
    const int32_t* segprob = A.seg_prob + A.soff[r];
    vm_record* REC = A.rec + A.soff[r];
    char* BLOB = A.rec_blob + A.blob_off[r];
    const long long blob_cap = A.blob_off[r + 1] - A.blob_off[r];
    int64_t* roffs = A.rec_coff + A.soff[r]; int32_t* rlens = A.rec_clen + A.soff[r];
    const bool need_reverse = A.gscore[r] < 0.0;
    const char clip = A.hardclip ? 'H' : 'S';
    const int dp_base = E.dp_base, pass = E.pass, filtered = E.filtered, nseg_snap = E.nseg_snap;
    long long w = 0; int k = 0;
    int fail = 0;
    for (int s = 0; s < S.nseg && !fail; ++s) {
        const vmx_anchor a0 = S.A[S.st[s]].s == 1 ? S.A[S.st[s]] : S.A[S.en[s] - 1];     // new_alignment[0]
        const vmx_anchor a1 = S.A[S.st[s]].s == 1 ? S.A[S.en[s] - 1] : S.A[S.st[s]];     // new_alignment[-1]
        const int c = vmx_p2c(R, a0.r); const long long bias = R.coff[c];
        vm_record rec; rec.read_idx = r; rec.contig = c; rec.mapq = A.mapq[r];
        long long q_st, q_en;
        if (a0.s == 1) { q_st = a0.q; q_en = (long long)a1.q + a1.l; rec.strand = need_reverse ? -1 : 1; }
        else { q_st = L - a0.q - a0.l; q_en = L - a1.q; rec.strand = need_reverse ? 1 : -1; }
        rec.q_st = q_st; rec.q_en = q_en; rec.r_st = a0.r - bias; rec.r_en = a1.r + a1.l - bias;
        const int np = segprob[s];
        // worst-case length check before writing; the problems' CIGAR lengths and query lengths summed across the lanes
        long long need = 0, qsum = 0;
        for (int x = lane; x < np; x += 64) { need += cig_len[dp_base + k + x]; qsum += cig_q[dp_base + k + x]; }
        need = vmx_wave_sum_i64(need) + 48; qsum = vmx_wave_sum_i64(qsum);
        if (w + need > blob_cap) { fail = VMX_EXT_CAPACITY_DEV; break; }
        const long long st = w;
        char tmp[24];
        if (q_st > 0) { const int nd = vmx_put_int(tmp, q_st); if (lane == 0) { for (int i = 0; i < nd; ++i) BLOB[w + i] = tmp[i]; BLOB[w + nd] = clip; } w += nd + 1; }
        const long long wc0 = w;                      // start of the segment's CIGAR body (behind the leading clip)
        for (int x = 0; x < np; ++x) {
            const int p = dp_base + k + x; const char* src = cig_pool + probs[p].cig_off; const int n = cig_len[p];
            int from = 0;
            if (A.mode == 4 && x > 0 && n > 0 && w > wc0) {
                // link_cigar (mammap_asm.py:22365-22410): the asm fork joins the pieces into ONE string, adding the counts when the last operator
                // written equals the first operator of the next piece. Every lane reads the same few bytes; lane 0 writes the joined count.
                __syncthreads();
                const char last = BLOB[w - 1];
                int j = 0; long long num2 = 0;
                while (j < n && src[j] >= '0' && src[j] <= '9') { num2 = num2 * 10 + (src[j] - '0'); ++j; }
                if (j < n && src[j] == last) {
                    long long i = w - 1, num1 = 0, factor = 1; bool brk = false;
                    while (i > wc0) { --i; const char ch = BLOB[i]; if (ch >= '0' && ch <= '9') { num1 += (long long)(ch - '0') * factor; factor *= 10; } else { brk = true; break; } }
                    w = brk ? i + 1 : wc0;
                    __syncthreads();              // every lane has parsed the old count before lane 0 overwrites it
                    const int nd = vmx_put_int(tmp, num1 + num2);
                    if (lane == 0) { for (int t = 0; t < nd; ++t) BLOB[w + t] = tmp[t]; BLOB[w + nd] = last; }
                    w += nd + 1;
                    from = j + 1;
                }
            }
            for (int i = from + lane; i < n; i += 64) BLOB[w + (i - from)] = src[i];
            w += n - from;
        }
        k += np;
        long long ql = qsum;
        if (a0.s == 1 && a1.l > 0) { const int nd = vmx_put_int(tmp, a1.l); if (lane == 0) { for (int i = 0; i < nd; ++i) BLOB[w + i] = tmp[i]; BLOB[w + nd] = 'M'; } w += nd + 1; ql += a1.l; }
        if (L - q_en > 0) { const int nd = vmx_put_int(tmp, L - q_en); if (lane == 0) { for (int i = 0; i < nd; ++i) BLOB[w + i] = tmp[i]; BLOB[w + nd] = clip; } w += nd + 1; if (!A.hardclip) ql += L - q_en; }
        if (q_st > 0 && !A.hardclip) ql += q_st;
        if (lane == 0) BLOB[w] = 0;
        rec.cigar_off = st; rec.cigar_len = w - st;
        if (lane == 0) { roffs[s] = st; rlens[s] = (int32_t)(w - st); }
        ++w;
        // :20779-20786 len(Cigar(cigarstring)) against the read (soft clips count, hard clips do not)
        if (!A.hardclip) { if (L != ql) { fail = VM_READ_RAISED_DEV; break; } }
        else { if ((q_en - q_st) != ql) { fail = VM_READ_RAISED_DEV; break; } }
        if (lane == 0) REC[s] = rec;
    }
    if (fail) { if (lane == 0) E.status = fail; return; }
    __syncthreads();                                          // the records and the text are complete before lane 0 reads them back
    if (lane != 0) return;
    E.nrec = S.nseg;
    if (need_reverse) { for (int a = 0, b = S.nseg - 1; a < b; ++a, --b) { vm_record t = REC[a]; REC[a] = REC[b]; REC[b] = t; } }
    // redo with nofilter (:24079-24080)
    if (pass == 0 && !A.nodiscard && filtered && S.nseg > 0) {
        if (vmx_pairedindel(BLOB, roffs, rlens, S.nseg, 30.0, (double*)(A.dup_d + A.blob_off[r] / 8), (int)(blob_cap / 8 - 1))) {
            E.redo = 1; E.pass = 1; E.nrec = 0;
            // restore the state saved after the first extension round
            vmx_segs P = vmx_read_segs(A, r, true);
            for (int s = 0; s < nseg_snap; ++s) { S.st[s] = P.st[s]; S.en[s] = P.en[s]; for (int t = P.st[s] - 1; t <= P.en[s]; ++t) S.A[t] = P.A[t]; }
            E.nseg = nseg_snap; E.filtered = 0; E.skip_ext = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------ divergence filter, tier 0
// An upper bound of the segment's edit distance from its own anchors: cutting the (query, target) pair of the problem at the start
// corner of every anchor gives pieces of an anchor (an exact match: counted by direct comparison) plus the short gap behind it
// (rebuild_chain_break keeps read gaps < 100, :23437-23484), and the sum of the pieces' costs is the cost of ONE valid alignment,
// hence >= the edit distance whatever the anchors are. For a read at the divergence the filter tolerates this bound is already
// below the threshold, which settles the segment without the 10^4-step serial bit-vector DP over the whole segment; the pieces
// are independent, one lane each. A segment whose corners are not monotone in both strings, or with a gap longer than 256 rows,
// gets -1 and goes to the banded tiers (k_ed_band.hip).
__device__ inline int vmx_ed_small(const uint8_t* P, int m, const uint8_t* T, int n) {       // exact unit-cost distance, m <= 256
    if (m == 0) return n;
    if (n == 0) return m;
    const int W = (m + 63) >> 6;
    unsigned long long q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0}, q3[4] = {0, 0, 0, 0}, q4[4] = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < W) {
            int lim = m - (w << 6); if (lim > 64) lim = 64;
            unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
            for (int x = 0; x < lim; ++x) {
                const uint8_t c = P[(w << 6) + x]; const unsigned long long bit = 1ULL << x;
                if (c == 0) a0 |= bit; else if (c == 1) a1 |= bit; else if (c == 2) a2 |= bit; else if (c == 3) a3 |= bit; else a4 |= bit;
            }
            q0[w] = a0; q1[w] = a1; q2[w] = a2; q3[w] = a3; q4[w] = a4;
        }
    }
    unsigned long long Pv[4] = {~0ULL, ~0ULL, ~0ULL, ~0ULL}, Mv[4] = {0, 0, 0, 0};
    int score = m;
    const int lastbit = (m - 1) & 63;
    for (int j = 0; j < n; ++j) {
        const int c = T[j];
        int hin = 1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < W) {
                unsigned long long Eq = c == 0 ? q0[w] : (c == 1 ? q1[w] : (c == 2 ? q2[w] : (c == 3 ? q3[w] : q4[w])));
                const unsigned long long neg = (unsigned long long)((unsigned)hin >> 31), pos = (unsigned long long)((unsigned)(-hin) >> 31);
                const unsigned long long Xv = Eq | Mv[w];
                Eq |= neg;
                const unsigned long long Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
                unsigned long long Ph = Mv[w] | ~(Xh | Pv[w]);
                unsigned long long Mh = Pv[w] & Xh;
                const int hb = (w == W - 1) ? lastbit : 63;
                const int hout = (int)((Ph >> hb) & 1ULL) - (int)((Mh >> hb) & 1ULL);
                Ph = (Ph << 1) | pos; Mh = (Mh << 1) | neg;
                Pv[w] = Mh | ~(Xv | Ph);
                Mv[w] = Ph & Xv;
                hin = hout;
            }
        }
        score += hin;        // delta of the last row
    }
    return score;
}

__global__ void __launch_bounds__(64) k_ed_anchor_bound(vmx_ext_args A, const int32_t* __restrict__ probread, const uint8_t* __restrict__ qpool,
                                                        const int64_t* __restrict__ qoff, const uint8_t* __restrict__ tpool,
                                                        const int64_t* __restrict__ toff, int64_t* __restrict__ ub_out) {
    const int lane = vmx_lane();
    const int n_prob = *A.round_count;
    for (int p = (int)blockIdx.x; p < n_prob; p += (int)gridDim.x) {
        const int r = probread[p];
        const vmx_ext_read E = A.er[r];
        const int s = p - E.prob_base;
        const vmx_segs S = vmx_read_segs(A, r, false);
        const int st = S.st[s], en = S.en[s];
        const vmx_anchor first = S.A[st], last = S.A[en - 1];
        const uint8_t* Q = qpool + qoff[p]; const int ql = (int)(qoff[p + 1] - qoff[p]);
        const uint8_t* T = tpool + toff[p]; const int tl = (int)(toff[p + 1] - toff[p]);
        const int K = en - 1 - st;                      // anchors inside the strings (the last one only contributes its start)
        const bool fwd = first.s == 1;
        long long cost = 0; bool bad = false;
        // corner k in string order: + strand anchors st .. en-2 ascending, - strand en-2 .. st (the query string is the reversed read)
        for (int k = lane - 1; k < K; k += 64) {
            int u0, v0, l0, u1, v1;
            if (k < 0) { u0 = 0; v0 = 0; l0 = 0; }         // leading piece from the string start to the first corner (lane 0)
            else {
                const vmx_anchor a = fwd ? S.A[st + k] : S.A[en - 2 - k];
                u0 = fwd ? a.q - first.q : last.q - (a.q + a.l);
                v0 = fwd ? (int)(a.r - first.r) : (int)(a.r - (last.r + last.l));
                l0 = a.l;
            }
            if (k + 1 < K) {
                const vmx_anchor b = fwd ? S.A[st + k + 1] : S.A[en - 2 - (k + 1)];
                u1 = fwd ? b.q - first.q : last.q - (b.q + b.l);
                v1 = fwd ? (int)(b.r - first.r) : (int)(b.r - (last.r + last.l));
            } else { u1 = ql; v1 = tl; }
            if (u0 < 0 || v0 < 0 || u1 < u0 || v1 < v0 || u1 > ql || v1 > tl) { bad = true; continue; }
            int lm = l0; if (lm > u1 - u0) lm = u1 - u0; if (lm > v1 - v0) lm = v1 - v0;       // matched part that fits before the next corner
            int mism = 0;
            for (int x = 0; x < lm; ++x) mism += Q[u0 + x] != T[v0 + x] ? 1 : 0;
            const int gm = u1 - (u0 + lm), gn = v1 - (v0 + lm);
            if (gm > 256) { bad = true; continue; }
            cost += mism + vmx_ed_small(Q + u0 + lm, gm, T + v0 + lm, gn);
        }
        const long long tot = vmx_wave_sum_i64(cost);
        const bool anybad = __any(bad) || K < 0;
        if (lane == 0) ub_out[p] = anybad ? -1LL : tot;
    }
}
