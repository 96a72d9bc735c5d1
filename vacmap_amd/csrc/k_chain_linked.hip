// k_chain_linked.hip — the batch-LINKED chain DPs of -mode asm on gfx950.
//
//   k_chain_linked   linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all   /root/reference/src/vacmap/mammap_asm.py:21686-21870
//                    (first round: minimizer anchors of 100 kb seeding windows, batches of more than 500 000 anchors) and
//                    linked_get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_all      :21504-21685
//                    (second round: 9-mer anchors along the first-round path; co-linear steps also pay readgapcost_list, no bail-out).
//   k_link_carry     what assembly_get_readmap_DP_test does between two batches (:23250-23272): the anchors within skipcost + 56 of the best
//                    score, re-based scores, negated predecessors, the running maximum — the state the next batch starts from.
//
// One wavefront per assembly contig and batch; the batch has up to ~600 000 anchors, so S, P and the score-sorted index live in HBM (L2).
// The candidate scan is the 64-wide descending-S scan of k_chain_global (exclusive prefix max -> exact sequential break index and strict-'>'
// winner); insertion points replay the reference's bisection (insertpoint_score :20031-20050) on (#below, #not above).
//
// COLD ENTRIES. The scan of anchor i visits entry j only while S[j] > max_scores - l_i, and after the top entry t has been evaluated
// max_scores >= S[t] + (1 + l_i - l_t) - pen, pen = max(skipcost + 36 (extra's last value), gapcost_list's maximum (+ the read-gap maximum)).
// So no scan ever goes below  S_top - (pen + Lmax)  =: S_top - M, and S_top never decreases: an anchor that scores <= S_top - M when it is
// inserted can never be visited. Such an anchor is counted (n_cold, its score tracked in cold_max) but not stored, and the bisection of every
// later entry is replayed with the cold count added to both bounds (the cold entries all lie below it), so the stored ("hot") entries are
// in the order the full index would give them. In practice nearly every anchor is hot when it is inserted — even an isolated noise hit hangs
// itself onto the best chain at the skip penalty and lands a few dozen places below the top — so an insertion shifts a short tail of the
// index, not the index. What the caller reads of S_arg afterwards — its last entry and the slice above best - skipcost - 56 down to the first
// entry at or below that (:23256-23264) — lies in the stored part as long as that boundary entry scores above cold_max; k_link_carry checks
// exactly that and refuses the contig otherwise, as it does when more anchors would be carried than the staging area holds or when GC-exact
// bails out to the fork's linked GC-fast (:23246-23247, not built on the device).
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_link.h"

__device__ __forceinline__ void vmx_link_geometry(int qi, long long ri, int si, int li, int qj, long long rj, int sj, int lj,
                                                  long long& readgap, long long& refgap, long long& bonus) {      // mammap_asm.py:21729-21757
    readgap = (long long)qi - qj - lj;
    if (readgap < 0) {
        bonus = (long long)qi + li - qj - lj;
        readgap = 0;
        const long long nov = (long long)qi - qj;
        if (si == sj) { if (si == 1) refgap = ri - rj - nov; else refgap = rj + lj - nov - ri - li; }
        else { if (sj == -1) refgap = ri + lj - nov - rj; else refgap = ri + li - rj - nov; }
    } else {
        bonus = li;
        if (si == sj) { if (si == 1) refgap = ri - rj - lj; else refgap = rj - ri - li; }
        else { if (sj == -1) refgap = ri - rj; else refgap = ri + li - rj - lj; }
    }
}

// SA[loc + 1 : cnt + 1] = SA[loc : cnt]; SA[loc] = val  (cnt entries before the call)
__device__ __forceinline__ void vmx_link_sa_insert(int32_t* SA, int loc, int cnt, int val, int lane) {
    for (int hi = cnt; hi > loc; hi -= 256) {
        const int x0 = hi - lane;
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (x0 > loc) v0 = SA[x0 - 1];
        if (x0 - 64 > loc) v1 = SA[x0 - 65];
        if (x0 - 128 > loc) v2 = SA[x0 - 129];
        if (x0 - 192 > loc) v3 = SA[x0 - 193];
        __syncthreads();
        if (x0 > loc) SA[x0] = v0;
        if (x0 - 64 > loc) SA[x0 - 64] = v1;
        if (x0 - 128 > loc) SA[x0 - 128] = v2;
        if (x0 - 192 > loc) SA[x0 - 192] = v3;
        __syncthreads();
    }
    if (lane == 0) SA[loc] = val;
    __syncthreads();
}

// one batch of one contig per workgroup (one wavefront)
__global__ void __launch_bounds__(64) k_chain_linked(vmx_link_job* __restrict__ jobs, int n_jobs, vmx_tables tab, const double* __restrict__ gapcost_list,
                                                    double skipcost, int maxdiff, int maxgap, int lc, double margin_base, double max_factor) {
    __shared__ double s_gapcost[64];
    const int lane = vmx_lane();
    for (int x = lane; x <= maxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    for (int jb = (int)blockIdx.x; jb < n_jobs; jb += (int)gridDim.x) {
        vmx_link_job& J = jobs[jb];
        vmx_link_state& ST = *J.state;
        if (ST.status != 0 || J.n_new <= 0) { if (lane == 0) J.ran = 0; continue; }
        const int n_pre = ST.n_pre;
        const int base = J.cap_pre - n_pre;                       // carried rows are right-aligned in front of the new ones
        const vmx_anchor* A = J.rows + base;
        const int n = n_pre + J.n_new;
        double* S = J.S + base; int32_t* P = J.P + base; int32_t* SA = J.SA;
        // M = pen + Lmax (file header): the longest anchor of the batch
        int lmax = 0;
        for (int i = lane; i < n; i += 64) { const int l = (int)A[i].l & 0xffff; lmax = l > lmax ? l : lmax; }
        lmax = vmx_wave_max_i32(lmax);
        const double M = margin_base + (double)lmax;
        double g_max_scores; int g_max_index; long long prereadloc; int pre_size;
        if (n_pre > 0) {
            for (int i = lane; i < n_pre; i += 64) { S[i] = ST.pre_S[i]; P[i] = ST.pre_P[i]; }
            g_max_scores = ST.g_max_scores; g_max_index = ST.g_max_index; prereadloc = ST.prereadloc; pre_size = n_pre;
        } else {
            if (lane == 0) { S[0] = (double)((int)A[0].l & 0xffff); P[0] = VMX_NOPRE; }
            g_max_scores = (double)((int)A[0].l & 0xffff); g_max_index = 0; prereadloc = A[0].q; pre_size = 1;
        }
        if (lane == 0) SA[0] = 0;
        __syncthreads();
        int testspace_en = 1;             // anchors [0, testspace_en) are finished and indexed (hot) or counted (cold)
        int hot = 1; long long n_cold = 0; double cold_max = -1e300;
        long long opcount = 0; bool bailed = false;
        auto index_anchor = [&](int k) {
            const double Sk = S[k];
            const double top = S[SA[hot - 1]];
            if (Sk <= top - M) { ++n_cold; cold_max = Sk > cold_max ? Sk : cold_max; return; }
            const int a = vmx_sorted_count(S, SA, hot, Sk, false, lane);
            int b = a;
            if (a < hot && S[SA[a]] == Sk) b = vmx_sorted_count(S, SA, hot, Sk, true, lane);
            // insertpoint_score (:20031-20050) replayed on the FULL index: the cold entries all lie below this score
            const long long af = a + n_cold, bf = b + n_cold;
            long long i = 0, j = hot + n_cold, loc = -1;
            while (i < j) {
                const long long mid = (i + j) >> 1;
                if (mid < af) i = mid + 1;
                else if (mid >= bf) j = mid;
                else { loc = mid + 1; break; }
            }
            if (loc < 0) loc = j;
            vmx_link_sa_insert(SA, (int)(loc - n_cold), hot, k, lane);
            ++hot;
        };
        for (int i = pre_size; i < n; ++i) {
            const vmx_anchor ai = A[i];
            const int qi = ai.q, li = (int)ai.l & 0xffff, si = (int)ai.s; const long long ri = ai.r;
            if (prereadloc < (long long)qi) {
                if (!lc && ((double)opcount / (double)i) > max_factor) { bailed = true; break; }     // :21757 max_factor (1000; a test hook lowers it)
                for (int k = testspace_en; k < i; ++k) index_anchor(k);
                testspace_en = i;
                prereadloc = qi;
            }
            const double dli = (double)li;
            double max_scores = dli; int pre_index = VMX_NOPRE;
            const int ncand = hot;                    // the candidates are S_arg[:testspace_en]: its hot part (the cold part is out of every scan's reach)
            for (int bs = ncand - 1; bs >= 0; bs -= 64) {
                const int x = bs - lane;
                const bool valid = x >= 0;
                int j = 0; double Sj = 0.0; double test = -1e300;
                if (valid) {
                    j = SA[x]; Sj = S[j];
                    const vmx_anchor aj = A[j];
                    long long readgap, refgap, bonus;
                    vmx_link_geometry(qi, ri, si, li, aj.q, aj.r, (int)aj.s, (int)aj.l & 0xffff, readgap, refgap, bonus);
                    long long gapcost = readgap - refgap; if (gapcost < 0) gapcost = -gapcost;
                    if (si == (int)aj.s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                        test = Sj + (double)bonus - s_gapcost[gapcost];
                        if (lc) test = test - (double)tab.readgap_r[readgap];
                    } else {
                        test = Sj - skipcost + (double)bonus - vmx_extra_cost(tab, gapcost);
                    }
                }
                const double incl = vmx_wave_incl_max_f64(test);
                double m_before = vmx_wave_shr1_f64_fill(incl, VMX_F64_NEG);
                m_before = m_before > max_scores ? m_before : max_scores;
                const bool brk = !valid || !(Sj > (m_before - dli));
                const unsigned long long mask = __ballot(brk);
                const int first = mask ? (__ffsll((unsigned long long)mask) - 1) : 64;
                opcount += first;
                if (first > 0) {
                    const double Mx = vmx_readlane_f64(incl, first - 1);
                    if (Mx > max_scores) {
                        const unsigned long long em = __ballot(test == Mx) & (first >= 64 ? ~0ULL : ((1ULL << first) - 1ULL));
                        const int wl = __ffsll((unsigned long long)em) - 1;
                        pre_index = vmx_readlane(j, wl);
                        max_scores = Mx;
                    }
                }
                if (first < 64) break;
            }
            if (lane == 0) { S[i] = max_scores; P[i] = pre_index; }
            if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
            __syncthreads();
        }
        if (!bailed) for (int k = testspace_en; k < n; ++k) index_anchor(k);
        if (lane == 0) {
            J.ran = 1; J.n = n; J.hot = hot; J.n_cold = n_cold; J.cold_max = cold_max; J.gmax = bailed ? -1 : g_max_index; J.opcount = opcount;
        }
        __syncthreads();
    }
}

// :23250-23272 for one contig after its batch. One wavefront per job.
__global__ void __launch_bounds__(64) k_link_carry(vmx_link_job* __restrict__ jobs, int n_jobs, double skipcost) {
    const int lane = vmx_lane();
    for (int jb = (int)blockIdx.x; jb < n_jobs; jb += (int)gridDim.x) {
        vmx_link_job& J = jobs[jb];
        vmx_link_state& ST = *J.state;
        if (!J.ran || ST.status != 0) continue;
        const int base = J.cap_pre - ST.n_pre;
        const vmx_anchor* A = J.rows + base;
        const double* S = J.S + base; const int32_t* P = J.P + base; const int32_t* SA = J.SA;
        const int n = J.n, hot = J.hot;
        if (J.gmax == -2) { if (lane == 0) ST.status = VM_LINK_RAISED; continue; }        // the linked GC-fast ran into the reference's IndexError
        if (J.gmax < 0) { if (lane == 0) ST.status = VM_LINK_BAILED; continue; }         // GC-exact bailed out: the host re-runs the batch with k_chain_linked_fast
        if (lane == 0) { ST.pre_g_max_index = (int)J.gmax; ST.have = 1; ST.last_base = base; ST.last_n = n; }
        if (P[J.gmax] < 0) { if (lane == 0) J.saved = 0; continue; }                        // :23250 `continue`: nothing carried, nothing saved
        if (n - 1 <= 0) { if (lane == 0) ST.status = VM_LINK_RAISED; continue; }            // raise Exception("ERROR: ") :23266
        const double top = S[SA[hot - 1]];
        const double lowest = top - skipcost - 36 - 20;
        // sliceiloc: walk down from the top while lowestscores < S[S_arg[sliceiloc]]; it stops at the first entry at or below that score, or at
        // index 0 of the FULL index. A literal walk, 64 entries per step (after the linked GC-fast the index is ordered by (int(S), diagonal), not by S)
        int slice = -1;
        for (int bs = hot - 1; bs >= 0 && slice < 0; bs -= 64) {
            const int x = bs - lane;
            const bool stop = x >= 0 && !(lowest < S[SA[x]]);
            const unsigned long long m = __ballot(stop);
            if (m) slice = bs - (__ffsll((unsigned long long)m) - 1);
        }
        bool ok = true;
        if (slice >= 0) { if (J.n_cold > 0 && !(S[SA[slice]] > J.cold_max)) ok = false; }
        else { if (J.n_cold > 0) ok = false; slice = 0; }                          // the walk would run into the cold entries / stops at index 0
        const int n_carry = hot - slice;
        if (!ok || n_carry > ST.cap_pre) { if (lane == 0) ST.status = VM_LINK_UNSUPPORTED; continue; }
        const double bs = S[SA[slice]];
        int prl = 0;
        for (int x = lane; x < n_carry; x += 64) {
            const int j = SA[slice + x];
            ST.pre_S[x] = S[j] - bs + 1000;
            ST.pre_P[x] = -P[j];
            ST.pre_rows[x] = A[j];
            const int q = A[j].q; prl = q > prl ? q : prl;
        }
        prl = vmx_wave_max_i32(prl);
        __syncthreads();
        if (lane == 0) {
            ST.n_pre = n_carry; ST.g_max_index = n_carry - 1; ST.g_max_scores = ST.pre_S[n_carry - 1]; ST.prereadloc = prl;
            J.saved = 1; ST.n_saved += 1;
        }
        __syncthreads();
    }
}

// the carried rows of a contig's state go in front of its next batch (right-aligned below cap_pre)
__global__ void k_link_place(vmx_link_job* __restrict__ jobs, int n_jobs) {
    for (int jb = (int)blockIdx.x; jb < n_jobs; jb += (int)gridDim.x) {
        vmx_link_job& J = jobs[jb];
        const vmx_link_state& ST = *J.state;
        const int n_pre = ST.n_pre, base = J.cap_pre - n_pre;
        for (int i = (int)threadIdx.x; i < n_pre; i += (int)blockDim.x) J.rows[base + i] = ST.pre_rows[i];
    }
}

// a batch's anchors in read order (yield_mapinfo's one_mapinfo[np.argsort(one_mapinfo[:, 0])], :22431 — stable): keys for the device-wide radix sort
// (stable, rocPRIM through vmx_index_prim.h) and the gather by the sorted index
__global__ void k_link_sort_keys(const vmx_anchor* __restrict__ rows, int64_t n, uint64_t* __restrict__ key, uint64_t* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { key[i] = (uint64_t)(uint32_t)rows[i].q; idx[i] = (uint64_t)i; }
}
__global__ void k_link_sort_gather(const vmx_anchor* __restrict__ rows, const uint64_t* __restrict__ idx, int64_t n, vmx_anchor* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = rows[idx[i]];
}
