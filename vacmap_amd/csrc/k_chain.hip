// k_chain.hip — global non-linear (SV-aware) anchor chaining on gfx950.
//
//   k_flip_sort      S2 get_reversed_chain_numpy_rough (/root/reference/src/vacmap/mammap_clrnano.py:21202-21217)
//                    + the stable argsort by read position of hit2work_1 (:23572)
//   k_chain_global   G2 get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all (:24828-25031), "GC-exact":
//                    one wavefront per read; S and the score-sorted index S_arg live in LDS (12 B per anchor) when they
//                    fit, else in HBM; anchors and coverage are streamed from HBM through 64-anchor register blocks, the
//                    64 best predecessors sit in a register window. The candidate scan visits 64 predecessors per wave step in
//                    descending-S order; an exclusive prefix-max recovers the exact sequential break index, `opcount`
//                    and the strict-'>' winner.
//   k_chain_select   G1 hit2work_1 peel / primary / MAPQ / secondaries (:23588-23707) + decode_hit (:23981-24020):
//                    one wavefront per read: arrays staged in LDS, lane 0 runs the serial peel.
// Scores are IEEE double in the reference's evaluation order; the library is compiled with -ffp-contract=off.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_select.h"
#include "vmx_link.h"

// ------------------------------------------------------------------------------------------------ block bitonic sort
__device__ void vmx_block_bitonic_u64(uint64_t* a, int N) {   // N power of two, all threads of the block call it
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t x = a[i], y = a[ixj];
                    bool asc = (i & k) == 0;
                    if ((x > y) == asc) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ S2 + sort by q
// rows: int64 (q, r, s, l) as produced by map(); keys: scratch of pow2(n) uint64 per read at key_off[r]
__global__ void k_flip_sort(const int64_t* __restrict__ rows, const int64_t* __restrict__ aoff, const int64_t* __restrict__ readlens,
                            int n_reads, uint64_t* __restrict__ key_pool, const int64_t* __restrict__ key_off,
                            vmx_anchor* __restrict__ sorted, int32_t* __restrict__ need_reverse) {
    __shared__ int s_cnt[2];
    __shared__ uint64_t s_sort[VMX_SORT_LDS / 2];            // the sort's LDS tile (round 4: the plain bitonic passes ran through HBM — 78 barriers and passes for a HiFi read's
                                                             // 3000 anchors). 16 KB: an ONT read's ~750 anchors fit; with 32 KB the workgroups queued for LDS behind the other batches' kernels
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const int64_t* A = rows + 4 * aoff[r];
        const int n = (int)(aoff[r + 1] - aoff[r]);
        const int64_t L = readlens[r];
        if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
        __syncthreads();
        int neg = 0, pos = 0;
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) { int64_t s = A[4 * i + 2]; if (s == -1) ++neg; else if (s == 1) ++pos; }
        neg = vmx_wave_sum_i32(neg); pos = vmx_wave_sum_i32(pos);
        if (vmx_lane() == 0) { atomicAdd(&s_cnt[0], neg); atomicAdd(&s_cnt[1], pos); }
        __syncthreads();
        const bool flip = n >= 3 && s_cnt[0] > s_cnt[1];
        int N = 1; while (N < n) N <<= 1;
        uint64_t* keys = key_pool + key_off[r];
        for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
            uint64_t k = ~0ULL;
            if (i < n) {
                int64_t q = A[4 * i], l = A[4 * i + 3];
                if (flip) { q = L - q - l; k = ((uint64_t)q << 32) | (uint64_t)(n - 1 - i); }
                else k = ((uint64_t)q << 32) | (uint64_t)i;
            }
            keys[i] = k;
        }
        __syncthreads();
        if (N > 1) vmx_block_sort_u64_tiled(keys, N, s_sort, VMX_SORT_LDS / 2);
        vmx_anchor* out = sorted + aoff[r];
        for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
            uint64_t k = keys[i];
            int src = (int)(k & 0xffffffffu);
            if (flip) src = n - 1 - src;
            vmx_anchor a;
            a.q = (int32_t)(k >> 32);
            a.l = (int16_t)A[4 * src + 3];
            a.s = (int16_t)(flip ? -A[4 * src + 2] : A[4 * src + 2]);
            a.r = A[4 * src + 1];
            out[i] = a;
        }
        if (threadIdx.x == 0) need_reverse[r] = flip ? 1 : 0;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ G2 GC-exact
// :19369-19387 literal (equal keys return mid+1 at the first probe that hits one)
__device__ __forceinline__ int vmx_insertpoint_score(const double* S, double target, int k, const int* SA) {
    int i = 0, j = k;
    if (S[SA[0]] > target) return 0;
    if (S[SA[k - 1]] < target) return k;
    while (i < j) {
        int mid = (i + j) >> 1;
        double now = S[SA[mid]];
        if (now < target) i = mid + 1;
        else if (now > target) j = mid;
        else return mid + 1;
    }
    return j;
}

// The same insertion point without the dependent probe chain. On the sorted index the literal loop above only sees three outcomes
// per probe: mid < a (score below target), mid >= b (above), a <= mid < b (equal: return mid + 1), with a = #scores < target and
// b = #scores <= target. a and b come from two wave-cooperative 64-ary searches; the probe sequence is then replayed on (a, b)
// in scalar registers, so ties land exactly where the reference's bisection puts them.
__device__ __forceinline__ int vmx_insertpoint_score_wave(const double* S, double target, int k, const int* SA, int lane) {
    const int a = vmx_sorted_count(S, SA, k, target, false, lane);
    int b = a;
    if (a < k && S[SA[a]] == target) b = vmx_sorted_count(S, SA, k, target, true, lane);
    int i = 0, j = k;
    while (i < j) {
        const int mid = (i + j) >> 1;
        if (mid < a) i = mid + 1;
        else if (mid >= b) j = mid;
        else return mid + 1;
    }
    return j;
}

// The same (a, b) found by walking the index down from its top, 64 entries per step: in the linked DPs of -mode asm the index holds 10^5..10^6
// entries and a new one lands a few dozen to a few hundred places below the top (every anchor, noise hits included, hangs itself onto the best
// chain at the skip penalty), so one or two steps replace the four dependent rounds of the 64-ary search over the whole index.
__device__ __forceinline__ int vmx_insertpoint_score_topdown(const double* S, double target, int k, const int* SA, int lane) {
    int a = 0;
    for (int base = k - 1; base >= 0; base -= 64) {
        const int x = base - lane;
        const bool lt = x >= 0 && S[SA[x]] < target;
        const unsigned long long m = __ballot(lt);
        if (m) { a = base - (__ffsll((unsigned long long)m) - 1) + 1; break; }
    }
    int b = a;
    if (a < k && S[SA[a]] == target) {
        b = k;
        for (int base = a; base < k; base += 64) {
            const int x = base + lane;
            const bool gt = x < k && S[SA[x]] > target;
            const unsigned long long m = __ballot(gt);
            if (m) { b = base + (__ffsll((unsigned long long)m) - 1); break; }
        }
    }
    int i = 0, j = k;
    while (i < j) {
        const int mid = (i + j) >> 1;
        if (mid < a) i = mid + 1;
        else if (mid >= b) j = mid;
        else return mid + 1;
    }
    return j;
}

// shared gap geometry of GC and LC (:24953-24984, :27418-27456)
// (vmx_gap_geometry_sel: vmx_kernels.h)
__device__ __forceinline__ void vmx_gap_geometry(int qi, long long ri, int si, int li, int qj, long long rj, int sj, int lj,
                                                 long long& readgap, long long& refgap, long long& bonus) {
    vmx_gap_geometry_sel<false>(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
}
__device__ __forceinline__ void vmx_gap_geometry_asm(int qi, long long ri, int si, int li, int qj, long long rj, int sj, int lj,
                                                     long long& readgap, long long& refgap, long long& bonus) {
    vmx_gap_geometry_sel<true>(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
}

// One read. IN_LDS is a compile-time switch so that the working arrays are plain LDS pointers (ds_read / ds_write) in the instantiation the
// buckets run and plain global pointers in the other; a run-time choice between the two would turn every access into a flat_load.
// VAR: 0 = modes H / L / S, 1 = mode R, 2 = -mode asm (mammap_asm.py:20551-20737: no coverage terms, vmx_gap_geometry_asm, H's scoring)
// LINK defers the wait for its own stores: the common path of an anchor reads nothing it has written (scores of the current group, window and
// anchor blocks are registers), so the wave does not stop at every anchor for its S / P / S_arg stores to be acknowledged; the rare paths that
// do read them back (an insertion or a scan through the index in HBM, a group that straddles an anchor block) wait first. The emulator's
// fibers need the barrier at every anchor (its lanes run one after the other between collectives).
#ifdef VMX_EMU
#define VMX_LINK_WAIT() ((void)0)
#define VMX_LINK_STEP_BARRIER(link) __syncthreads()
#else
#define VMX_LINK_WAIT() __syncthreads()
#define VMX_LINK_STEP_BARRIER(link) do { if (!(link)) __syncthreads(); } while (0)
#endif
// LINK (with VAR 2, IN_LDS false): the batch-LINKED forms of -mode asm (mammap_asm.py:21686-21870 GC-exact, :21504-21685 LC): the first lk->n_pre
// rows carry S / P from the previous batch (already in S_out / P_out), the index starts with row 0 alone, the loop behind the carried rows,
// the running maximum and prereadloc come from the caller; lk->lc: co-linear steps also pay readgapcost_list[readgap], no bail-out
struct vmx_link_in { int n_pre; double g_max_scores; int g_max_index; long long prereadloc; int lc; double max_factor; };
template <bool IN_LDS, int VAR, bool LINK = false>
__device__ __forceinline__ void vmx_chain_global_read(const vmx_anchor* __restrict__ anchors, int rd, int64_t a0, int n, long long rmin, char* smem,
                                                      const double* s_gapcost, int lds_cap, const vmx_tables& tab, double oskipcost, int omaxdiff,
                                                      int maxgap, double* __restrict__ S_out, int32_t* __restrict__ P_out,
                                                      int32_t* __restrict__ SA_out, uint8_t* __restrict__ cov_pool,
                                                      int64_t* __restrict__ gmax_out, int64_t* __restrict__ opcount_out,
                                                      double* __restrict__ FP_pool, double* __restrict__ PP_pool, const vmx_link_in* lk = nullptr) {
    const int lane = vmx_lane();
    constexpr bool in_lds = IN_LDS;
    constexpr bool rmode = VAR == 1;
    constexpr bool nocov = VAR != 0;
    {
        const vmx_anchor* A = anchors + a0;
        // working arrays: LDS when the read fits, else straight in the HBM output arrays
        // LDS layout (VMX_GC_BYTES_PER_ANCHOR = 12): S f64 | S_arg i32 — only what the scan may touch at random. Anchors and coverage stay in
        // HBM: the loop takes them in order from 64-anchor register blocks (one coalesced load per 64 steps, a block ahead; the current anchor
        // comes out by v_readlane), the candidates' fields live in the register window (vmx_cwin), and the rare paths (an insertion below the
        // window or among equal scores, a scan past 64 candidates) read A[j] directly. P (written once per anchor) goes to its HBM output array.
        (void)rmin; (void)in_lds;
        double* S; int* SA; uint8_t* COV = nocov ? nullptr : cov_pool + a0;
        int* P = P_out + a0;
        if constexpr (IN_LDS) { S = (double*)smem; SA = (int*)(S + lds_cap); }
        else { S = S_out + a0; SA = SA_out + a0; }
        // coverage (number of anchors sharing the read position, capped at 20: :24865-24868)
        if constexpr (!nocov) for (int i = lane; i < n; i += 64) {
            vmx_anchor a = A[i];
            int c = 1;
            for (int x = i - 1; x >= 0 && A[x].q == a.q && c < 20; --x) ++c;
            for (int x = i + 1; x < n && A[x].q == a.q && c < 20; ++x) ++c;
            COV[i] = (uint8_t)c;
        }
        __syncthreads();
#define AQ(i) (A[i].q)
#define AR(i) ((long long)A[i].r)
#define AL(i) ((int)A[i].l & 0xffff)
#define AS(i) ((int)A[i].s)
        int i0 = 1;                       // first anchor the loop computes (LINK: behind the carried rows)
        if constexpr (LINK) { if (lk->n_pre > 0) i0 = lk->n_pre; }
        const bool carried = LINK && i0 > 1;
        int prereadloc = carried ? (int)lk->prereadloc : AQ(0);
        // rmode: mode R's body (mammap_noprefercloser.py:22839-23057) has no coverage terms; a non-co-linear step costs the fixed skipcost,
        // remembered per anchor in fixed_penatly / pre_penatly (FP / PP, in HBM) and refunded after skipcost co-linear bases
        double* FP = rmode ? FP_pool + a0 : nullptr; double* PP = rmode ? PP_pool + a0 : nullptr;
        double skipcost = nocov ? oskipcost : oskipcost + (double)COV[0];
        int maxdiff = nocov ? omaxdiff : omaxdiff - (int)COV[0]; if (maxdiff < 10) maxdiff = 10;
        int testspace_en = 1;
        if (lane == 0) { SA[0] = 0; if (!carried) { S[0] = (double)AL(0); P[0] = VMX_NOPRE; } if (rmode) { FP[0] = 0.0; PP[0] = 0.0; } }
        __syncthreads();
        double g_max_scores = carried ? lk->g_max_scores : (double)AL(0); int g_max_index = carried ? lk->g_max_index : 0;
        long long opcount = 0;
        long long dbg_fallback = 0, dbg_block2 = 0, dbg_groups = 0;      // LINK: tuning counters (insertions through HBM, scans past the window, position advances)
        bool bailed = false;
        // candidate window: the testspace_en entries of S_arg, best first, the top 64 of them in registers
        double gS = 0.0; int gbase = 0x7fffffff;      // LINK: score of the (lane)-th anchor computed since the last position advance (anchor gbase + lane)
        vmx_cwin win; win.j = 0; win.q = AQ(0); win.ls = AL(0) | (AS(0) << 16); win.S = carried ? S[0] : (double)AL(0); win.r = AR(0);
        // anchors [bb, bb + 64) in registers (lane t: anchor bb + t), the next block already on its way
        int bq = 0, bls = 0, bcov = 0, nbq = 0, nbls = 0, nbcov = 0; long long br = 0, nbr = 0;
        const int cb = (i0 - 1) & ~63;    // the block anchor i0 - 1 lies in (0 unless LINK)
        { const int x = cb + lane < n ? cb + lane : n - 1; const vmx_anchor a = A[x]; bq = a.q; bls = ((int)a.l & 0xffff) | ((int)a.s << 16); br = a.r; bcov = nocov ? 0 : COV[x]; }
        { const int x = cb + 64 + lane < n ? cb + 64 + lane : n - 1; const vmx_anchor a = A[x]; nbq = a.q; nbls = ((int)a.l & 0xffff) | ((int)a.s << 16); nbr = a.r; nbcov = nocov ? 0 : COV[x]; }
        int pq = win.q, pls = win.ls; long long pr = win.r; double pS = win.S;      // anchor i-1 and its score
        if (carried) { const int x = i0 - 1; pq = AQ(x); pls = AL(x) | (AS(x) << 16); pr = AR(x); pS = S[x]; }
        for (int i = i0; i < n; ++i) {
            if ((i & 63) == 0) {
                bq = nbq; bls = nbls; br = nbr; bcov = nbcov;
                const int x = i + 64 + lane < n ? i + 64 + lane : n - 1; const vmx_anchor a = A[x]; nbq = a.q; nbls = ((int)a.l & 0xffff) | ((int)a.s << 16); nbr = a.r; nbcov = nocov ? 0 : COV[x];
            }
            // the current anchor is the same in every lane: scalar registers, scalar branches on its strand
            const int bl = i & 63, bb = i & ~63;
            const int qi = vmx_readlane(bq, bl); const int lsi = vmx_readlane(bls, bl); const int li = lsi & 0xffff, si = lsi >> 16;
            long long ri; { union { long long d; int w[2]; } u; u.d = br; u.w[0] = vmx_readlane(u.w[0], bl); u.w[1] = vmx_readlane(u.w[1], bl); ri = u.d; }
            if (prereadloc < qi) {
                if (LINK ? (!lk->lc && ((double)opcount / (double)i) > lk->max_factor) : (((double)opcount / (double)i) > 1000.0)) { bailed = true; break; }   // :24914 max_factor
                for (int k = testspace_en; k < i; ++k) {
                    double Sk; int qk, lsk; long long rk;
                    if (k == i - 1) { Sk = pS; qk = pq; lsk = pls; rk = pr; }
                    else if (k >= bb) {
                        // (LINK: the scores of the anchors computed since the last advance wait in a register, lane = place in that group — several
                        // noise hits share a read position there, and S[k] would be a load of a value this wave stored a moment ago)
                        const int kl = k - bb;
                        // (a value that does come from memory is handed over in scalar registers: the wait for the load then sits inside this rare
                        // branch. Left in vector registers, the compiler has to wait at the merge point — for EVERY outstanding memory operation,
                        // i.e. for the acknowledgement of the S / P / S_arg stores of the anchors before, at every insertion)
                        if (LINK && k >= gbase && k - gbase < 64) Sk = vmx_readlane_f64(gS, k - gbase); else { if constexpr (LINK) VMX_LINK_WAIT(); Sk = IN_LDS ? S[k] : vmx_uniform_f64(S[k]); }
                        qk = vmx_readlane(bq, kl); lsk = vmx_readlane(bls, kl);
                        union { long long d; int w[2]; } u; u.d = br; u.w[0] = vmx_readlane(u.w[0], kl); u.w[1] = vmx_readlane(u.w[1], kl); rk = u.d;
                    } else {
                        if constexpr (LINK) VMX_LINK_WAIT();
                        Sk = S[k]; qk = AQ(k); lsk = AL(k) | (AS(k) << 16); rk = AR(k);
                        if constexpr (!IN_LDS) { Sk = vmx_uniform_f64(Sk); qk = vmx_uniform_i32(qk); lsk = vmx_uniform_i32(lsk); rk = vmx_uniform_i64(rk); }
                    }
                    // the reference's bisection (:19369-19387) puts a score without an equal behind all smaller ones; among equals the place
                    // depends on its probe sequence (vmx_insertpoint_score_wave): the window handles the first case
                    const int W = k < 64 ? k : 64;               // k entries so far
                    const unsigned long long gt = __ballot(lane < W && win.S > Sk), ge = __ballot(lane < W && win.S >= Sk);
                    const int above = __popcll(gt);
                    if constexpr (LINK) {
                        // Equal scores are the rule here, not the exception: against an hg38-size reference nine in ten anchors are noise hits, each
                        // hangs itself onto the best chain at skipcost + extra's last value (36), and all that do so between two steps of the chain
                        // score S_top - 51 + l exactly. When the run of equal scores ends inside the window its bounds (#entries below, #not above)
                        // follow from the two ballots, the reference's bisection is replayed on them in scalar registers, and the entry is placed
                        // among its equals in the window — no walk through the index in HBM.
                        const int cge = __popcll(ge);
                        if (above < 64 && (cge < W || W == k)) {
                            int at = above;
                            if (gt != ge) {
                                const int a = k - cge, b = k - above;
                                // The run of equals ends within 64 places of the top, so the bisection over [0, k) first goes right log2(k / cge)
                                // times in a row. Those steps have a closed form: with g = k - i, a step to the right is g <- (g - 1) >> 1, i.e.
                                // g_t + 1 = (k + 1) >> t, and step t is taken while its probe k - 1 - g_(t+1) lies below a, i.e. while
                                // (k + 1) >> (t + 1) >= cge + 1. The loop below starts behind them (it was ~20 scalar rounds per insertion).
                                int i2, j2 = k, loc = -1;
                                {
                                    const unsigned n1 = (unsigned)k + 1u, c1 = (unsigned)cge + 1u;
                                    int T = (31 - __builtin_clz(n1)) - (31 - __builtin_clz(c1));
                                    if ((n1 >> T) < c1) --T;
                                    i2 = k + 1 - (int)(n1 >> T);
                                }
                                while (i2 < j2) { const int mid = (i2 + j2) >> 1; if (mid < a) i2 = mid + 1; else if (mid >= b) j2 = mid; else { loc = mid + 1; break; } }
                                if (loc < 0) loc = j2;
                                at = k - loc;
                            }
                            if (at < 64) {
                                vmx_cwin_insert(win, at, k, Sk, qk, lsk, rk, lane);
                                if (lane <= at) SA[k - lane] = win.j;
                                continue;
                            }
                        }
                    }
                    if (gt == ge && above < 64 && (above < W || W == k)) {
                        vmx_cwin_insert(win, above, k, Sk, qk, lsk, rk, lane);
                        if (lane <= above) SA[k - lane] = win.j;
                    } else {
                        ++dbg_fallback;
                        if constexpr (LINK) VMX_LINK_WAIT();
                        const int loc = LINK ? vmx_insertpoint_score_topdown(S, Sk, k, SA, lane) : vmx_insertpoint_score_wave(S, Sk, k, SA, lane);
                        vmx_sarg_insert4(SA, loc, k, lane);
                        // (an entry that lands below the 64 of the window leaves the window as it is)
                        if ((!LINK || k - loc < 64) && lane <= k) { const int j = SA[k - lane]; win.j = j; win.S = S[j]; win.q = AQ(j); win.ls = AL(j) | (AS(j) << 16); win.r = AR(j); }
                    }
                }
                testspace_en = i; ++dbg_groups; gbase = i;
                if (!nocov) {
                    const int covi = vmx_readlane(bcov, bl);
                    skipcost = oskipcost + (double)covi;
                    maxdiff = omaxdiff - covi; if (maxdiff < 10) maxdiff = 10;
                }
                prereadloc = qi;
            }
            const double dli = (double)li;
            double max_scores = dli; int pre_index = VMX_NOPRE;
            double fp_i = 0.0, pp_i = 0.0;                       // rmode: the winner's fixed_penatly / pre_penatly
            for (int base = testspace_en - 1; base >= 0; base -= 64) {
                const int x = base - lane;
                const bool valid = x >= 0;
                int j = 0; double Sj = 0.0; double test = -1e300;
                double nfp = 0.0, npp = 0.0;                     // rmode: what i inherits if this candidate wins
                if (valid) {
                    int qj, lj, sj; long long rj;
                    if (base == testspace_en - 1) { j = win.j; Sj = win.S; qj = win.q; lj = win.ls & 0xffff; sj = win.ls >> 16; rj = win.r; }   // first 64: registers
                    else { j = SA[x]; Sj = S[j]; qj = AQ(j); lj = AL(j); sj = AS(j); rj = AR(j); if (lane == 0) ++dbg_block2; }      // (LINK: the wait is in front of the loop's second turn, below)
                    long long readgap, refgap, bonus;
                    if constexpr (VAR == 2) vmx_gap_geometry_asm(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
                    else vmx_gap_geometry(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
                    long long gapcost = readgap - refgap; if (gapcost < 0) gapcost = -gapcost;
                    if (rmode) {
                        if (si == sj && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                            test = Sj + (double)bonus - s_gapcost[gapcost];
                            const double fpj = FP[j], ppj = PP[j];
                            if (fpj < 0 && (fpj + (double)bonus) >= 0) test += ppj;
                            if (fpj < 0 && (fpj + (double)bonus) < 0) { nfp = fpj + (double)bonus; npp = ppj; }
                        } else {
                            test = Sj + (double)bonus - skipcost;
                            nfp = -skipcost + (double)bonus; npp = skipcost;
                        }
                    } else {
                        // both forms are computed and one is selected: a wave's candidates are a mix of co-linear and other steps, and as two
                        // branches each paid its own exec-mask region around a handful of adds
                        const bool col = si == sj && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff;
                        double tc = Sj + (double)bonus - s_gapcost[col ? (int)gapcost : 0];
                        if constexpr (LINK) { if (lk->lc) tc = tc - (double)tab.readgap_r[col ? readgap : 0]; }      // :21644
                        const double tn = Sj - skipcost + (double)bonus - vmx_extra_cost(tab, col ? 0x7fffffffffffffffLL : gapcost);
                        test = col ? tc : tn;
                    }
                }
                const double incl = vmx_wave_incl_max_f64(test);                 // prefix max of the candidates' scores, in scan order
                double m_before = vmx_wave_shr1_f64_fill(incl, VMX_F64_NEG);
                m_before = m_before > max_scores ? m_before : max_scores;         // the running max the sequential loop holds at this candidate
                const bool brk = !valid || !(Sj > (m_before - dli));
                const unsigned long long mask = __ballot(brk);
                const int first = mask ? (__ffsll((unsigned long long)mask) - 1) : 64;
                opcount += first;
                if (first > 0) {
                    const double M = vmx_readlane_f64(incl, first - 1);          // best score among the candidates before the break
                    if (M > max_scores) {                                        // strict >: the first (highest-S) candidate reaching M wins
                        const unsigned long long em = __ballot(test == M) & (first >= 64 ? ~0ULL : ((1ULL << first) - 1ULL));
                        const int wl = __ffsll((unsigned long long)em) - 1;
                        pre_index = vmx_readlane(j, wl);
                        if (rmode) { fp_i = vmx_readlane_f64(nfp, wl); pp_i = vmx_readlane_f64(npp, wl); }
                        max_scores = M;
                    }
                }
                if (first < 64) break;
                if constexpr (LINK) VMX_LINK_WAIT();              // the next block of candidates comes from the index in HBM
            }
            if (lane == 0) { S[i] = max_scores; P[i] = pre_index; if (rmode) { FP[i] = fp_i; PP[i] = pp_i; } }
            if constexpr (LINK) { if (i - gbase < 64 && lane == i - gbase) gS = max_scores; }
            if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
            pS = max_scores; pq = qi; pls = lsi; pr = ri;
            if constexpr (IN_LDS && VAR != 1) vmx_wave_lds_fence(); else VMX_LINK_STEP_BARRIER(LINK);     // (mode R: FP / PP go through HBM)
        }
        __syncthreads();
        if (!bailed) {
            for (int k = testspace_en; k < n; ++k) {
                const int loc = vmx_insertpoint_score_wave(S, S[k], k, SA, lane);
                vmx_sarg_insert4(SA, loc, k, lane);
            }
        }
        if constexpr (IN_LDS) {
            for (int i = lane; i < n; i += 64) { S_out[a0 + i] = S[i]; SA_out[a0 + i] = SA[i]; }
        }
        if (lane == 0) { gmax_out[rd] = bailed ? -1 : g_max_index; opcount_out[rd] = opcount; if constexpr (LINK) { opcount_out[rd + 1] = dbg_fallback; opcount_out[rd + 2] = dbg_block2; opcount_out[rd + 3] = dbg_groups; } }
        __syncthreads();
#undef AQ
#undef AR
#undef AL
#undef AS
    }
}

__global__ void __launch_bounds__(64) k_chain_global(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ aoff,
                                                     const int32_t* __restrict__ rlist, int nlist, int lds_cap, vmx_tables tab,
                                                     const double* __restrict__ gapcost_list, double oskipcost, int omaxdiff,
                                                     int maxgap, double* __restrict__ S_out, int32_t* __restrict__ P_out,
                                                     int32_t* __restrict__ SA_out, uint8_t* __restrict__ cov_pool,
                                                     int64_t* __restrict__ gmax_out, int64_t* __restrict__ opcount_out, int rmode,
                                                     double* __restrict__ FP_pool, double* __restrict__ PP_pool) {
    VMX_SETPRIO(3);
    VMX_DYN_SHARED(char, smem);
    __shared__ double s_gapcost[64];
    const int lane = vmx_lane();
    for (int x = lane; x <= omaxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    for (int li_ = blockIdx.x; li_ < nlist; li_ += gridDim.x) {
        const int rd = rlist[li_];
        const int64_t a0 = aoff[rd];
        const int n = (int)(aoff[rd + 1] - a0);
        if (n <= 0) { if (lane == 0) { gmax_out[rd] = -2; opcount_out[rd] = 0; } continue; }
        const vmx_anchor* A = anchors + a0;
        const long long rmin = 0;
#define VMX_GC_CALL(L, R) vmx_chain_global_read<L, R>(anchors, rd, a0, n, rmin, smem, s_gapcost, lds_cap, tab, oskipcost, omaxdiff, maxgap, S_out, P_out, SA_out, cov_pool, \
                                                       gmax_out, opcount_out, FP_pool, PP_pool)
        // rmode: 0 / 1 / 2 = the VAR above
        if (n <= lds_cap) { if (rmode == 1) VMX_GC_CALL(true, 1); else if (rmode == 2) VMX_GC_CALL(true, 2); else VMX_GC_CALL(true, 0); }
        else { if (rmode == 1) VMX_GC_CALL(false, 1); else if (rmode == 2) VMX_GC_CALL(false, 2); else VMX_GC_CALL(false, 0); }
#undef VMX_GC_CALL
    }
}

// -mode asm, contigs of 500 kb and more: one batch of one contig per workgroup (one wavefront), the register-window form of k_chain_linked
// (k_chain_linked.hip keeps the plain form: VMX_LINK_PLAIN=1 selects it). Every entry is stored (no cold entries): hot = n.
__global__ void __launch_bounds__(64) k_chain_linked_win(vmx_link_job* __restrict__ jobs, int n_jobs, vmx_tables tab, const double* __restrict__ gapcost_list,
                                                        double skipcost, int maxdiff, int maxgap, int lc, double max_factor) {
    VMX_DYN_SHARED(char, smem);
    __shared__ double s_gapcost[64];
    const int lane = vmx_lane();
    for (int x = lane; x <= maxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    for (int jb = (int)blockIdx.x; jb < n_jobs; jb += (int)gridDim.x) {
        vmx_link_job& J = jobs[jb];
        vmx_link_state& ST = *J.state;
        if (ST.status != 0 || J.n_new <= 0) { if (lane == 0) J.ran = 0; continue; }
        // everything the loop branches on is the same in every lane; read from the job record it would live in vector registers and every
        // `if` / loop of the DP would be an exec-mask region instead of a scalar branch
        const int n_pre = vmx_uniform_i32(ST.n_pre), base = vmx_uniform_i32(J.cap_pre) - n_pre, n = n_pre + vmx_uniform_i32(J.n_new);
        double* S = VMX_GLOBAL_PTR(double, J.S + base); int32_t* P = VMX_GLOBAL_PTR(int32_t, J.P + base);
        int32_t* SAg = VMX_GLOBAL_PTR(int32_t, J.SA); const vmx_anchor* rows = VMX_GLOBAL_PTR(const vmx_anchor, J.rows + base);
        for (int i = lane; i < n_pre; i += 64) { S[i] = ST.pre_S[i]; P[i] = ST.pre_P[i]; }
        __syncthreads();
        vmx_link_in lk; lk.n_pre = n_pre; lk.g_max_scores = vmx_uniform_f64(ST.g_max_scores); lk.g_max_index = vmx_uniform_i32(ST.g_max_index);
        lk.prereadloc = vmx_uniform_i64(ST.prereadloc); lk.lc = lc; lk.max_factor = max_factor;
        int64_t gm = 0, opc = 0;
        __shared__ int64_t s_out[8];
        vmx_chain_global_read<false, 2, true>(rows, 0, 0, n, 0, smem, s_gapcost, 0, tab, skipcost, maxdiff, maxgap, S, P, SAg, nullptr, &s_out[0], &s_out[1], nullptr, nullptr, &lk);
        __syncthreads();
        gm = s_out[0]; opc = s_out[1];
        if (lane == 0) { J.ran = 1; J.n = n; J.hot = n; J.n_cold = 0; J.cold_max = -1e300; J.gmax = gm; J.opcount = opc;
                         if (J.dbg) { J.dbg[0] += n; J.dbg[1] += s_out[2]; J.dbg[2] += s_out[3]; J.dbg[3] += s_out[4]; J.dbg[4] += opc; } }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ G1 select
// One wavefront per read. `rlist` (nlist reads, nullptr = reads 0 .. nlist-1) is one size class: `lds_cap` anchors fit the dynamic LDS
// (17 B each: S, P, S_arg and the `used` flags of the serial peel; lds_cap = 0: everything stays in HBM). After the peel the same LDS
// holds the small arrays of the ranking step. Phases:
//   1 wave   stage S / P / S_arg, clear `used`
//   2 lane 0 the peel (dependent walks S_arg -> used -> P at LDS latency); chain lists go to the read's scratch in HBM (stores only)
//   3 wave   read positions of the chain nodes (cq[t] = A[cidx[t]].q: the loads lane 0 used to make one after the other at HBM latency)
//   4 lane 0 order / read bins / primaries / MAPQ / secondaries on LDS arrays
//   5 wave   copy of the selected paths (decode_hit :23981-24020)
// hit2work_1's peel (:23588-23640, vmx_select_peel) on the whole wavefront. The serial form visits the chain ends in descending S (S_arg) and
// walks each down its predecessors until it meets an anchor an earlier chain took: an anchor therefore belongs to the end with the highest
// priority (position in S_arg; the best chain's end above all) among the anchors whose predecessor path leads through it — the maximum of
// the priorities over its SUBTREE in the predecessor forest. That maximum, every anchor's depth and its root come out of ceil(log2(depth))
// rounds of pointer doubling (M'[J[a]] = max(M[J[a]], M[a]), J' = J o J, D' = D + D o J); a chain is then the stretch of one owner along
// a path, its last anchor ("bottom") the one whose predecessor has another owner, its score S[end] - S[predecessor of the bottom], and the
// serial order of the kept chains is descending priority: one scan over the priorities lays out cscore / coff, and every anchor writes
// itself to cidx at (its chain's offset) + (depth of the end) - (its depth). Same cscore / coff / cidx as the serial peel, bit for bit
// (the score is the same single subtraction). lds: 16 n bytes; n < 65535.
__device__ __forceinline__ int vmx_select_peel_wave(int n, const double* __restrict__ S, const int32_t* __restrict__ P, const int32_t* __restrict__ SA, int gmax, int mode,
                                                    char* lds, double* cscore, int* coff, int* cidx, int lane) {
    uint32_t* M = (uint32_t*)lds; uint32_t* Mn = M + n;
    unsigned short* J = (unsigned short*)(Mn + n); unsigned short* Jn = J + n; unsigned short* D = Jn + n; unsigned short* Dn = D + n;
    for (int x = lane; x < n; x += 64) M[SA[x]] = (uint32_t)x;
    __syncthreads();
    for (int a = lane; a < n; a += 64) {
        const uint32_t pr = a == gmax ? (uint32_t)n : M[a];
        const int p = P[a];
        M[a] = pr; Mn[a] = pr; J[a] = (unsigned short)(p == VMX_NOPRE ? a : p); D[a] = p == VMX_NOPRE ? 0 : 1;
    }
    __syncthreads();
    for (int round = 0; round < 17; ++round) {
        bool ch = false;
        for (int a = lane; a < n; a += 64) {
            const int j = J[a];
            if (j != a) atomicMax(&Mn[j], M[a]);
            const int jj = J[j];
            Jn[a] = (unsigned short)jj; Dn[a] = (unsigned short)(D[a] + D[j]); ch = ch || jj != j;
        }
        __syncthreads();
        for (int a = lane; a < n; a += 64) { M[a] = Mn[a]; J[a] = Jn[a]; D[a] = Dn[a]; }
        __syncthreads();
        if (!__any(ch)) break;
    }
    // bottoms: the last anchor of every owner's stretch (priority n = the best chain, whose stretch runs to a root)
    uint32_t* BOT = Mn;
    for (int a = lane; a < n; a += 64) BOT[a] = 0xffffffffu;
    __syncthreads();
    for (int a = lane; a < n; a += 64) {
        const int p = P[a]; const uint32_t o = M[a];
        if (o < (uint32_t)n && (p == VMX_NOPRE || M[p] != o)) BOT[o] = (uint32_t)a;
    }
    __syncthreads();
    const double scores = S[gmax];
    if (!(scores > 40)) return -1;                                   // hit == False: unmapped whatever follows
    const double accept = (mode == 0) ? 60.0 : 40.0;
    unsigned short* CH = Jn; unsigned short* CO = Dn;               // chain index / offset of a priority's chain (0xffff: dropped or none)
    int nch = 1, w = (int)D[gmax] + 1;
    if (lane == 0) { cscore[0] = scores; coff[0] = 0; }
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int pr = n - 1 - (i0 + lane);
        int keep = 0, len = 0; double sc = 0.0;
        if (pr >= 0) {
            const uint32_t b = BOT[pr];
            if (b != 0xffffffffu) {
                const int e = SA[pr]; const int pb = P[b];
                sc = S[e]; if (pb != VMX_NOPRE) sc = sc - S[pb];
                keep = sc > 40; len = (int)D[e] - (int)D[b] + 1;
            }
        }
        const int ik = vmx_wave_incl_scan_i32(keep), il = vmx_wave_incl_scan_i32(keep ? len : 0);
        if (pr >= 0) {
            if (keep) { const int c = nch + ik - 1, o = w + il - len; cscore[c] = sc; coff[c] = o; CH[pr] = (unsigned short)c; CO[pr] = (unsigned short)o; }
            else CH[pr] = 0xffff;
        }
        nch += __shfl(ik, 63); w += __shfl(il, 63);
    }
    if (lane == 0) coff[nch] = w;
    __syncthreads();
    const int dg = (int)D[gmax];
    for (int a = lane; a < n; a += 64) {
        const uint32_t o = M[a];
        if (o == (uint32_t)n) cidx[dg - (int)D[a]] = a;
        else if (CH[o] != 0xffff) cidx[(int)CO[o] + (int)D[SA[o]] - (int)D[a]] = a;
    }
    __syncthreads();
    const double max_scores = scores > 0 ? scores : 0;
    if (!(max_scores > accept)) return -1;
    return nch;
}

// the two loops of vmx_select_rank that touch every chain pair / every chain node, on the whole wavefront: order[] (descending score, equal
// scores in descending index = the stable argsort reversed, then "best chain first") and the chains' unique read-position bins (bins / boff)
__device__ __forceinline__ void vmx_select_order_bins_wave(int nch, int w, int mode, const double* cscore, const int* coff, const int* cq, int* order, int* bins, int* boff, int lane) {
    for (int c = lane; c < nch; c += 64) {
        const double sc = cscore[c]; int pos = 0;
        for (int d = 0; d < nch; ++d) { const double sd = cscore[d]; pos += (sd > sc || (sd == sc && d > c)) ? 1 : 0; }
        order[pos] = c;
    }
    __syncthreads();
    if (mode != 4 && lane == 0 && order[0] != 0) { for (int i = 0; i < nch; ++i) if (order[i] == 0) { order[i] = order[0]; order[0] = 0; break; } }
    // a node opens a bin when it is its chain's first or its bin differs from the node before it; bins are written in node order, and a
    // chain's boff is the number of bins opened before its first node
    int carry = 0;
    for (int t0 = 0; t0 < w; t0 += 64) {
        const int t = t0 + lane;
        int f = 0, b = 0, cs = -1;
        if (t < w) {
            b = cq[t] / 100;
            int lo = 0, hi = nch; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= t) lo = mid; else hi = mid; }      // the node's chain
            const bool start = coff[lo] == t;
            f = (start || cq[t - 1] / 100 != b) ? 1 : 0;
            if (start) cs = lo;
        }
        const int inc = vmx_wave_incl_scan_i32(f);
        const int at = carry + inc - f;
        if (f) bins[at] = b;
        if (cs >= 0) boff[cs] = at;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) boff[nch] = carry;
    __syncthreads();
}

__global__ void __launch_bounds__(64) k_chain_select(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ aoff, const int64_t* __restrict__ readlens,
                               const int32_t* __restrict__ rlist, int nlist, int lds_cap, const double* __restrict__ S, const int32_t* __restrict__ P, const int32_t* __restrict__ SA,
                               const int64_t* __restrict__ gmax, const int32_t* __restrict__ need_reverse, int mode,
                               char* __restrict__ scratch, const int64_t* __restrict__ scratch_off,
                               int32_t* __restrict__ out_mapq, double* __restrict__ out_score, int32_t* __restrict__ out_npaths,
                               int32_t* __restrict__ out_path_len, vmx_anchor* __restrict__ out_path_anchors) {
    VMX_SETPRIO(3);
    VMX_DYN_SHARED(char, s_buf);
    __shared__ int s_hdr[4];
    const int lane = vmx_lane();
    const int lds_bytes = lds_cap * 17 + 64;
    const bool serial_peel = readlens == nullptr;                    // (test knob: a null length array selects the one-lane peel)

    for (int x = (int)blockIdx.x; x < nlist; x += (int)gridDim.x) {
        const int r = rlist ? rlist[x] : x;
        const int64_t a0 = aoff[r];
        const int n = (int)(aoff[r + 1] - a0);
        const bool run = n > 2 && gmax[r] >= 0;
        if (!run) {
            if (lane == 0) { out_mapq[r] = 0; out_score[r] = need_reverse[r] ? -0.0 : 0.0; out_npaths[r] = 0; }
            continue;
        }
        const vmx_anchor* A = anchors + a0;
        const bool in_lds = n <= lds_cap;
        vmx_select_scr W = vmx_select_scratch(scratch + scratch_off[r], n);
        double* s_S = (double*)s_buf; int32_t* s_P = (int32_t*)(s_S + n); int32_t* s_SA = s_P + n; unsigned char* s_used = (unsigned char*)(s_SA + n);
        if (in_lds && n < 65535 && !serial_peel) { }                  // (the wave-parallel peel stages nothing)
        else if (in_lds) { for (int i = lane; i < n; i += 64) { s_S[i] = S[a0 + i]; s_P[i] = P[a0 + i]; s_SA[i] = SA[a0 + i]; s_used[i] = 0; } }
        else { for (int i = lane; i < n; i += 64) W.used[i] = 0; }
        __syncthreads();
        if (in_lds && n < 65535 && !serial_peel) {                   // the whole wavefront (arrays of the doubling rounds in LDS, S / P / S_arg read from HBM)
            const int nchw = vmx_select_peel_wave(n, S + a0, P + a0, SA + a0, (int)gmax[r], mode, s_buf, W.cscore, W.coff, W.cidx, lane);
            __syncthreads();
            if (lane == 0) { s_hdr[0] = nchw; s_hdr[1] = nchw > 0 ? W.coff[nchw] : 0; }
        } else if (lane == 0) {
            int nch;
            if (in_lds) nch = vmx_select_peel(n, s_S, s_P, s_SA, (int)gmax[r], mode, s_used, W.cscore, W.coff, W.cidx);
            else nch = vmx_select_peel(n, S + a0, P + a0, SA + a0, (int)gmax[r], mode, W.used, W.cscore, W.coff, W.cidx);
            s_hdr[0] = nch; s_hdr[1] = nch > 0 ? W.coff[nch] : 0;
        }
        __syncthreads();
        const int nch = s_hdr[0], w = s_hdr[1];
        __syncthreads();
        if (nch <= 0) {
            if (lane == 0) { out_mapq[r] = 0; out_score[r] = need_reverse[r] ? -0.0 : 0.0; out_npaths[r] = 0; }
            continue;
        }
        // the ranking step's arrays: in LDS (over the peel's, which are done with) when they fit, else in the read's HBM scratch
        const int need = 8 * nch + 4 * (5 * nch + 4) + 8 * w + 32;
        const bool fit = need <= lds_bytes && lds_cap > 0;
        double* l_cscore = (double*)s_buf; int* l_coff = (int*)(l_cscore + nch); int* l_order = l_coff + nch + 1; int* l_prim = l_order + nch; int* l_sec = l_prim + nch;
        int* l_boff = l_sec + nch; int* l_cq = l_boff + nch + 1; int* l_bins = l_cq + w;
        if (fit) {
            for (int c = lane; c < nch; c += 64) l_cscore[c] = W.cscore[c];
            for (int c = lane; c <= nch; c += 64) l_coff[c] = W.coff[c];
            for (int t = lane; t < w; t += 64) l_cq[t] = A[W.cidx[t]].q;
        } else {
            for (int t = lane; t < w; t += 64) W.cq[t] = A[W.cidx[t]].q;
        }
        __syncthreads();
        if (fit) vmx_select_order_bins_wave(nch, w, mode, l_cscore, l_coff, l_cq, l_order, l_bins, l_boff, lane);
        else vmx_select_order_bins_wave(nch, w, mode, W.cscore, W.coff, W.cq, W.order, W.bins, W.boff, lane);
        if (lane == 0) {
            int mapq = 0, nsec, pidx = 0;
            if (fit) nsec = vmx_select_rank(nch, mode, l_cscore, l_coff, l_cq, S + a0, W.cidx, l_order, l_bins, l_boff, l_prim, l_sec, &mapq, &pidx, true);
            else nsec = vmx_select_rank(nch, mode, W.cscore, W.coff, W.cq, S + a0, W.cidx, W.order, W.bins, W.boff, W.prim, W.sec, &mapq, &pidx, true);
            const double sc = fit ? l_cscore[pidx] : W.cscore[pidx];
            if (nsec == -2) {                          // -mode asm: decode_hit's edlib tie-break among equal chains is made by the host (vmx_asm_resolve_ties)
                out_mapq[r] = 0; out_score[r] = need_reverse[r] ? -0.0 : 0.0; out_npaths[r] = -7;
            } else { out_mapq[r] = mapq; out_score[r] = need_reverse[r] ? -sc : sc; out_npaths[r] = nsec + 1; }
            s_hdr[2] = nsec; s_hdr[3] = pidx;
        }
        __syncthreads();
        const int nsec = s_hdr[2], pidx = s_hdr[3];
        if (nsec < 0) continue;
        // decode_hit: return_path_list = [best path] + secondaries
        int wr = 0;
        for (int pi = 0; pi <= nsec; ++pi) {
            const int c = pi == 0 ? pidx : (fit ? l_sec[pi - 1] : W.sec[pi - 1]);
            const int t0 = fit ? l_coff[c] : W.coff[c], t1 = fit ? l_coff[c + 1] : W.coff[c + 1];
            if (lane == 0) out_path_len[a0 + pi] = t1 - t0;
            for (int t = t0 + lane; t < t1; t += 64) out_path_anchors[a0 + wr + (t - t0)] = A[W.cidx[t]];
            wr += t1 - t0;
        }
        __syncthreads();
    }
}
