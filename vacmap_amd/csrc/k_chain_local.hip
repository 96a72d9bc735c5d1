// k_chain_local.hip — local chain DP: LC-exact (get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list,
// /root/reference/src/vacmap/mammap_clrnano.py:27305-27528) and LC-mm (..._fine_list_mismatch, :28250-28476).
// One wavefront per read; S and the score-sorted index S_arg live in LDS (12 B per anchor) when they fit, else in HBM; the anchors
// (sorted by read END, :28585) are streamed from HBM through 64-anchor register blocks and the 64 best predecessors are kept in a
// register window; reads are bucketed by anchor count so that a workgroup only claims the LDS its read needs.
// Same 64-wide descending-S candidate scan as k_chain_global with the LC rules: the loop breaks on
// S[j] < max - l_i (strict) and `opcount` is bumped before that test (:27410-27415); overlapping predecessors with bonus <= 0 are
// skipped; traceback trims overlaps (:27508-27526). Scores are IEEE double in the reference's evaluation order (-ffp-contract=off).
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"

// ------------------------------------------------------------------------------------------------ L3 / L4 local chain DP
// The reference's smallorequal (:13229-13265) returns, on the score-sorted index, the position of the last score <= target, whatever
// its probe sequence; the kernel computes that count directly (vmx_sorted_count, vmx_device.h).

// One read. IN_LDS is a compile-time switch so that the working arrays are plain LDS pointers (ds_read / ds_write) in the instantiation the
// buckets run, and plain global pointers in the other one; a run-time choice between the two would make every access a flat_load.
// VAR (0 LC-exact, 1 LC-mm, 2 `_scar`, 3 the -mode asm fork's LC) is compile-time too: each instantiation carries one scoring rule.
// VAR 3 = mammap_asm.py:16540-16730: anchors sorted by read START and the candidate window advances on read starts (:16609), no opcount
// switch, no `bonus <= 0` skip, the fork's gap geometry (:16641-16669), co-linear steps pay 0.5*log2 + 0.1*log2(readgap) (:16555, :16536),
// a non-co-linear step costs skipcost + extra[gapcost] evaluated as ((S_j - skipcost) + bonus) - extra (:16686), and the traceback trims the
// EARLIER anchor's end where two chained anchors overlap on the read (:16720-16726).
template <bool IN_LDS, int VAR>
__device__ __forceinline__ void vmx_chain_local_read(const vmx_anchor* __restrict__ anchors, const int32_t* __restrict__ n_guides_total, int rd, int64_t a0, int n,
                                                     long long rmin, char* smem, const double* s_gapcost, float* s_rgc, int lds_cap, const vmx_tables& tab,
                                                     double skip_exact, double skip_mm, int maxdiff, int maxgap, int mode, double* __restrict__ S_pool,
                                                     int32_t* __restrict__ P_pool, int32_t* __restrict__ SA_pool, double* __restrict__ out_score,
                                                     vmx_anchor* __restrict__ out_chain, int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant,
                                                     int32_t* __restrict__ status, double* __restrict__ FP_pool, double* __restrict__ PP_pool) {
    const int lane = vmx_lane();
    const long long l2c_size = (long long)tab.log2cache_n - 1;
    constexpr bool in_lds = IN_LDS;
    {
        // mode R runs one variant, `_scar` (mammap_noprefercloser.py:23419-23628): anchors sorted by read START, a non-co-linear step costs
        // the fixed skipcost, remembered per anchor in fixed_penatly / pre_penatly (FP / PP, in HBM) and refunded once the chain has gone
        // on co-linearly for skipcost bases; no opcount switch
        constexpr bool scar = VAR == 2;
        constexpr bool mm = VAR == 1;
        constexpr bool asmv = VAR == 3;
        double* FP = scar ? FP_pool + a0 : nullptr; double* PP = scar ? PP_pool + a0 : nullptr;
        const double skipcost = mm ? skip_mm : skip_exact;
        const float* rgc_g = mm ? tab.large_readgap : ((mode == 3 || asmv) ? tab.readgap_r : tab.readgap_h);
        for (int x = lane; x < 100; x += 64) s_rgc[x] = rgc_g[x];          // read-gap cost table of this read's variant (100 entries, maxgap <= 99)
        const float* rgc = s_rgc;
        const vmx_anchor* A = anchors + a0;
        // LDS layout (VMX_LC_BYTES_PER_ANCHOR = 12): S f64 | S_arg i32 — only what the scan may touch at random. The anchors themselves stay in
        // HBM: the loop takes them in order from 64-anchor register blocks (one coalesced load per 64 steps, a block ahead; the current anchor
        // comes out by v_readlane), the candidates' fields live in the register window, and the rare paths (an insertion below the window, a
        // scan past 64 candidates, the traceback) read A[j] directly. P (written once per anchor, read by the traceback) stays in HBM.
        (void)rmin; (void)in_lds;
        double* S; int* SA;
        int* P = P_pool + a0;
        if constexpr (IN_LDS) { S = (double*)smem; SA = (int*)(S + lds_cap); }
        else { S = S_pool + a0; SA = SA_pool + a0; }
        __syncthreads();
#define AQ(i) (A[i].q)
#define AR(i) ((long long)A[i].r)
#define AL(i) ((int)A[i].l & 0xffff)
#define AS(i) ((int)A[i].s)
        long long prereadloc = asmv ? (long long)AQ(0) : (long long)AQ(0) + AL(0);
        int testspace_en = 1;
        if (lane == 0) { SA[0] = 0; S[0] = (double)AL(0); P[0] = VMX_NOPRE; if (scar) { FP[0] = 0.0; PP[0] = 0.0; } }
        __syncthreads();
        double g_max_scores = (double)AL(0); int g_max_index = 0;
        long long opcount = 0;
        bool need_fast = false;
        // candidate window (vmx_cwin): the testspace_en entries of S_arg, best first, the top 64 of them in registers
        vmx_cwin win; win.j = 0; win.q = AQ(0); win.ls = AL(0) | (AS(0) << 16); win.S = (double)AL(0); win.r = AR(0);
        // anchors [bb, bb + 64) in registers (lane t: anchor bb + t), the next block already on its way
        int bq = 0, bls = 0; long long br = 0, nbr = 0; int nbq = 0, nbls = 0;
        { const int x = lane < n ? lane : n - 1; const vmx_anchor a = A[x]; bq = a.q; bls = ((int)a.l & 0xffff) | ((int)a.s << 16); br = a.r; }
        { const int x = 64 + lane < n ? 64 + lane : n - 1; const vmx_anchor a = A[x]; nbq = a.q; nbls = ((int)a.l & 0xffff) | ((int)a.s << 16); nbr = a.r; }
        int pq = win.q, pls = win.ls; long long pr = win.r; double pS = win.S;      // anchor i-1 and its score
        for (int i = 1; i < n; ++i) {
            // the current anchor is the same in every lane: scalar registers, scalar branches on its strand
            if ((i & 63) == 0) {
                bq = nbq; bls = nbls; br = nbr;
                const int x = i + 64 + lane < n ? i + 64 + lane : n - 1; const vmx_anchor a = A[x]; nbq = a.q; nbls = ((int)a.l & 0xffff) | ((int)a.s << 16); nbr = a.r;
            }
            const int bl = i & 63;
            const int qi = vmx_readlane(bq, bl); const int lsi = vmx_readlane(bls, bl); const int li = lsi & 0xffff, si = lsi >> 16;
            long long ri; { union { long long d; int w[2]; } u; u.d = br; u.w[0] = vmx_readlane(u.w[0], bl); u.w[1] = vmx_readlane(u.w[1], bl); ri = u.d; }
            if (prereadloc < (asmv ? (long long)qi : (long long)qi + li)) {
                if (!scar && !asmv && opcount > 100000 && ((double)opcount / (double)prereadloc) > 1000.0) { need_fast = true; break; }   // :27380 -> *_fast
                for (int k = testspace_en; k < i; ++k) {
                    // smallorequal(...) + 1 (:13229-13265) on a sorted array = number of scores <= S[k]: the new entry goes above its equals
                    double Sk; int qk, lsk; long long rk;
                    if (k == i - 1) { Sk = pS; qk = pq; lsk = pls; rk = pr; }
                    else { Sk = S[k]; qk = AQ(k); lsk = AL(k) | (AS(k) << 16); rk = AR(k); }
                    const int W = k < 64 ? k : 64;               // k entries so far
                    const int above = __popcll(__ballot(lane < W && win.S > Sk));
                    if (above < 64 && (above < W || W == k)) {
                        vmx_cwin_insert(win, above, k, Sk, qk, lsk, rk, lane);
                        if (lane <= above) SA[k - lane] = win.j;
                    } else {                                     // lands below the window: search + shift in the full index, then reload the window
                        const int loc = vmx_sorted_count(S, SA, k, Sk, true, lane);
                        vmx_sarg_insert4(SA, loc, k, lane);
                        if (lane <= k) { const int j = SA[k - lane]; win.j = j; win.S = S[j]; win.q = AQ(j); win.ls = AL(j) | (AS(j) << 16); win.r = AR(j); }
                    }
                }
                testspace_en = i;
                prereadloc = asmv ? (long long)qi : (long long)qi + li;
            }
            const double dli = (double)li;
            double max_scores = dli; int pre_index = VMX_NOPRE;
            double fp_i = 0.0, pp_i = 0.0;                       // scar: the winner's fixed_penatly / pre_penatly
            for (int base = testspace_en - 1; base >= 0; base -= 64) {
                const int x = base - lane;
                const bool valid = x >= 0;
                int j = 0; double Sj = 0.0; double test = -1e300;
                double nfp = 0.0, npp = 0.0;                     // scar: what i inherits if this candidate wins
                if (valid) {
                    int qj, lj, sj; long long rj;
                    if (base == testspace_en - 1) { j = win.j; Sj = win.S; qj = win.q; lj = win.ls & 0xffff; sj = win.ls >> 16; rj = win.r; }   // first 64: registers
                    else { j = SA[x]; Sj = S[j]; qj = AQ(j); lj = AL(j); sj = AS(j); rj = AR(j); }
                    long long readgap, refgap, bonus;
                    if (asmv) vmx_gap_geometry_sel<true>(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
                    else vmx_gap_geometry_sel<false>(qi, ri, si, li, qj, rj, sj, lj, readgap, refgap, bonus);
                    const bool skip = !asmv && bonus <= 0;            // only an overlap can bring the bonus to zero or below (:27425)
                    if (!skip) {
                        long long gapcost = readgap - refgap; if (gapcost < 0) gapcost = -gapcost;
                        if (scar) {
                            if (si == sj && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                                test = Sj + (double)bonus - s_gapcost[gapcost] - (double)rgc[readgap];
                                const double fpj = FP[j], ppj = PP[j];
                                if (fpj < 0 && (fpj + (double)bonus) >= 0) test += ppj;                      // refund (:23557-23559)
                                if (fpj < 0 && (fpj + (double)bonus) < 0) { nfp = fpj + (double)bonus; npp = ppj; }
                            } else {
                                test = Sj + (double)bonus - skipcost;                                          // :23577-23578
                                nfp = -skipcost + (double)bonus; npp = skipcost;
                            }
                        } else {
                            // both forms computed, one selected (see k_chain.hip): no exec-mask region per kind of step
                            const bool col = si == sj && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff;
                            const double tc = Sj + (double)bonus - s_gapcost[col ? gapcost : 0] - (double)rgc[col ? readgap : 0];
                            const long long gx = col ? 0x7fffffffffffffffLL : gapcost;
                            double tn;
                            if (asmv) tn = Sj - skipcost + (double)bonus - vmx_extra_cost(tab, gx);
                            else if (!mm) {
                                const double ex = vmx_extra_cost(tab, gx);
                                const double pen = (si != sj ? (skipcost < 50.0 ? skipcost : 50.0) : skipcost) + ex;
                                tn = Sj + (double)bonus - pen;
                            } else {
                                const double pen = skipcost + tab.log2cache[gapcost < l2c_size ? gapcost : l2c_size];
                                tn = Sj + (double)bonus - pen;
                            }
                            test = col ? tc : tn;
                        }
                    }
                }
                const double incl = vmx_wave_incl_max_f64(test);                 // prefix max of the candidates' scores, in scan order
                double m_before = vmx_wave_shr1_f64_fill(incl, VMX_F64_NEG);
                m_before = m_before > max_scores ? m_before : max_scores;         // the running max the sequential loop holds at this candidate
                const bool brk = valid && (Sj < (m_before - dli));       // strict; opcount is bumped BEFORE this test (:27410-27415)
                const unsigned long long bmask = __ballot(brk);
                const unsigned long long vmask = __ballot(valid);
                const int first = bmask ? (__ffsll((unsigned long long)bmask) - 1) : 64;
                opcount += bmask ? (first + 1) : __popcll(vmask);
                if (first > 0) {
                    const double M = vmx_readlane_f64(incl, first - 1);          // best score among the candidates before the break
                    if (M > max_scores) {                                        // strict >: the first (highest-S) candidate reaching M wins
                        const unsigned long long em = __ballot(test == M) & (first >= 64 ? ~0ULL : ((1ULL << first) - 1ULL));
                        const int wl = __ffsll((unsigned long long)em) - 1;
                        pre_index = vmx_readlane(j, wl);
                        if (scar) { fp_i = vmx_readlane_f64(nfp, wl); pp_i = vmx_readlane_f64(npp, wl); }
                        max_scores = M;
                    }
                }
                if (first < 64) break;
            }
            if (lane == 0) { S[i] = max_scores; P[i] = pre_index; if (scar) { FP[i] = fp_i; PP[i] = pp_i; } }
            if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
            pS = max_scores; pq = qi; pls = li | (si << 16); pr = ri;
            if constexpr (IN_LDS && VAR != 2) vmx_wave_lds_fence(); else __syncthreads();     // (scar: FP / PP go through HBM)
        }
        __syncthreads();       // (P, written through the loop without waiting, is read back by the traceback)
        // traceback with overlap trimming :27508-27526 (serial, lane 0)
        if (lane == 0) {
            if (need_fast) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_FASTPATH_DEV; }
            else {
                vmx_anchor* O = out_chain + a0;
                int w = 0; int take = g_max_index;
                vmx_anchor pre; pre.q = AQ(take); pre.r = AR(take); pre.l = (int16_t)AL(take); pre.s = (int16_t)AS(take);
                O[w++] = pre;
                while (P[take] != VMX_NOPRE) {
                    take = P[take];
                    vmx_anchor now; now.q = AQ(take); now.r = AR(take); now.l = (int16_t)AL(take); now.s = (int16_t)AS(take);
                    if (asmv) {
                        vmx_anchor t = now;
                        if (!(pre.q >= now.q + now.l)) { t.l = (int16_t)(pre.q - now.q); if (now.s != 1) t.r = now.r + now.l - pre.q + now.q; }
                        O[w++] = t;
                        pre = now;
                        continue;
                    }
                    if (pre.q < now.q + now.l) {
                        int ov = now.q + now.l - pre.q;
                        vmx_anchor t = pre; t.q = pre.q + ov; t.l = (int16_t)(pre.l - ov); if (pre.s == 1) t.r = pre.r + ov;
                        O[w - 1] = t;
                    }
                    O[w++] = now;
                    pre = now;
                }
                out_len[rd] = w; out_score[rd] = g_max_scores; status[rd] = 0;
            }
            out_variant[rd] = asmv ? 3 : (scar ? 2 : (mm ? 1 : 0));
        }
        __syncthreads();
#undef AQ
#undef AR
#undef AL
#undef AS
    }
}

__global__ void __launch_bounds__(64) k_chain_local(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ la_off,
                                                    const int32_t* __restrict__ la_cnt, const int32_t* __restrict__ n_guides_total,
                                                    const int32_t* __restrict__ rlist, int nlist, int lds_cap, vmx_tables tab,
                                                    const double* __restrict__ gapcost_list, double skip_exact, double skip_mm, int maxdiff,
                                                    int maxgap, int mode, double* __restrict__ S_pool, int32_t* __restrict__ P_pool,
                                                    int32_t* __restrict__ SA_pool, double* __restrict__ out_score,
                                                    vmx_anchor* __restrict__ out_chain, int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant,
                                                    int32_t* __restrict__ status, double* __restrict__ FP_pool, double* __restrict__ PP_pool) {
    VMX_SETPRIO(3);
    VMX_DYN_SHARED(char, smem);
    __shared__ double s_gapcost[64];
    __shared__ float s_rgc[128];
    const int lane = vmx_lane();
    for (int x = lane; x <= maxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    for (int li_ = blockIdx.x; li_ < nlist; li_ += gridDim.x) {
        const int rd = rlist[li_];
        const int64_t a0 = la_off[rd];
        const int n = la_cnt[rd];
        if (n <= 0) { if (lane == 0) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_RAISED_DEV; } continue; }   // np.array([]) indexing raises
        const vmx_anchor* A = anchors + a0;
        const long long rmin = 0;
        const int var = mode == 4 ? 3 : (mode == 3 ? 2 : (n_guides_total[rd] > 1 ? 1 : 0));
#define VMX_LC_CALL(L, V) vmx_chain_local_read<L, V>(anchors, n_guides_total, rd, a0, n, rmin, smem, s_gapcost, s_rgc, lds_cap, tab, skip_exact, skip_mm, maxdiff, maxgap, \
                                                      mode, S_pool, P_pool, SA_pool, out_score, out_chain, out_len, out_variant, status, FP_pool, PP_pool)
        if (n <= lds_cap) { if (var == 0) VMX_LC_CALL(true, 0); else if (var == 1) VMX_LC_CALL(true, 1); else if (var == 2) VMX_LC_CALL(true, 2); else VMX_LC_CALL(true, 3); }
        else { if (var == 0) VMX_LC_CALL(false, 0); else if (var == 1) VMX_LC_CALL(false, 1); else if (var == 2) VMX_LC_CALL(false, 2); else VMX_LC_CALL(false, 3); }
#undef VMX_LC_CALL
    }
}
