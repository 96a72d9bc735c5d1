// k_chain_fast.hip — the heuristic "_fast" chain variants, taken by repeat-dense reads:
//   GC-fast      get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all     /root/reference/src/vacmap/mammap_clrnano.py:25033-25339
//   LC-fast      ..._merged_fine_list_fast                                                     :26938-27303
//   LC-mm-fast   ..._merged_fine_list_mismatch_fast                                            :27891-28249
// (+ insertpoint_score_distance :17200, closest2targetdistance :17228).
//
// The reference keeps the finished anchors in an index sorted by (int(S), diagonal key); S_i_count[c] = number of entries with
// int(S) = c, so the bucket of score level c is the slice [en - count, en) below the buckets of the higher levels. For anchor i it
// walks the levels from max_score_i downwards while  c > max - (l_i + 1);  a bucket of more than 5 entries is represented by the
// entry whose diagonal key is closest to i's, a smaller one is evaluated entry by entry from its end.
// One wavefront per read. A step looks at 64 consecutive score levels: lane = level. The bucket slices of the 64 levels come from a
// prefix sum of their counts (one coalesced load), every lane evaluates its own bucket, and the walk's sequential "running max"
// test becomes the same exclusive prefix-max + first-break-lane pattern as in k_chain_global (the condition is monotone: once a
// level fails, all lower ones fail). Insertion positions replay the reference's bisection on (a, b) = (#entries below, #entries not
// above the new key), found by a wave-cooperative 64-ary search, so equal keys land where the reference puts them.
// Everything lives in HBM (these reads have 10^4..10^5 anchors); scores are IEEE double in the reference's evaluation order.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"
#include "vmx_link.h"

struct vmx_fast_io {
    const vmx_anchor* A; int n;
    double* S; int32_t* P; int32_t* SA;            // outputs / working arrays (n each)
    int32_t* Si; int64_t* T;                        // int(S) and diagonal key per anchor
    int32_t* CNT; int cnt_n;                        // S_i_count (cnt_n = last read position + 50)
    uint8_t* COV;                                   // anchors sharing the read position, capped at 20 (GC-fast only)
    double* FP; double* PP;                         // fixed_penatly / pre_penatly (mode R's GC-fast only)
};

// (Si, T) of entry x of the index vs (ts, td): -1 below, 0 equal, +1 above
__device__ __forceinline__ int vmx_fast_cmp(const vmx_fast_io& io, int x, int ts, long long td) {
    const int e = io.SA[x];
    const int s = io.Si[e];
    if (s != ts) return s < ts ? -1 : 1;
    const long long d = io.T[e];
    return d < td ? -1 : (d > td ? 1 : 0);
}

// number of index entries (first k) strictly below (le = false) / not above (le = true) the key (ts, td); wave-cooperative
__device__ __forceinline__ int vmx_fast_count(const vmx_fast_io& io, int k, int ts, long long td, bool le, int lane) {
    int lo = 0, hi = k;
    while (hi - lo > 64) {
        const int stride = (hi - lo + 63) >> 6;
        const int x = lo + (lane + 1) * stride - 1;
        bool in = false;
        if (x < hi) { const int c = vmx_fast_cmp(io, x, ts, td); in = le ? c <= 0 : c < 0; }
        const int c = __popcll(__ballot(in));
        const int nlo = lo + c * stride;
        int nhi = nlo + stride - 1; if (nhi > hi) nhi = hi;
        lo = nlo; hi = nhi;
    }
    const int x = lo + lane;
    bool in = false;
    if (x < hi) { const int c = vmx_fast_cmp(io, x, ts, td); in = le ? c <= 0 : c < 0; }
    return lo + __popcll(__ballot(in));
}

// insertpoint_score_distance (:17200-17226) for entry k (key (Si[k], T[k])) into the first k entries: the literal bisection only
// distinguishes below / above / equal, so it is replayed on the two counts
__device__ __forceinline__ int vmx_fast_insertpoint(const vmx_fast_io& io, int k, int lane) {
    const int ts = io.Si[k]; const long long td = io.T[k];
    const int a = vmx_fast_count(io, k, ts, td, false, lane);
    int b = a;
    if (a < k && vmx_fast_cmp(io, a, ts, td) == 0) b = vmx_fast_count(io, k, ts, td, true, lane);
    int i = 0, j = k;
    while (i < j) {
        const int mid = (i + j) >> 1;
        if (mid < a) i = mid + 1;
        else if (mid >= b) j = mid;
        else return mid + 1;
    }
    return j;
}

// SA[loc+1 : k+1] = SA[loc : k]; SA[loc] = k, by one wave, 1024 entries per round (16 per lane held in registers between the loads
// and the stores; a single wavefront's memory operations are issued in order)
__device__ __forceinline__ void vmx_fast_insert(int32_t* SA, int loc, int k, int lane) {
    for (int hi = k; hi > loc; hi -= 1024) {
        int v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int x = hi - u * 64 - lane; v[u] = x > loc ? SA[x - 1] : 0; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int x = hi - u * 64 - lane; if (x > loc) SA[x] = v[u]; }
        __syncthreads();
    }
    if (lane == 0) SA[loc] = k;
    __syncthreads();
}

// closest2targetdistance :17228-17252 — literal (one lane, its own bucket)
__device__ __forceinline__ int vmx_fast_closest(const vmx_fast_io& io, long long target, int st_loc, int en_loc) {
    int i = st_loc, j = en_loc;
    if (io.T[io.SA[i]] >= target) return i;
    if (io.T[io.SA[j - 1]] <= target) return j - 1;
    while (i < j) {
        const int mid = (i + j) >> 1;
        const long long now = io.T[io.SA[mid]];
        if (now < target) i = mid + 1;
        else if (now > target) j = mid;
        else return mid;
    }
    if ((target - io.T[io.SA[j - 1]]) < (io.T[io.SA[j]] - target)) return j - 1;
    return j;
}

struct vmx_fast_cost {
    const double* gapcost; const float* rgc; vmx_tables tab; double skipcost; int maxdiff, maxgap; long long extra_size, l2c_size;
};

// one candidate j for anchor i (:25191-25237 / :27098-27166). VARIANT 0 GC-fast, 1 LC-fast, 2 LC-mm-fast, 3 GC-fast of mode R
// (mammap_noprefercloser.py:23059-23417: fixed skipcost with refund; fpj / ppj = fixed_penatly[j] / pre_penatly[j], *nfp / *npp = what
// i inherits if j wins). returns false when LC's `bonus <= 0` fires (the candidate is skipped)
template <int VARIANT>
__device__ __forceinline__ bool vmx_fast_eval(const vmx_anchor& ai, const vmx_anchor& aj, double Sj, const vmx_fast_cost& C, double* test,
                                              double fpj = 0.0, double ppj = 0.0, double* nfp = nullptr, double* npp = nullptr) {
    const int qi = ai.q, li = ai.l, si = ai.s, qj = aj.q, lj = aj.l, sj = aj.s; const long long ri = ai.r, rj = aj.r;
    long long readgap = (long long)qi - qj - lj, refgap, bonus;
    if (VARIANT == 4) {                                   // -mode asm: mammap_asm.py:20866-20895 (non_overlap_size form, no +-1)
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
    } else if (readgap < 0) {
        bonus = (long long)qi + li - qj - lj;
        if ((VARIANT == 1 || VARIANT == 2) && bonus <= 0) return false;
        readgap = 0;
        const long long overlap = (long long)qj + lj - qi;
        if (si == sj) { if (si == 1) refgap = ri + overlap - (rj + lj); else refgap = rj - (ri + bonus); }
        else { if (sj == -1) refgap = ri + overlap - rj + 1; else refgap = ri + bonus - 1 - (rj + lj); }
    } else {
        bonus = li;
        if (si == sj) { if (si == 1) refgap = ri - rj - lj; else refgap = rj - ri - li; }
        else { if (sj == -1) refgap = ri - rj + 1; else refgap = ri + li - 1 - rj - lj; }
    }
    long long gapcost = readgap - refgap; if (gapcost < 0) gapcost = -gapcost;
    if (VARIANT == 3) {
        if (si == sj && refgap >= 0 && readgap <= C.maxgap && gapcost <= C.maxdiff) {
            double t = Sj + (double)bonus - C.gapcost[gapcost];
            if (fpj < 0 && (fpj + (double)bonus) >= 0) t += ppj;
            if (fpj < 0 && (fpj + (double)bonus) < 0) { *nfp = fpj + (double)bonus; *npp = ppj; } else { *nfp = 0.0; *npp = 0.0; }
            *test = t;
        } else {
            *test = Sj + (double)bonus - C.skipcost;
            *nfp = -C.skipcost + (double)bonus; *npp = C.skipcost;
        }
        return true;
    }
    if (si == sj && refgap >= 0 && readgap <= C.maxgap && gapcost <= C.maxdiff) {
        if (VARIANT == 0 || VARIANT == 4) *test = Sj + (double)bonus - C.gapcost[gapcost];
        else *test = Sj + (double)bonus - C.gapcost[gapcost] - (double)C.rgc[readgap];
    } else if (VARIANT == 0 || VARIANT == 4) {
        *test = Sj - C.skipcost + (double)bonus - vmx_extra_cost(C.tab, gapcost);
    } else if (VARIANT == 1) {
        const double ex = vmx_extra_cost(C.tab, gapcost);
        double pen;
        if (si != sj) pen = (C.skipcost < 50.0 ? C.skipcost : 50.0) + ex;
        else pen = C.skipcost + ex;
        *test = Sj + (double)bonus - pen;
    } else {
        const double pen = C.skipcost + C.tab.log2cache[gapcost < C.l2c_size ? gapcost : C.l2c_size];
        *test = Sj + (double)bonus - pen;
    }
    return true;
}

// the DP. returns g_max_index (>= 0), -4 (the reference would not terminate: LC-fast's `continue` in the large-bucket branch,
// :27103-27104) or -5 (an integer score outside S_i_count: IndexError in the reference). *gmax_score gets the best score.
// lk (VARIANT 4 only): the LINKED form, mammap_asm.py:21871-22158 — S / P of the first lk->n_pre rows are the carried state (already in io.S /
// io.P), S_i = int(pre_S), the bucket index starts with row 0 alone and max_score_i = S_i[0] (:21893-21901), the loop starts behind the carried rows
struct vmx_fast_link { int n_pre; double g_max_scores; int g_max_index; long long prereadloc; };
template <int VARIANT>
__device__ int vmx_fast_dp(const vmx_fast_io& io, vmx_fast_cost C, const double* gapcost_list, double oskipcost, int omaxdiff, double* gmax_score, const vmx_fast_link* lk = nullptr) {
    const int lane = vmx_lane();
    const int n = io.n;
    const vmx_anchor* A = io.A;
    const long long readlength = (long long)A[n - 1].q + 1000;
    for (int x = lane; x < io.cnt_n; x += 64) io.CNT[x] = 0;
    for (int i = lane; i < n; i += 64) {
        const vmx_anchor a = A[i];
        io.T[i] = a.s == 1 ? (long long)a.r - a.q + readlength : -((long long)a.r + a.q + readlength);
        if (VARIANT == 0) {
            int c = 1;
            for (int x = i - 1; x >= 0 && A[x].q == a.q && c < 20; --x) ++c;
            for (int x = i + 1; x < n && A[x].q == a.q && c < 20; ++x) ++c;
            io.COV[i] = (uint8_t)c;
        }
    }
    __syncthreads();
    const vmx_anchor a0 = A[0];
    constexpr bool GC = VARIANT == 0 || VARIANT == 3 || VARIANT == 4;     // 4: GC-fast of the -mode asm fork (mammap_asm.py:20738-21037), no coverage terms
    long long prereadloc = GC ? (long long)a0.q : (long long)a0.q + a0.l;
    C.gapcost = gapcost_list; C.skipcost = oskipcost; C.maxdiff = omaxdiff;
    int testspace_en_i = 1;
    if ((int)a0.l >= io.cnt_n) return -5;
    const bool linked = VARIANT == 4 && lk && lk->n_pre > 0;
    if (lane == 0) { io.SA[0] = 0; if (!linked) { io.S[0] = (double)a0.l; io.P[0] = VMX_NOPRE; } io.Si[0] = a0.l; io.CNT[a0.l] = 1; if (VARIANT == 3) { io.FP[0] = 0.0; io.PP[0] = 0.0; } }
    __syncthreads();
    double g_max_scores = (double)a0.l; int g_max_index = 0;
    int max_score_i = 0;
    int err = 0;
    int pre_size = 1;
    if (VARIANT == 4 && lk && lk->n_pre > 0) {
        for (int i = lane; i < lk->n_pre; i += 64) io.Si[i] = (int32_t)(long long)io.S[i];
        __syncthreads();
        const int s0 = io.Si[0];
        if (s0 < 0 || s0 >= io.cnt_n) return -5;
        if (lane == 0) { io.CNT[a0.l] = 0; io.CNT[s0] = 1; }
        __syncthreads();
        pre_size = lk->n_pre; g_max_scores = lk->g_max_scores; g_max_index = lk->g_max_index; prereadloc = lk->prereadloc; max_score_i = s0;
    }
    for (int i = pre_size; i < n && !err; ++i) {
        const vmx_anchor ai = A[i];
        const long long pos_i = GC ? (long long)ai.q : (long long)ai.q + ai.l;
        if (prereadloc < pos_i) {
            for (int k = testspace_en_i; k < i; ++k) {                         // :25132-25146
                const int sk = io.Si[k];
                if (sk < 0 || sk >= io.cnt_n) { err = -5; break; }
                if (lane == 0) io.CNT[sk] += 1;
                if (sk > max_score_i) max_score_i = sk;
                __syncthreads();
                const int loc = vmx_fast_insertpoint(io, k, lane);
                vmx_fast_insert(io.SA, loc, k, lane);
            }
            if (err) break;
            testspace_en_i = i;
            if (VARIANT == 0) {
                const int cv = io.COV[i];
                C.skipcost = oskipcost + (double)cv;                            // :25151
                int md = omaxdiff - cv; C.maxdiff = md < 10 ? 10 : md;          // :25152
            }
            prereadloc = pos_i;
        }
        double max_scores = (double)ai.l; int pre_index = VMX_NOPRE;
        double fp_i = 0.0, pp_i = 0.0;
        const double f = (double)(ai.l + 1);
        const long long ti = io.T[i];
        int en_top = testspace_en_i;                      // end of the bucket of the highest level of the current window
        for (int top = max_score_i; top >= 0; top -= 64) {
            const int c = top - lane;                     // this lane's score level
            const int cnt = c >= 0 ? io.CNT[c] : 0;
            const int incl_cnt = vmx_wave_incl_scan_i32(cnt);
            const int en_loc = en_top - (incl_cnt - cnt), st_loc = en_loc - cnt;
            double test = -1e300; int jbest = VMX_NOPRE; bool hang = false;
            double nfp = 0.0, npp = 0.0;
            if (cnt > 0 && st_loc >= 0) {
                if (cnt > 5) {
                    const int j = io.SA[vmx_fast_closest(io, ti, st_loc, en_loc)];
                    double t, a = 0.0, b = 0.0;
                    const double fpj = VARIANT == 3 ? io.FP[j] : 0.0, ppj = VARIANT == 3 ? io.PP[j] : 0.0;
                    if (vmx_fast_eval<VARIANT>(ai, A[j], io.S[j], C, &t, fpj, ppj, &a, &b)) { test = t; jbest = j; nfp = a; npp = b; }
                    else hang = true;                     // reference: `continue` with unchanged loop state
                } else {
                    for (int x = en_loc - 1; x >= st_loc; --x) {
                        const int j = io.SA[x];
                        double t, a = 0.0, b = 0.0;
                        const double fpj = VARIANT == 3 ? io.FP[j] : 0.0, ppj = VARIANT == 3 ? io.PP[j] : 0.0;
                        if (vmx_fast_eval<VARIANT>(ai, A[j], io.S[j], C, &t, fpj, ppj, &a, &b) && t > test) { test = t; jbest = j; nfp = a; npp = b; }
                    }
                }
            }
            // the sequential walk visits level c iff c > (max before it) - f; levels are visited top down
            const double incl = vmx_wave_incl_max_f64(test);
            double m_before = vmx_wave_shr1_f64_fill(incl, VMX_F64_NEG);
            m_before = m_before > max_scores ? m_before : max_scores;
            const bool brk = c < 0 || !((double)c > (m_before - f));
            const unsigned long long bmask = __ballot(brk);
            const int first = bmask ? (__ffsll((unsigned long long)bmask) - 1) : 64;
            const unsigned long long vis = first >= 64 ? ~0ULL : ((1ULL << first) - 1ULL);
            if (__ballot(hang) & vis) { err = -4; break; }
            if (first > 0) {
                const double M = vmx_readlane_f64(incl, first - 1);
                if (M > max_scores) {
                    const unsigned long long em = __ballot(test == M) & vis;
                    const int wl = __ffsll((unsigned long long)em) - 1;
                    pre_index = vmx_readlane(jbest, wl);
                    if (VARIANT == 3) { fp_i = vmx_readlane_f64(nfp, wl); pp_i = vmx_readlane_f64(npp, wl); }
                    max_scores = M;
                }
            }
            if (first < 64) break;
            en_top -= vmx_readlane(incl_cnt, 63);
        }
        if (err) break;
        if (lane == 0) { io.S[i] = max_scores; io.Si[i] = (int32_t)(long long)max_scores; io.P[i] = pre_index; if (VARIANT == 3) { io.FP[i] = fp_i; io.PP[i] = pp_i; } }   // truncation toward zero (:25314)
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
        __syncthreads();
    }
    if (err) return err;
    if (GC) {                                                                    // :25324-25336
        for (int k = testspace_en_i; k < n; ++k) {
            const int sk = io.Si[k];
            if (sk < 0 || sk >= io.cnt_n) return -5;
            if (lane == 0) io.CNT[sk] += 1;
            __syncthreads();
            const int loc = vmx_fast_insertpoint(io, k, lane);
            vmx_fast_insert(io.SA, loc, k, lane);
        }
    }
    *gmax_score = g_max_scores;
    return g_max_index;
}

// G3: reads whose GC-exact launch left gmax = -1 (fast_enable, :23570, or the opcount bail-out, :24914). One workgroup (one wave) per read.
__global__ void __launch_bounds__(64) k_chain_global_fast(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ aoff, int n_reads,
                                                          const int64_t* __restrict__ roff, vmx_tables tab, const double* __restrict__ gapcost_list,
                                                          double oskipcost, int omaxdiff, int maxgap, double* __restrict__ S_out,
                                                          int32_t* __restrict__ P_out, int32_t* __restrict__ SA_out, uint8_t* __restrict__ cov_pool,
                                                          int32_t* __restrict__ si_pool, int64_t* __restrict__ t_pool, int32_t* __restrict__ cnt_pool,
                                                          int64_t* __restrict__ gmax_out, int32_t* __restrict__ ran, int rmode,
                                                          double* __restrict__ FP_pool, double* __restrict__ PP_pool) {
    const int rd = (int)blockIdx.x;
    if (rd >= n_reads) return;
    const int64_t a0 = aoff[rd];
    const int n = (int)(aoff[rd + 1] - a0);
    if (n <= 2 || gmax_out[rd] != -1) return;
    if (ran && vmx_lane() == 0) ran[rd] = 1;
    vmx_fast_io io;
    io.A = anchors + a0; io.n = n; io.S = S_out + a0; io.P = P_out + a0; io.SA = SA_out + a0; io.Si = si_pool + a0; io.T = t_pool + a0;
    io.CNT = cnt_pool + roff[rd] + 50 * (int64_t)rd; io.cnt_n = io.A[n - 1].q + 50;
    io.COV = cov_pool + a0; io.FP = rmode == 1 ? FP_pool + a0 : nullptr; io.PP = rmode == 1 ? PP_pool + a0 : nullptr;
    vmx_fast_cost C; C.gapcost = gapcost_list; C.rgc = nullptr; C.tab = tab; C.skipcost = oskipcost; C.maxdiff = omaxdiff; C.maxgap = maxgap;
    C.extra_size = (long long)tab.extra_n - 1; C.l2c_size = (long long)tab.log2cache_n - 1;
    double gs = 0.0;
    const int g = rmode == 1 ? vmx_fast_dp<3>(io, C, gapcost_list, oskipcost, omaxdiff, &gs) : (rmode == 2 ? vmx_fast_dp<4>(io, C, gapcost_list, oskipcost, omaxdiff, &gs) : vmx_fast_dp<0>(io, C, gapcost_list, oskipcost, omaxdiff, &gs));
    if (vmx_lane() == 0) gmax_out[rd] = g >= 0 ? g : -2;       // -2: the reference raises on this read (treated as unmapped)
}

// L5: reads whose LC launch ended with VM_READ_FASTPATH_DEV (:27380 / :28333). anchors sorted by read end.
__global__ void __launch_bounds__(64) k_chain_local_fast(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ la_off,
                                                         const int32_t* __restrict__ la_cnt, const int32_t* __restrict__ n_guides_total, int n_reads,
                                                         const int64_t* __restrict__ roff, vmx_tables tab, const double* __restrict__ gapcost_list,
                                                         double skip_exact, double skip_mm, int maxdiff, int maxgap, int mode,
                                                         double* __restrict__ S_pool, int32_t* __restrict__ P_pool, int32_t* __restrict__ SA_pool,
                                                         int32_t* __restrict__ si_pool, int64_t* __restrict__ t_pool, int32_t* __restrict__ cnt_pool,
                                                         double* __restrict__ out_score, vmx_anchor* __restrict__ out_chain,
                                                         int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant, int32_t* __restrict__ status) {
    const int rd = (int)blockIdx.x;
    if (rd >= n_reads) return;
    if (status[rd] != VM_READ_FASTPATH_DEV) return;
    const int64_t a0 = la_off[rd];
    const int n = la_cnt[rd];
    const bool mm = n_guides_total[rd] > 1;
    vmx_fast_io io;
    io.A = anchors + a0; io.n = n; io.S = S_pool + a0; io.P = P_pool + a0; io.SA = SA_pool + a0; io.Si = si_pool + a0; io.T = t_pool + a0;
    io.CNT = cnt_pool + roff[rd] + 50 * (int64_t)rd; io.cnt_n = io.A[n - 1].q + 50;
    io.COV = nullptr; io.FP = nullptr; io.PP = nullptr;
    vmx_fast_cost C; C.gapcost = gapcost_list; C.tab = tab; C.maxdiff = maxdiff; C.maxgap = maxgap;
    C.rgc = mm ? tab.large_readgap : (mode == 3 ? tab.readgap_r : tab.readgap_h);
    C.extra_size = (long long)tab.extra_n - 1; C.l2c_size = (long long)tab.log2cache_n - 1;
    const double skipcost = mm ? skip_mm : skip_exact;
    C.skipcost = skipcost;
    double gs = 0.0;
    const int g = mm ? vmx_fast_dp<2>(io, C, gapcost_list, skipcost, maxdiff, &gs) : vmx_fast_dp<1>(io, C, gapcost_list, skipcost, maxdiff, &gs);
    if (vmx_lane() == 0) {
        if (g < 0) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_RAISED_DEV; }
        else {
            // traceback with overlap trimming :27283-27301
            const vmx_anchor* A = io.A; const int32_t* P = io.P;
            vmx_anchor* O = out_chain + a0;
            int w = 0; int take = g;
            vmx_anchor pre = A[take];
            O[w++] = pre;
            while (P[take] != VMX_NOPRE) {
                take = P[take];
                const vmx_anchor now = A[take];
                if (pre.q < now.q + now.l) {
                    const int ov = now.q + now.l - pre.q;
                    vmx_anchor t = pre; t.q = pre.q + ov; t.l = (int16_t)(pre.l - ov); if (pre.s == 1) t.r = pre.r + ov;
                    O[w - 1] = t;
                }
                O[w++] = now;
                pre = now;
            }
            out_len[rd] = w; out_score[rd] = gs; status[rd] = 0;
        }
        out_variant[rd] = mm ? 1 : 0;
    }
}


// -mode asm, contigs of 500 kb and more: the batches whose linked GC-exact bailed out (k_chain_linked left gmax = -1, mammap_asm.py:23246-23247)
// are chained again with the fork's linked GC-fast (:21871-22158). One wavefront per job; jobs that did not bail out return at once.
__global__ void __launch_bounds__(64) k_chain_linked_fast(vmx_link_job* __restrict__ jobs, int n_jobs, vmx_tables tab, const double* __restrict__ gapcost_list,
                                                         double skipcost, int maxdiff, int maxgap) {
    const int lane = vmx_lane();
    for (int jb = (int)blockIdx.x; jb < n_jobs; jb += (int)gridDim.x) {
        vmx_link_job& J = jobs[jb];
        vmx_link_state& ST = *J.state;
        if (!J.ran || J.gmax != -1 || ST.status != 0 || !J.Si) continue;
        const int n_pre = ST.n_pre, base = J.cap_pre - n_pre, n = n_pre + J.n_new;
        vmx_fast_io io;
        io.A = J.rows + base; io.n = n; io.S = J.S + base; io.P = J.P + base; io.SA = J.SA; io.Si = J.Si; io.T = J.T; io.CNT = J.CNT; io.cnt_n = io.A[n - 1].q + 50;
        io.COV = nullptr; io.FP = nullptr; io.PP = nullptr;
        for (int i = lane; i < n_pre; i += 64) { io.S[i] = ST.pre_S[i]; io.P[i] = ST.pre_P[i]; }
        __syncthreads();
        vmx_fast_cost C; C.gapcost = gapcost_list; C.rgc = nullptr; C.tab = tab; C.skipcost = skipcost; C.maxdiff = maxdiff; C.maxgap = maxgap;
        C.extra_size = (long long)tab.extra_n - 1; C.l2c_size = (long long)tab.log2cache_n - 1;
        vmx_fast_link lk; lk.n_pre = n_pre; lk.g_max_scores = ST.g_max_scores; lk.g_max_index = ST.g_max_index; lk.prereadloc = ST.prereadloc;
        double gs = 0.0;
        const int g = vmx_fast_dp<4>(io, C, gapcost_list, skipcost, maxdiff, &gs, &lk);
        if (lane == 0) { J.gmax = g >= 0 ? g : -2; J.n = n; J.hot = n; J.n_cold = 0; J.cold_max = -1e300; }      // -2: the reference raises (an integer score outside S_i_count)
        __syncthreads();
    }
}
