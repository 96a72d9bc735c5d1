// k_chain_rows.hip — the chain DPs with FOUR reads per wavefront, one per 16-lane DPP row (SURVEY §8(a) rows G2, L3, L4, L5 `_scar`).
//
//   k_chain_global_rows   G2 get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all
//                         (/root/reference/src/vacmap/mammap_clrnano.py:24828-25031; mode R: mammap_noprefercloser.py:22839-23057)
//   k_chain_local_rows    L3 LC-exact (:27305-27528), L4 LC-mm (:28250-28476), L5 `_scar` (mammap_noprefercloser.py:23419-23628)
//
// Why rows. The reference scans the predecessors of an anchor in descending score and stops at the first one that cannot win any more
// (:24940 `if S[j] > max_scores - l_i ... else break`): on ONT and HiFi reads 2-3 candidates are looked at per anchor, 16 or more for one
// anchor in a thousand (tools/ubench/chain_stats.py). k_chain_global / k_chain_local (k_chain.hip, k_chain_local.hip) give a read a whole
// wavefront and evaluate 64 candidates per step — 60 lanes of every instruction are wasted, and a read owns a wave slot and 12 B of LDS per
// anchor. Here a ROW of 16 lanes owns a read: its 16 best predecessors sit in registers (the window, lane t = t-th best), one step of the
// loop finishes one anchor of each of the wave's four reads, nothing is kept in LDS, and the common path reads nothing it has written
// (S / P / S_arg go to HBM as stores only), so no wait for a store sits on the per-anchor chain.
//
// What a step does for anchor i of a row (same arithmetic, same order as the reference, IEEE double, -ffp-contract=off):
//   1 the anchor's fields come out of a 16-anchor register block by ds_bpermute, a step ahead of their use;
//   2 if its read position (LC: read end) passes `prereadloc` the candidate space advances: te = i (:24905-24932). The reference inserts the
//     anchors te .. i-1 into the sorted index at that moment; here every anchor is inserted as soon as its score is known — the index
//     goes through the same sequence of states, since anchor k is inserted into the entries 0 .. k-1 either way — and an entry with
//     j >= te is simply not VISIBLE to the scan yet (it is passed over: no break, no opcount);
//   3 every lane evaluates its window entry against the anchor; an inclusive prefix maximum along the row (row_shr:1/2/4/8) gives the
//     running maximum the sequential loop would hold at each candidate, a ballot cut to the row's 16 bits the first candidate that breaks
//     the loop, `opcount` the number of visible candidates before it (T4), and the LAST strict increase before the break the winner (T2);
//   4 the score, its predecessor and the row's running best are stored / updated; the new entry goes into the window at the number of
//     larger scores (GC: among equal scores where the reference's bisection :19369-19387 puts it, replayed on the two counts), the top
//     of S_arg is rewritten by the lanes at or above it.
// Rare paths, taken by a row on its own while the other rows of the wave wait: a scan that passes all 16 window entries goes on through
// S_arg / S / the anchors in HBM, 16 candidates per step; an entry that lands below the window (or among equal scores that reach below
// it) is placed by a 16-ary search and a shift of S_arg in HBM. Both first wait for the row's own stores.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_rows.h"
#include "vmx_local.h"

#define VMX_RW_NONE 0x7fffffff          /* window lane without an entry */

// window entry = one candidate predecessor: anchor index, score, read position, length | strand bit << 16, reference position
// (mode R / `_scar`: + fixed_penatly / pre_penatly of that anchor)
struct vmx_rwin { int j, q, ls; double S; long long r; double fp, pp; };
// the anchor being computed, the same in the 16 lanes of its row
struct vmx_rcur { int q, l, sneg; long long r; double dl; int te; double skipcost; int maxdiff; };

// strand of an encoded ls word: bit 16 set = -1
#define VMX_RW_LS(l, s) (((int)(l) & 0xffff) | ((s) < 0 ? 0x10000 : 0))

// gap geometry of :24953-24984 / :27418-27456 (vmx_gap_geometry_sel<false>, vmx_kernels.h) on the row kernels' strand bits
__device__ __forceinline__ void vmx_rw_geometry(const vmx_rcur& c, int qj, int lj, int snegj, long long rj, int& readgap, long long& refgap, int& bonus) {
    const int rg = c.q - qj - lj;
    const int m = rg < 0 ? rg : 0;
    readgap = rg - m;
    bonus = c.l + m;
    const long long d = c.r - rj;
    const bool same = c.sneg == snegj, neg = c.sneg != 0;
    // (selects: the strand is the same in the 16 lanes of a row, but the four rows of the wave differ, and a branch would be an exec-mask region)
    const int cpos = (same ? -lj : 1) - m, cneg = same ? -c.l - m : c.l - lj + m - 1;
    const long long t = (neg && same) ? -d : d;
    refgap = t + (long long)(neg ? cneg : cpos);
}

// KIND: 0 GC modes H / L / S, 1 GC mode R, 2 LC-exact, 3 LC-mm, 4 `_scar`
template <int KIND> struct vmx_rw_traits {
    static constexpr bool gc = KIND <= 1;
    static constexpr bool pen = KIND == 1 || KIND == 4;           // fixed_penatly / pre_penatly bookkeeping
};

struct vmx_rw_costs {
    const double* s_gapcost; const float* s_rgc; double skip; int maxgap; long long l2c_size; const double* log2cache;
};

// extra[min(g, extra_n - 1)] (vmx_extra_cost, vmx_kernels.h) for a gap already cut to 32 bits, without a branch on the common forms: the linear
// prefix of :15371-15376 in closed form, the table's last value 36 at and beyond its end; only the logarithmic stretch in between (gaps of 25 kb to
// 163 kb: a step across an SV) is loaded, behind a branch no lane takes as a rule.
__device__ __forceinline__ double vmx_rw_extra(const vmx_tables& tab, int g) {
    const double dg = (double)g;
    const double a = dg * 0.01;
    double ex = (double)(float)((a < 10.0 ? a : 10.0) + dg * 0.001);
    ex = g >= tab.extra_n - 1 ? 36.0 : ex;
    if (g >= tab.extra_arith_n && g < tab.extra_n - 1) ex = (double)tab.extra[g];
    return ex;
}

// score of the step candidate -> anchor (test_scores), and for the penalty variants what the anchor inherits if that candidate wins.
// Evaluated by every lane, entry or not (the caller drops what does not count): selects, no exec-mask regions.
template <int KIND>
__device__ __forceinline__ double vmx_rw_test(const vmx_rcur& c, const vmx_rw_costs& K, const vmx_tables& tab, const vmx_rwin& w, double& nfp, double& npp) {
    const int lj = w.ls & 0xffff, snegj = (w.ls >> 16) & 1;
    int readgap, bonus; long long refgap;
    vmx_rw_geometry(c, w.q, lj, snegj, w.r, readgap, refgap, bonus);
    long long gap64 = (long long)readgap - refgap; if (gap64 < 0) gap64 = -gap64;
    const int gapcost = gap64 > 0x40000000LL ? 0x40000000 : (int)gap64;      // every use saturates far below 2^30 (maxdiff <= 62, the tables' ends)
    const bool col = c.sneg == snegj && refgap >= 0 && readgap <= K.maxgap && gapcost <= c.maxdiff;
    const double Sj = w.S, db = (double)bonus;
    nfp = 0.0; npp = 0.0;
    if constexpr (KIND == 0) {
        // both forms are computed and one is selected (k_chain.hip)
        const double tc = Sj + db - K.s_gapcost[col ? gapcost : 0];
        const double tn = Sj - c.skipcost + db - vmx_rw_extra(tab, col ? 0x40000000 : gapcost);
        return col ? tc : tn;
    } else if constexpr (KIND == 1) {
        double tc = Sj + db - K.s_gapcost[col ? gapcost : 0];
        const double f2 = w.fp + db;
        const bool owes = w.fp < 0;
        tc = (owes && f2 >= 0) ? tc + w.pp : tc;                                // refund (mammap_noprefercloser.py:22983-22985)
        const double tn = Sj + db - c.skipcost;
        nfp = col ? ((owes && f2 < 0) ? f2 : 0.0) : -c.skipcost + db;
        npp = col ? ((owes && f2 < 0) ? w.pp : 0.0) : c.skipcost;
        return col ? tc : tn;
    } else if constexpr (KIND == 2 || KIND == 3) {
        const double tc = Sj + db - K.s_gapcost[col ? gapcost : 0] - (double)K.s_rgc[col ? readgap : 0];
        double tn;
        if constexpr (KIND == 2) {
            const double ex = vmx_rw_extra(tab, col ? 0x40000000 : gapcost);
            const double pen = (c.sneg != snegj ? (K.skip < 50.0 ? K.skip : 50.0) : K.skip) + ex;
            tn = Sj + db - pen;
        } else {
            const long long gx = col ? 0 : (long long)gapcost;
            const double pen = K.skip + K.log2cache[gx < K.l2c_size ? gx : K.l2c_size];
            tn = Sj + db - pen;
        }
        const double t = col ? tc : tn;
        return bonus <= 0 ? -1e300 : t;                    // only an overlap brings the bonus to zero or below (:27425): evaluated to nothing, still counted
    } else {
        double tc = Sj + db - K.s_gapcost[col ? gapcost : 0] - (double)K.s_rgc[col ? readgap : 0];
        const double f2 = w.fp + db;
        const bool owes = w.fp < 0;
        tc = (owes && f2 >= 0) ? tc + w.pp : tc;                                // refund (:23557-23559)
        const double tn = Sj + db - K.skip;                                     // :23577-23578
        nfp = col ? ((owes && f2 < 0) ? f2 : 0.0) : -K.skip + db;
        npp = col ? ((owes && f2 < 0) ? w.pp : 0.0) : K.skip;
        return bonus <= 0 ? -1e300 : (col ? tc : tn);
    }
}

// One block of 16 candidates in scan order (lane 0 first). A lane without an entry (w.j = VMX_RW_NONE) ends the index; an entry with j >= te
// is passed over. Carries the sequential loop's state: max_scores / pre_index (/ the penalties the anchor inherits), opcount. Returns the lane
// of the first candidate that ends the loop (16: none).
template <int KIND>
__device__ __forceinline__ int vmx_rw_scan16(const vmx_rcur& c, const vmx_rw_costs& K, const vmx_tables& tab, const vmx_rwin& w,
                                             double& max_scores, int& pre_index, double& fp_i, double& pp_i, int& opcount) {
    constexpr bool gc = vmx_rw_traits<KIND>::gc;
    double nfp, npp;
    const double tv = vmx_rw_test<KIND>(c, K, tab, w, nfp, npp);
    const bool vis = w.j < c.te;                                              // (VMX_RW_NONE is above every te)
    const double test = vis ? tv : VMX_F64_NEG;
    const double incl = vmx_row_incl_max0_f64(test);                          // prefix max of the candidates' scores, in scan order (of max(score, 0): max_scores > 0)
    const double m_before = vmx_max_f64(vmx_row_shr0_f64<1>(incl), max_scores);      // the running max the sequential loop holds at this candidate
    const double lim = m_before - c.dl;
    // GC :24940 breaks on S[j] <= max - l_i, LC :27415 on S[j] < max - l_i
    const vmx_rmask vmk = vmx_mask(vis), emk = vmx_mask(w.j == VMX_RW_NONE), cmk = gc ? vmx_mask(!(w.S > lim)) : vmx_mask(w.S < lim);
    const unsigned bm = vmx_row_bits((vmk & cmk) | emk), vm = vmx_row_bits(vmk);
    const unsigned im = vmx_row_bits(vmx_mask(test > m_before));              // strict >: the candidates at which the sequential loop updates (an entry out of sight holds -inf)
    const int first = __ffs(bm | 0x10000u) - 1;
    const unsigned below = (1u << first) - 1u;
    // opcount: GC counts the candidates that pass the test, LC counts before the test, i.e. the breaking candidate too (:27410-27415) — the
    // lane that stopped the scan, when it is an entry in sight and not the end of the index
    opcount += __popc(vm & below);
    if constexpr (!gc) opcount += (int)((vm >> first) & 1u);
    const unsigned upd = im & below;
#ifdef VMX_RW_DPP_WINNER
    // the last update wins: its score is the largest of the candidates before the break (every update is an increase), its lane the highest
    // bit of `upd`; both reach every lane of the row by a butterfly of DPP rotations (no LDS round trip on the per-anchor chain)
    const int l16 = vmx_lane() & 15;
    const int wl = 31 - __clz((int)(upd | 1u));
    const double gM = vmx_row_allmax_f64(((upd >> l16) & 1u) ? test : 0.0);
    const int gj = vmx_row_allmax_i32(l16 == wl ? w.j : -1);
    max_scores = upd ? gM : max_scores; pre_index = upd ? gj : pre_index;
    if constexpr (vmx_rw_traits<KIND>::pen) { const double gf = vmx_row_get_f64(nfp, wl), gp = vmx_row_get_f64(npp, wl); fp_i = upd ? gf : fp_i; pp_i = upd ? gp : pp_i; }
#else
    // the last update wins; fetched whether there is one or not (lane 0 then, dropped): three or seven ds_bpermute instead of a branch
    const int wl = upd ? 31 - __clz((int)upd) : 0;
    const double gM = vmx_row_get_f64(test, wl); const int gj = vmx_row_get_i32(w.j, wl);
    max_scores = upd ? gM : max_scores; pre_index = upd ? gj : pre_index;
    if constexpr (vmx_rw_traits<KIND>::pen) { const double gf = vmx_row_get_f64(nfp, wl), gp = vmx_row_get_f64(npp, wl); fp_i = upd ? gf : fp_i; pp_i = upd ? gp : pp_i; }
#endif
    return first;
}

// number of x in [0, k) with S[SA[x]] < target (le: <= target); S ascending along SA. 16-ary search by the row's 16 lanes.
__device__ __forceinline__ int vmx_rw_sorted_count(const double* S, const int* SA, int k, double target, bool le, int l16) {
    int lo = 0, hi = k;                   // elements [0, lo) qualify, elements [hi, k) do not
    while (hi - lo > 16) {
        const int stride = (hi - lo + 15) >> 4;
        const int x = lo + (l16 + 1) * stride - 1;
        bool in = false;
        if (x < hi) { const double v = S[SA[x]]; in = le ? v <= target : v < target; }
        const int cnt = __popc(vmx_row_ballot(in));
        const int nlo = lo + cnt * stride;
        int nhi = nlo + stride - 1; if (nhi > hi) nhi = hi;
        lo = nlo; hi = nhi;
    }
    const int x = lo + l16;
    bool in = false;
    if (x < hi) { const double v = S[SA[x]]; in = le ? v <= target : v < target; }
    return lo + __popc(vmx_row_ballot(in));
}
// SA[loc + 1 : k + 1] = SA[loc : k]; SA[loc] = k, by the row's 16 lanes from the top down
__device__ __forceinline__ void vmx_rw_sarg_insert(int* SA, int loc, int k, int l16) {
    for (int hi = k; hi > loc; hi -= 16) {
        const int x = hi - l16;
        int v = 0;
        if (x > loc) v = SA[x - 1];
        vmx_row_sync();
        if (x > loc) SA[x] = v;
        vmx_row_sync();
    }
    if (l16 == 0) SA[loc] = k;
    vmx_row_sync();
}
// the reference's insertpoint_score (:19369-19387) replayed on a = #scores < target and b = #scores <= target (k_chain.hip)
__device__ __forceinline__ int vmx_rw_bisect_replay(int a, int b, int k) {
    int i = 0, j = k;
    while (i < j) {
        const int mid = (i + j) >> 1;
        if (mid < a) i = mid + 1;
        else if (mid >= b) j = mid;
        else return mid + 1;
    }
    return j;
}

__device__ __forceinline__ void vmx_rw_load_entry(vmx_rwin& w, int j, const vmx_anchor* A, const double* S, const double* FP, const double* PP, bool pen) {
    const vmx_anchor a = A[j];
    w.j = j; w.q = a.q; w.ls = VMX_RW_LS(a.l, a.s); w.S = S[j]; w.r = a.r;
    if (pen) { w.fp = FP[j]; w.pp = PP[j]; }
}

// The loop over one read's anchors, for one row. A: the read's anchors in the order the variant wants them (GC: by read position; LC-exact /
// LC-mm: by read end; `_scar`: by read start); S / P / SA: the read's arrays in HBM (outputs of GC, scratch of LC); COV: GC's coverage bytes.
// Returns the index of the best anchor (GC: -1 after the opcount bail-out :24914; LC: -1 when the reference would switch to its *_fast twin :27380).
// WW: entries the window holds (16; the CPU-emulator tests also run 3, so that the rare paths are the common ones there: VMX_RW_WIN)
template <int KIND, int WW>
__device__ __forceinline__ int vmx_rw_chain(const vmx_anchor* __restrict__ A, int n, const vmx_rw_costs& K, const vmx_tables& tab, double oskipcost, int omaxdiff,
                                            double* __restrict__ S, int32_t* __restrict__ P, int32_t* __restrict__ SA, const uint8_t* __restrict__ COV,
                                            double* __restrict__ FP, double* __restrict__ PP, double& best_score, long long& opcount_out, unsigned long long& slow_out) {
    constexpr bool gc = vmx_rw_traits<KIND>::gc;
    constexpr bool pen = vmx_rw_traits<KIND>::pen;
    constexpr bool cov = KIND == 0;
    const int l16 = vmx_lane() & 15;
    // anchors [bb, bb + 16) in registers (lane t: anchor bb + t), the next block already on its way — kept as loaded until it takes over
    // (anything computed from it at once would make the wave wait for the load, and for every store before it, right there)
    int bq, bls; long long br;
    vmx_anchor nba; int nbc = 0;
    { const int x = l16 < n ? l16 : n - 1; const vmx_anchor a = A[x]; bq = a.q; bls = VMX_RW_LS(a.l, a.s) | (cov ? (int)COV[x] << 17 : 0); br = a.r; }
    { const int x = 16 + l16 < n ? 16 + l16 : n - 1; nba = A[x]; if (cov) nbc = COV[x]; }
    const int q0 = vmx_row_get_i32(bq, 0), ls0 = vmx_row_get_i32(bls, 0); const long long r0 = vmx_row_get_i64(br, 0);
    const int l0 = ls0 & 0xffff;
    vmx_rwin win;
    win.j = l16 == 0 ? 0 : VMX_RW_NONE; win.q = q0; win.ls = ls0 & 0x1ffff; win.S = l16 == 0 ? (double)l0 : VMX_F64_NEG; win.r = r0; win.fp = 0.0; win.pp = 0.0;
    if (l16 == 0) { SA[0] = 0; S[0] = (double)l0; P[0] = VMX_NOPRE; if (pen) { FP[0] = 0.0; PP[0] = 0.0; } }
    vmx_rcur c;
    c.te = 1; c.skipcost = cov ? oskipcost + (double)(ls0 >> 17) : oskipcost; c.maxdiff = omaxdiff;
    if (cov) { c.maxdiff = omaxdiff - (ls0 >> 17); if (c.maxdiff < 10) c.maxdiff = 10; }
    int prereadloc = gc ? q0 : q0 + l0;
    double g_max_scores = (double)l0; int g_max_index = 0;
    int opcount = 0;                     // (bounded by the bail-out rules below: 1000 per anchor)
    unsigned long long slow = 0;          // tuning counters: scans that left the window (low word), insertions through HBM (high word)
    bool bailed = false;
    // anchor 1, fetched a step ahead
    int nq = vmx_row_get_i32(bq, 1), nls = vmx_row_get_i32(bls, 1); long long nr = vmx_row_get_i64(br, 1);
    for (int i = 1; i < n; ++i) {
        c.q = nq; c.l = nls & 0xffff; c.sneg = (nls >> 16) & 1; c.r = nr; c.dl = (double)c.l;
        const int covi = nls >> 17;
        {   // anchor i + 1 for the next step
            const int x = i + 1;
            if ((x & 15) == 0) {
                bq = nba.q; bls = VMX_RW_LS(nba.l, nba.s) | (cov ? nbc << 17 : 0); br = nba.r;
                const int y = x + 16 + l16 < n ? x + 16 + l16 : n - 1;
                nba = A[y]; if (cov) nbc = COV[y];
            }
            nq = vmx_row_get_i32(bq, x & 15); nls = vmx_row_get_i32(bls, x & 15); nr = vmx_row_get_i64(br, x & 15);
        }
        const int key = gc ? c.q : c.q + c.l;
        const bool adv = prereadloc < key;
        // (opcount / i > 1000 in doubles <=> opcount > 1000 i for these magnitudes: the quotient of two integers below 2^53 that exceeds
        // 1000 does so by at least 1 / i, far above the spacing of doubles at 1000)
        if (gc) { if (adv && (unsigned)opcount > 1000u * (unsigned)i) { bailed = true; break; } }                                  // :24914 max_factor
        else if (KIND != 4) { if (adv && opcount > 100000 && (long long)opcount > 1000LL * (long long)prereadloc) { bailed = true; break; } }      // :27380 -> *_fast
        c.te = adv ? i : c.te;
        if (cov) { int md = omaxdiff - covi; md = md < 10 ? 10 : md; c.skipcost = adv ? oskipcost + (double)covi : c.skipcost; c.maxdiff = adv ? md : c.maxdiff; }
        prereadloc = adv ? key : prereadloc;
        double max_scores = c.dl; int pre_index = VMX_NOPRE; double fp_i = 0.0, pp_i = 0.0;
        int first = vmx_rw_scan16<KIND>(c, K, tab, win, max_scores, pre_index, fp_i, pp_i, opcount);
        if (first >= WW && i > WW) {
            // the scan passed the whole window: on through the index in HBM, 16 candidates per step
            ++slow;
            vmx_row_sync();
            for (int base = i - 1 - WW; base >= 0; base -= 16) {
                const int x = base - l16;
                vmx_rwin w; w.j = VMX_RW_NONE; w.q = 0; w.ls = 0; w.S = 0.0; w.r = 0; w.fp = 0.0; w.pp = 0.0;
                if (x >= 0) vmx_rw_load_entry(w, SA[x], A, S, FP, PP, pen);
                first = vmx_rw_scan16<KIND>(c, K, tab, w, max_scores, pre_index, fp_i, pp_i, opcount);
                if (first < 16) break;
            }
        }
        if (l16 == 0) { S[i] = max_scores; P[i] = pre_index; if (pen) { FP[i] = fp_i; PP[i] = pp_i; } }
        { const bool up = max_scores > g_max_scores; g_max_scores = up ? max_scores : g_max_scores; g_max_index = up ? i : g_max_index; }
        // the new entry's place: k = i entries so far, the window holds the top min(k, 16)
        const unsigned gt = vmx_row_ballot(win.S > max_scores);
        const int above = __popc(gt);
        const int W = i < WW ? i : WW;
        int at = (above < WW && (above < W || W == i)) ? above : -1;             // LC: above its equals (smallorequal + 1, :13229-13265)
        if constexpr (gc) {
            const unsigned ge = vmx_row_ballot(win.S >= max_scores);
            if (gt != ge) {                                                      // equal scores in the window: where the reference's bisection puts the new one
                const int cge = __popc(ge);
                at = -1;
                if (cge < W || W == i) { at = i - vmx_rw_bisect_replay(i - cge, i - above, i); if (at >= WW) at = -1; }      // the run of equal scores ends inside the window
            }
        }
        if (at >= 0) {
            const int sj = vmx_row_shr_i32<1>(win.j, win.j), sq = vmx_row_shr_i32<1>(win.q, win.q), sls = vmx_row_shr_i32<1>(win.ls, win.ls);
            const double sS = vmx_row_shr_f64<1>(win.S, win.S); const long long sr = vmx_row_shr_i64<1>(win.r, win.r);
            const bool sh = l16 > at, here = l16 == at;
            win.j = sh ? sj : (here ? i : win.j); win.q = sh ? sq : (here ? c.q : win.q); win.ls = sh ? sls : (here ? (c.l | (c.sneg << 16)) : win.ls);
            win.S = sh ? sS : (here ? max_scores : win.S); win.r = sh ? sr : (here ? c.r : win.r);
            if constexpr (pen) {
                const double sfp = vmx_row_shr_f64<1>(win.fp, win.fp), spp = vmx_row_shr_f64<1>(win.pp, win.pp);
                win.fp = sh ? sfp : (here ? fp_i : win.fp); win.pp = sh ? spp : (here ? pp_i : win.pp);
            }
            if (WW < 16 && l16 >= WW) { win.j = VMX_RW_NONE; win.S = VMX_F64_NEG; }
            if (l16 <= at) SA[i - l16] = win.j;
        } else {
            // below the window, or among equal scores that reach below it: search and shift in HBM, then the window again from the index
            slow += 1ULL << 32;
            vmx_row_sync();
            int loc;
            if (gc) {
                const int a = vmx_rw_sorted_count(S, SA, i, max_scores, false, l16);
                int b = a;
                if (a < i && S[SA[a]] == max_scores) b = vmx_rw_sorted_count(S, SA, i, max_scores, true, l16);
                loc = vmx_rw_bisect_replay(a, b, i);
            } else loc = vmx_rw_sorted_count(S, SA, i, max_scores, true, l16);
            vmx_rw_sarg_insert(SA, loc, i, l16);
            if (i - loc < WW) {
                const int x = i - l16;
                win.j = VMX_RW_NONE; win.S = VMX_F64_NEG;
                if (x >= 0 && l16 < WW) vmx_rw_load_entry(win, SA[x], A, S, FP, PP, pen);
            }
        }
    }
    // (the reference's closing insertions :25018-25027 have all happened)
    best_score = g_max_scores; opcount_out = (long long)opcount; slow_out = slow;
    return bailed ? -1 : g_max_index;
}

// WW = 16 is the product. The WW = 3 instantiations (k_chain_global_rows_w3 / k_chain_local_rows_w3) are TEST kernels built into the same library: the host
// launches them instead when VMX_RW_WIN=3 is in the environment (read at every launch), so that the rare paths — the scan that leaves the window and walks
// S_arg / S in HBM, the insertion below the window, both behind the row's own stores — are the COMMON ones, on the GPU (store -> load ordering inside one wave
// is what the CPU emulator cannot show: VERDICT r5) as on the emulator.
// ------------------------------------------------------------------------------------------------ G2 GC-exact, four reads per wave
// rlist: the reads of the launch, most anchors first; workgroup (= wavefront) b takes the reads 4 b .. 4 b + 3. rmode: 0 modes H / L / S, 1 mode R.
// dbg (optional, VMX_DBG_CHAIN=1): [0] anchors, [1] scans that left the window, [2] insertions through HBM, [3] opcount; [4..7] the same of k_chain_local_rows
template <int WW>
__device__ __forceinline__ void vmx_chain_global_rows_body(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ aoff,
                                                          const int32_t* __restrict__ rlist, int nlist, const vmx_tables& tab,
                                                          const double* __restrict__ gapcost_list, double oskipcost, int omaxdiff,
                                                          int maxgap, double* __restrict__ S_out, int32_t* __restrict__ P_out,
                                                          int32_t* __restrict__ SA_out, uint8_t* __restrict__ cov_pool,
                                                          int64_t* __restrict__ gmax_out, int64_t* __restrict__ opcount_out, int rmode,
                                                          double* __restrict__ FP_pool, double* __restrict__ PP_pool, unsigned long long* __restrict__ dbg) {
    VMX_SETPRIO(3);
    __shared__ double s_gapcost[64];
    const int lane = vmx_lane(), l16 = lane & 15;
    for (int x = lane; x <= omaxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    const int slot = (int)blockIdx.x * 4 + (lane >> 4);
    if (slot >= nlist) return;
    const int rd = rlist[slot];
    const int64_t a0 = aoff[rd];
    const int n = (int)(aoff[rd + 1] - a0);
    if (n <= 0) { if (l16 == 0) { gmax_out[rd] = -2; opcount_out[rd] = 0; } return; }
    const vmx_anchor* A = anchors + a0;
    uint8_t* COV = cov_pool + a0;
    if (rmode == 0) {
        // coverage (number of anchors sharing the read position, capped at 20: :24865-24868)
        for (int i = l16; i < n; i += 16) {
            const int q = A[i].q;
            int cnt = 1;
            for (int x = i - 1; x >= 0 && A[x].q == q && cnt < 20; --x) ++cnt;
            for (int x = i + 1; x < n && A[x].q == q && cnt < 20; ++x) ++cnt;
            COV[i] = (uint8_t)cnt;
        }
        vmx_row_sync();
    }
    vmx_rw_costs K; K.s_gapcost = s_gapcost; K.s_rgc = nullptr; K.skip = oskipcost; K.maxgap = maxgap; K.l2c_size = 0; K.log2cache = nullptr;
    double best; long long opc; unsigned long long slow; int g;
    if (rmode == 0) g = vmx_rw_chain<0, WW>(A, n, K, tab, oskipcost, omaxdiff, S_out + a0, P_out + a0, SA_out + a0, COV, nullptr, nullptr, best, opc, slow);
    else g = vmx_rw_chain<1, WW>(A, n, K, tab, oskipcost, omaxdiff, S_out + a0, P_out + a0, SA_out + a0, nullptr, FP_pool + a0, PP_pool + a0, best, opc, slow);
    if (l16 == 0) {
        gmax_out[rd] = g; opcount_out[rd] = opc;
        if (dbg) { atomicAdd(&dbg[0], (unsigned long long)n); atomicAdd(&dbg[1], slow & 0xffffffffULL); atomicAdd(&dbg[2], slow >> 32); atomicAdd(&dbg[3], (unsigned long long)opc); }
    }
}


#define VMX_GLOBAL_ROWS_ARGS const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ aoff, const int32_t* __restrict__ rlist, int nlist, vmx_tables tab, \
    const double* __restrict__ gapcost_list, double oskipcost, int omaxdiff, int maxgap, double* __restrict__ S_out, int32_t* __restrict__ P_out, int32_t* __restrict__ SA_out, \
    uint8_t* __restrict__ cov_pool, int64_t* __restrict__ gmax_out, int64_t* __restrict__ opcount_out, int rmode, double* __restrict__ FP_pool, double* __restrict__ PP_pool, \
    unsigned long long* __restrict__ dbg
__global__ void __launch_bounds__(64) k_chain_global_rows(VMX_GLOBAL_ROWS_ARGS) {
    vmx_chain_global_rows_body<16>(anchors, aoff, rlist, nlist, tab, gapcost_list, oskipcost, omaxdiff, maxgap, S_out, P_out, SA_out, cov_pool, gmax_out, opcount_out, rmode, FP_pool, PP_pool, dbg);
}
__global__ void __launch_bounds__(64) k_chain_global_rows_w3(VMX_GLOBAL_ROWS_ARGS) {          // test kernel: 3-entry window (see above)
    vmx_chain_global_rows_body<3>(anchors, aoff, rlist, nlist, tab, gapcost_list, oskipcost, omaxdiff, maxgap, S_out, P_out, SA_out, cov_pool, gmax_out, opcount_out, rmode, FP_pool, PP_pool, dbg);
}

// ------------------------------------------------------------------------------------------------ L3 / L4 / L5 local chain DP, four reads per wave
// rlist: the reads of the launch, most local anchors first; workgroup b takes the reads 4 b .. 4 b + 3. A read runs LC-mm when its guide list held
// more than one chain (n_guides_total > 1, :28583-28590), else LC-exact, `_scar` in mode R; `want` names the variant of this launch (0 / 1 / 2) and
// a row whose read wants another one leaves at once — the variants are different code, and rows of one wave in different variants would run one
// after the other. Traceback with overlap trimming (:27508-27526) by the row's first lane.
template <int WW>
__device__ __forceinline__ void vmx_chain_local_rows_body(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ la_off,
                                                         const int32_t* __restrict__ la_cnt, const int32_t* __restrict__ n_guides_total,
                                                         const int32_t* __restrict__ rlist, int nlist, int want, const vmx_tables& tab,
                                                         const double* __restrict__ gapcost_list, double skip_exact, double skip_mm, int maxdiff,
                                                         int maxgap, int mode, double* __restrict__ S_pool, int32_t* __restrict__ P_pool,
                                                         int32_t* __restrict__ SA_pool, double* __restrict__ out_score,
                                                         vmx_anchor* __restrict__ out_chain, int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant,
                                                         int32_t* __restrict__ status, double* __restrict__ FP_pool, double* __restrict__ PP_pool,
                                                         unsigned long long* __restrict__ dbg, const int32_t* __restrict__ nlist_dev) {
    VMX_SETPRIO(3);
    if (nlist_dev) nlist = *nlist_dev;                 // the list was built on the device (the local stage without a host wait): the launch covers an upper bound
    __shared__ double s_gapcost[64];
    __shared__ float s_rgc[128];
    const int lane = vmx_lane(), l16 = lane & 15;
    for (int x = lane; x <= maxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    {
        const float* rgc_g = want == 1 ? tab.large_readgap : (mode == 3 ? tab.readgap_r : tab.readgap_h);
        for (int x = lane; x < 100; x += 64) s_rgc[x] = rgc_g[x];          // read-gap cost table of this launch's variant (100 entries, maxgap <= 99)
    }
    __syncthreads();
    const int slot = (int)blockIdx.x * 4 + (lane >> 4);
    if (slot >= nlist) return;
    const int rd = rlist[slot];
    const int64_t a0 = la_off[rd];
    const int n = la_cnt[rd];
    const int var = mode == 3 ? 2 : (n_guides_total[rd] > 1 ? 1 : 0);
    if (var != want) return;
    if (n <= 0) { if (l16 == 0) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_RAISED_DEV; } return; }   // np.array([]) indexing raises
    const vmx_anchor* A = anchors + a0;
    vmx_rw_costs K; K.s_gapcost = s_gapcost; K.s_rgc = s_rgc; K.skip = want == 1 ? skip_mm : skip_exact; K.maxgap = maxgap;
    K.l2c_size = (long long)tab.log2cache_n - 1; K.log2cache = tab.log2cache;
    double* S = S_pool + a0; int32_t* P = P_pool + a0; int32_t* SA = SA_pool + a0;
    double best = 0.0; long long opc = 0; unsigned long long slow = 0; int g;
    if (want == 0) g = vmx_rw_chain<2, WW>(A, n, K, tab, K.skip, maxdiff, S, P, SA, nullptr, nullptr, nullptr, best, opc, slow);
    else if (want == 1) g = vmx_rw_chain<3, WW>(A, n, K, tab, K.skip, maxdiff, S, P, SA, nullptr, nullptr, nullptr, best, opc, slow);
    else g = vmx_rw_chain<4, WW>(A, n, K, tab, K.skip, maxdiff, S, P, SA, nullptr, FP_pool + a0, PP_pool + a0, best, opc, slow);
    vmx_row_sync();                                   // (P, written through the loop without waiting, is read back by the traceback)
    if (g < 0) { if (l16 == 0) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_FASTPATH_DEV; } }
    else {
        // Traceback with overlap trimming (:27508-27526) by the whole row. The reference walks take = P[take] from the best anchor; one lane doing
        // that pays a round trip to HBM per chain node (two dependent loads: 5-7 ms for the 7000-node chain of a 100 kb read, most of the kernel's
        // time on a batch of long reads). Predecessors have smaller indices and sit a few places below as a rule, so the walk goes down the
        // anchors in blocks of 16: the block's P and anchors are loaded by the 16 lanes at once (the next block down already on its way),
        // the hops inside a block are register reads (ds_bpermute), the chain nodes found in it are written in one step — each trimmed by the
        // node that follows it in the walk (its own predecessor: the reference rewrites O[w - 1] when it reaches `now`), at the place the serial
        // walk would have written it.
        vmx_anchor* O = out_chain + a0;
        int w = 0, cur = g;
        int b = cur & ~15;
        int Pn = VMX_NOPRE; vmx_anchor An; An.q = 0; An.l = 0; An.s = 0; An.r = 0;
        { const int x = b + l16; if (x < n) { Pn = P[x]; An = A[x]; } }
        while (cur != VMX_NOPRE) {
            const int Pt = Pn; const vmx_anchor At = An;
            const int bnext = b - 16;
            if (bnext >= 0) { Pn = P[bnext + l16]; An = A[bnext + l16]; }              // the block below: the walk goes there next as a rule
            unsigned m = 0;
            int nxt;
            while (true) {
                m |= 1u << (cur - b);
                nxt = vmx_row_get_i32(Pt, cur - b);
                if (nxt == VMX_NOPRE || nxt < b) break;
                cur = nxt;
            }
            const bool member = (m >> l16) & 1u;
            if (member) {
                vmx_anchor t = At;
                if (Pt != VMX_NOPRE) {
                    const vmx_anchor now = A[Pt];
                    const int nowl = (int)now.l & 0xffff;
                    if (At.q < now.q + nowl) { const int ov = now.q + nowl - At.q; t.q = At.q + ov; t.l = (int16_t)(((int)At.l & 0xffff) - ov); if (At.s == 1) t.r = At.r + ov; }
                }
                O[w + __popc(m >> (l16 + 1))] = t;
            }
            w += __popc(m);
            cur = nxt;
            if (cur != VMX_NOPRE) {
                const int nb2 = cur & ~15;
                if (nb2 != bnext) { Pn = P[nb2 + l16]; An = A[nb2 + l16]; }            // a longer jump: that block instead
                b = nb2;
            }
        }
        if (l16 == 0) { out_len[rd] = w; out_score[rd] = best; status[rd] = 0; }
    }
    if (l16 == 0) {
        out_variant[rd] = want;
        if (dbg) { atomicAdd(&dbg[4], (unsigned long long)n); atomicAdd(&dbg[5], slow & 0xffffffffULL); atomicAdd(&dbg[6], slow >> 32); atomicAdd(&dbg[7], (unsigned long long)opc); }
    }
}

#define VMX_LOCAL_ROWS_ARGS const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ la_off, const int32_t* __restrict__ la_cnt, const int32_t* __restrict__ n_guides_total, \
    const int32_t* __restrict__ rlist, int nlist, int want, vmx_tables tab, const double* __restrict__ gapcost_list, double skip_exact, double skip_mm, int maxdiff, int maxgap, int mode, \
    double* __restrict__ S_pool, int32_t* __restrict__ P_pool, int32_t* __restrict__ SA_pool, double* __restrict__ out_score, vmx_anchor* __restrict__ out_chain, \
    int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant, int32_t* __restrict__ status, double* __restrict__ FP_pool, double* __restrict__ PP_pool, unsigned long long* __restrict__ dbg, \
    const int32_t* __restrict__ nlist_dev
__global__ void __launch_bounds__(64) k_chain_local_rows(VMX_LOCAL_ROWS_ARGS) {
    vmx_chain_local_rows_body<16>(anchors, la_off, la_cnt, n_guides_total, rlist, nlist, want, tab, gapcost_list, skip_exact, skip_mm, maxdiff, maxgap, mode, S_pool, P_pool, SA_pool, out_score,
                                  out_chain, out_len, out_variant, status, FP_pool, PP_pool, dbg, nlist_dev);
}
__global__ void __launch_bounds__(64) k_chain_local_rows_w3(VMX_LOCAL_ROWS_ARGS) {           // test kernel: 3-entry window (see above)
    vmx_chain_local_rows_body<3>(anchors, la_off, la_cnt, n_guides_total, rlist, nlist, want, tab, gapcost_list, skip_exact, skip_mm, maxdiff, maxgap, mode, S_pool, P_pool, SA_pool, out_score,
                                 out_chain, out_len, out_variant, status, FP_pool, PP_pool, dbg, nlist_dev);
}
