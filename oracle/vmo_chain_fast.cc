// vmo_chain_fast.cc — CPU ORACLE (test infrastructure): the heuristic "_fast" chain variants.
//
// Restates the LIVE definitions in /root/reference/src/vacmap/mammap_clrnano.py:
//   insertpoint_score_distance                                                       :17200-17226
//   closest2targetdistance                                                           :17228-17252
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_fast_all       :25033-25339   (G3, "GC-fast")
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_fast             :26938-27303   (L5, "LC-fast")
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_mismatch_fast    :27891-28249   (L5, "LC-mm-fast")
// All three keep the candidates bucketed by int(S) in an index sorted by (int(S), diagonal key); a bucket of more than fast_t = 5
// entries is represented by the entry whose diagonal key is closest to the current anchor's. Ported literally, including
//   * the visit order (buckets from max_score_i downwards; inside a small bucket from its end to its start),
//   * max_score_i starting at 0 and never seeing anchor 0 (:25111, :25136): when anchor 0 outscores everything inserted later the
//     bucket boundaries are off by one, exactly as in the reference,
//   * S_i[i] = max_scores truncating toward zero (:25314).
// Two places where the reference itself misbehaves are reported as a raised read (negative status) instead:
//   * an integer score outside S_i_count (IndexError under CPython, out-of-bounds under numba),
//   * LC-fast's `continue` at :27104 / :27191's twin in the large-bucket branch re-enters the while loop without changing its
//     state, i.e. the reference does not terminate.
#include "vmo_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>

namespace vmo {

static const int64_t NOPRE = -9999999;
static std::atomic<int64_t> g_fast_calls[3];     // how often GC-fast / LC-fast / LC-mm-fast ran (tests: which paths a case exercised)

// :17200-17226 — literal
static int64_t insertpoint_score_distance(const int64_t* S_i, int64_t target_score, int64_t k, const int64_t* S_arg_i, int64_t target_distance,
                                          const int64_t* distance_arr) {
    int64_t i = 0, j = k;
    if (S_i[S_arg_i[0]] > target_score) return 0;
    if (S_i[S_arg_i[k - 1]] < target_score) return k;
    while (i < j) {
        const int64_t mid = (i + j) / 2;
        const int64_t now_score = S_i[S_arg_i[mid]];
        if (now_score < target_score) i = mid + 1;
        else if (now_score > target_score) j = mid;
        else {
            const int64_t now_distance = distance_arr[S_arg_i[mid]];
            if (now_distance < target_distance) i = mid + 1;
            else if (now_distance > target_distance) j = mid;
            else return mid + 1;
        }
    }
    return j;
}

// :17228-17252 — literal
static int64_t closest2targetdistance(int64_t target_distance, const int64_t* distance_arr, const int64_t* S_arg_i, int64_t st_loc, int64_t en_loc) {
    int64_t i = st_loc, j = en_loc;
    if (distance_arr[S_arg_i[i]] >= target_distance) return i;
    if (distance_arr[S_arg_i[j - 1]] <= target_distance) return j - 1;
    while (i < j) {
        const int64_t mid = (i + j) / 2;
        const int64_t now_distance = distance_arr[S_arg_i[mid]];
        if (now_distance < target_distance) i = mid + 1;
        else if (now_distance > target_distance) j = mid;
        else return mid;
    }
    if ((target_distance - distance_arr[S_arg_i[j - 1]]) < (distance_arr[S_arg_i[j]] - target_distance)) return j - 1;
    return j;
}

// gap geometry :25191-25219 / :27098-27129; lc = true adds LC's `bonus <= 0` test (returns false when it fires)
static inline bool geometry(const Anchor& ai, const Anchor& aj, bool lc, int64_t& readgap, int64_t& refgap, int64_t& bonus) {
    readgap = ai.q - aj.q - aj.l;
    if (readgap < 0) {
        bonus = ai.q + ai.l - aj.q - aj.l;
        if (lc && bonus <= 0) return false;
        readgap = 0;
        const int64_t overlap = aj.q + aj.l - ai.q;
        if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r + overlap - (aj.r + aj.l); else refgap = aj.r - (ai.r + bonus); }
        else { if (aj.s == -1) refgap = ai.r + overlap - aj.r + 1; else refgap = ai.r + bonus - 1 - (aj.r + aj.l); }
    } else {
        bonus = ai.l;
        if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r - aj.r - aj.l; else refgap = aj.r - ai.r - ai.l; }
        else { if (aj.s == -1) refgap = ai.r - aj.r + 1; else refgap = ai.r + ai.l - 1 - aj.r - aj.l; }
    }
    return true;
}

// the shared skeleton. variant 0 = GC-fast, 1 = LC-fast, 2 = LC-mm-fast, 3 = GC-fast of mode R (mammap_noprefercloser.py:23059-23417: no
// coverage terms, fixed skipcost with refund). Returns g_max_index, or < -1 on the two misbehaviours above.
// variant 4 = GC-fast of the -mode asm fork (mammap_asm.py:20738-21037: no coverage terms, gap_geometry_asm, plain skipcost + extra) and, with
// link->n_pre > 0, its LINKED form (:21871-22158: the first n_pre rows carry S / P from the previous batch, S_i = int(pre_S), the bucket index
// starts with row 0 only, max_score_i = S_i[0]).
static int64_t fast_dp(const std::vector<Anchor>& A, int variant, int kmersize, double oskipcost, int omaxdiff, int maxgap, int mode,
                       std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg_i, double* g_max_out, const LinkState* link = nullptr) {
    const Tables& T = tables();
    const int64_t extra_size = (int64_t)T.extra.size() - 1;
    const int64_t log2cache_size = (int64_t)T.log2cache.size() - 1;
    const int64_t repeat_weight = 20;
    const int64_t fast_t = 5;
    const int64_t n = (int64_t)A.size();
    const bool lc = variant == 1 || variant == 2;
    const bool gcr = variant == 3;
    const bool asmv = variant == 4;
    g_fast_calls[(gcr || asmv) ? 0 : variant].fetch_add(1);
    std::vector<double> pre_pen(gcr ? A.size() : 0, 0.0), fixed_pen(gcr ? A.size() : 0, 0.0);
    std::vector<double> gapcost_list(omaxdiff + 1, 0.0);
    for (int g = 1; g <= omaxdiff; ++g) {
        if (!lc || g <= 10) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);     // :25052 / :26956
        else gapcost_list[g] = (0.01 * kmersize * g + 2 * T.log2int[g]);                       // :26958
    }
    const std::vector<float>& readgapcost = variant == 2 ? T.large_readgap : (mode == VMO_MODE_R ? T.readgap_r : T.readgap_h);
    S.assign(n, 0.0); P.assign(n, 0); S_arg_i.assign(n, 0);
    std::vector<int64_t> S_i(n, 0), target_arr(n, 0);
    const int64_t lastq = A[n - 1].q;
    std::vector<int64_t> cov(lastq + (lc ? 5000 : 1), 0);                                      // :25070 / :26977
    const int64_t readlength = lastq + 1000;
    for (int64_t i = 0; i < n; ++i) {
        if (A[i].q < 0 || A[i].q >= (int64_t)cov.size()) return -5;
        cov[A[i].q] = std::min(cov[A[i].q] + 1, repeat_weight);
        if (A[i].s == 1) target_arr[i] = A[i].r - A[i].q + readlength;
        else target_arr[i] = -(A[i].r + A[i].q + readlength);
    }
    int64_t prereadloc = lc ? A[0].q + A[0].l : A[0].q;
    int64_t pre_size = 1;
    double skipcost = oskipcost;                      // GC-fast: updated with the coverage on every position advance; LC-fast: never
    int64_t maxdiff = omaxdiff;
    int64_t testspace_en_i = 1;
    S_arg_i[0] = 0;
    S[0] = (double)A[0].l; S_i[0] = A[0].l; P[0] = NOPRE;
    double g_max_scores = (double)A[0].l; int64_t g_max_index = 0;
    std::vector<int64_t> S_i_count(lastq + 50, 0);
    if (A[0].l < 0 || A[0].l >= (int64_t)S_i_count.size()) return -5;
    S_i_count[A[0].l] = 1;
    int64_t max_score_i = 0;
    if (link && link->n_pre > 0) {                   // :21893-21901
        S_i_count[A[0].l] = 0;
        for (int64_t i = 0; i < link->n_pre; ++i) { S[i] = link->pre_S[i]; S_i[i] = (int64_t)link->pre_S[i]; P[i] = link->pre_P[i]; }
        pre_size = link->n_pre; g_max_scores = link->g_max_scores; g_max_index = link->g_max_index; prereadloc = link->prereadloc;
        if (S_i[0] < 0 || S_i[0] >= (int64_t)S_i_count.size()) return -5;
        S_i_count[S_i[0]] = 1;
        max_score_i = S_i[0];
    }
    auto insert_pending = [&](int64_t upto) -> bool {
        int64_t k = testspace_en_i;
        while (k < upto) {
            if (S_i[k] < 0 || S_i[k] >= (int64_t)S_i_count.size()) return false;
            S_i_count[S_i[k]] += 1;
            if (S_i[k] > max_score_i) max_score_i = S_i[k];
            const int64_t loc = insertpoint_score_distance(S_i.data(), S_i[k], k, S_arg_i.data(), target_arr[k], target_arr.data());
            memmove(S_arg_i.data() + loc + 1, S_arg_i.data() + loc, sizeof(int64_t) * (size_t)(k - loc));
            S_arg_i[loc] = k;
            ++k;
        }
        testspace_en_i = k;
        return true;
    };
    for (int64_t i = pre_size; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        const int64_t pos_i = lc ? A[i].q + A[i].l : A[i].q;
        if (prereadloc < pos_i) {
            if (!insert_pending(i)) return -5;
            if (!lc && !gcr && !asmv) {
                skipcost = oskipcost + (double)cov[A[i].q];                                     // :25151
                maxdiff = std::max<int64_t>(omaxdiff - cov[A[i].q], 10);                        // :25152
            }
            prereadloc = pos_i;
        }
        int64_t c_score_i = max_score_i;
        int64_t st_loc = testspace_en_i, en_loc = testspace_en_i;
        const int64_t f_kmersize = A[i].l + 1;
        while ((double)c_score_i > (max_scores - (double)f_kmersize)) {
            const int64_t now_count = S_i_count[c_score_i];            // c_score_i >= 0 here: max_scores - f_kmersize >= -1
            if (now_count == 0) { --c_score_i; continue; }
            st_loc = en_loc - now_count;
            if (st_loc < 0) return -5;
            auto eval = [&](int64_t j, bool& hang) {
                int64_t readgap, refgap, bonus;
                if (asmv) gap_geometry_asm(A[i], A[j], readgap, refgap, bonus);
                else if (!geometry(A[i], A[j], lc, readgap, refgap, bonus)) { hang = true; return; }
                int64_t gapcost = std::llabs(readgap - refgap);
                double test;
                if (gcr) {
                    if (A[i].s == A[j].s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                        test = S[j] + (double)bonus - gapcost_list[gapcost];
                        if (fixed_pen[j] < 0 && (fixed_pen[j] + (double)bonus) >= 0) test += pre_pen[j];
                        if (test > max_scores) {
                            max_scores = test; pre_index = j;
                            if (fixed_pen[j] < 0 && (fixed_pen[j] + (double)bonus) < 0) { fixed_pen[i] = fixed_pen[j] + (double)bonus; pre_pen[i] = pre_pen[j]; }
                            else { fixed_pen[i] = 0; pre_pen[i] = 0; }
                        }
                    } else {
                        const double tmp_penalty = skipcost;
                        test = S[j] + (double)bonus - tmp_penalty;
                        if (test > max_scores) { max_scores = test; pre_index = j; fixed_pen[i] = -tmp_penalty + (double)bonus; pre_pen[i] = tmp_penalty; }
                    }
                    return;
                }
                if (A[i].s == A[j].s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                    if (!lc) test = S[j] + (double)bonus - gapcost_list[gapcost];
                    else test = S[j] + (double)bonus - gapcost_list[gapcost] - (double)readgapcost[readgap];
                } else if (variant == 0 || asmv) {
                    if (gapcost > extra_size) gapcost = extra_size;
                    test = S[j] - skipcost + (double)bonus - (double)T.extra[gapcost];
                } else if (variant == 1) {
                    if (gapcost > extra_size) gapcost = extra_size;
                    double pen;
                    if (A[i].s != A[j].s) pen = std::min(50.0, skipcost) + (double)T.extra[gapcost];
                    else pen = skipcost + (double)T.extra[gapcost];
                    test = S[j] + (double)bonus - pen;
                } else {
                    const double pen = skipcost + T.log2cache[std::min<int64_t>(log2cache_size, gapcost)];
                    test = S[j] + (double)bonus - pen;
                }
                if (test > max_scores) { max_scores = test; pre_index = j; }
            };
            if (now_count > fast_t) {
                const int64_t j = S_arg_i[closest2targetdistance(target_arr[i], target_arr.data(), S_arg_i.data(), st_loc, en_loc)];
                bool hang = false;
                eval(j, hang);
                if (hang) return -4;              // the reference's `continue` re-enters the loop with unchanged state: it never returns
            } else {
                for (int64_t tmp_j = en_loc - 1; tmp_j >= st_loc; --tmp_j) {
                    bool skip = false;
                    eval(S_arg_i[tmp_j], skip);   // here `continue` only skips the candidate
                }
            }
            en_loc = st_loc;
            --c_score_i;
        }
        S[i] = max_scores;
        S_i[i] = (int64_t)max_scores;             // truncation toward zero
        P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    if (!lc && !insert_pending(n)) return -5;      // :25324-25336 (the LC variants return a path, not the index)
    if (g_max_out) *g_max_out = g_max_scores;
    return g_max_index;
}

// G3 :25033-25339. A sorted by q (stable). S_arg comes back ordered by (int(S), diagonal key) — hit2work_1 peels in that order (:25339).
int64_t chain_global_fast(const std::vector<Anchor>& A, int kmersize, double oskipcost, int omaxdiff, int maxgap,
                          std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg, bool rmode) {
    const int64_t g = fast_dp(A, rmode ? 3 : 0, kmersize, oskipcost, omaxdiff, maxgap, 0, S, P, S_arg, nullptr);
    if (g < 0) { set_error(g == -4 ? "GC-fast: reference does not terminate on this input" : "GC-fast: integer score outside S_i_count"); return -2; }
    return g;
}

int64_t chain_global_fast_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, const LinkState* link,
                              std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg) {
    const int64_t g = fast_dp(A, 4, kmersize, skipcost, maxdiff, maxgap, 0, S, P, S_arg, nullptr, link);
    if (g < 0) { set_error(g == -4 ? "GC-fast (asm): reference does not terminate on this input" : "GC-fast (asm): integer score outside S_i_count"); return -2; }
    return g;
}

// L5 :26938-27303 (mismatch = false) / :27891-28249 (mismatch = true). A sorted by q+l (stable).
int local_chain_fast(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, bool mismatch, int mode,
                     double* score, Path& path) {
    std::vector<double> S; std::vector<int64_t> P, SA;
    double gmax = 0.0;
    const int64_t g = fast_dp(A, mismatch ? 2 : 1, kmersize, skipcost, maxdiff, maxgap, mode, S, P, SA, &gmax);
    if (g < 0) { set_error(g == -4 ? "LC-fast: reference does not terminate on this input" : "LC-fast: integer score outside S_i_count"); return -3; }
    // traceback with overlap trimming :27283-27301
    path.clear();
    int64_t take = g;
    path.push_back(A[take]);
    Anchor preitem = A[take];
    while (P[take] != NOPRE) {
        take = P[take];
        const Anchor& now = A[take];
        if (preitem.q < now.q + now.l) {
            const int64_t ov = now.q + now.l - preitem.q;
            if (preitem.s == 1) path.back() = Anchor{preitem.q + ov, preitem.r + ov, preitem.s, preitem.l - ov};
            else path.back() = Anchor{preitem.q + ov, preitem.r, preitem.s, preitem.l - ov};
        }
        path.push_back(now);
        preitem = now;
    }
    *score = gmax;
    return 0;
}

}  // namespace vmo

extern "C" void vmo_fast_counters(int64_t out[3], int reset) {
    for (int i = 0; i < 3; ++i) { out[i] = vmo::g_fast_calls[i].load(); if (reset) vmo::g_fast_calls[i].store(0); }
}
