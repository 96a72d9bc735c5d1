// vmo_chain.cc — CPU ORACLE (test infrastructure): strand flip, global non-linear chain, chain peeling/selection.
//
// Restates, function by function, the LIVE definitions in /root/reference/src/vacmap/mammap_clrnano.py:
//   get_reversed_chain_numpy_rough                                   :21202-21217   (S2)
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_d_all   :24828-25031   (G2, "GC-exact")
//   insertpoint_score                                                :19369-19387
//   hit2work_1 (+ nested select_secondary_alignment)                 :23491-23734   (G1)
//   decode_hit                                                       :23981-24020
// Mode deltas: accept threshold 60 (H) / 40 (L,S,R) (:23650 vs mammap_ccs.py:23649),
// secondary min span 50 / 100 (R).
// Scores are IEEE double, evaluated in the reference's left-to-right order; compile with -ffp-contract=off.
// np.argsort is taken as STABLE (the harness patches the reference the same way).
#include "vmo_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

namespace vmo {

static const int64_t NOPRE = -9999999;

// :21202-21217
bool strand_flip(std::vector<Anchor>& a, int64_t readlen) {
    if (a.size() < 3) return false;
    int64_t neg = 0, pos = 0;
    for (const Anchor& x : a) { if (x.s == -1) ++neg; else if (x.s == 1) ++pos; }
    if (neg > pos) {
        for (Anchor& x : a) { x.q = readlen - x.q - x.l; x.s = -x.s; }
        std::reverse(a.begin(), a.end());
        return true;
    }
    return false;
}

// :19369-19387 — literal port (do not replace by upper_bound: equal keys return mid+1 at the first probe that hits one)
static int64_t insertpoint_score(const double* S, double target, int64_t k, const int64_t* S_arg) {
    int64_t i = 0, j = k;
    if (S[S_arg[0]] > target) return 0;
    if (S[S_arg[k - 1]] < target) return k;
    while (i < j) {
        int64_t mid = (i + j) / 2;
        double now = S[S_arg[mid]];
        if (now < target) i = mid + 1;
        else if (now > target) j = mid;
        else return mid + 1;
    }
    return j;
}

static inline void sarg_insert(int64_t* S_arg, int64_t loc, int64_t k) {
    // S_arg[loc+1 : k+1] = S_arg[loc : k] ; S_arg[loc] = k   (:24923-24924)
    memmove(S_arg + loc + 1, S_arg + loc, sizeof(int64_t) * (size_t)(k - loc));
    S_arg[loc] = k;
}

// gap geometry shared by GC and LC (:24953-24984, :27418-27456). Returns false if LC's `bonus <= 0 -> continue` fires.
static inline void gap_geometry(const Anchor& ai, const Anchor& aj, int64_t& readgap, int64_t& refgap, int64_t& bonus) {
    readgap = ai.q - aj.q - aj.l;
    if (readgap < 0) {
        bonus = ai.q + ai.l - aj.q - aj.l;
        readgap = 0;
        int64_t overlap = aj.q + aj.l - ai.q;
        if (ai.s == aj.s) {
            if (ai.s == 1) refgap = ai.r + overlap - (aj.r + aj.l);
            else refgap = aj.r - (ai.r + bonus);
        } else {
            if (aj.s == -1) refgap = ai.r + overlap - aj.r + 1;
            else refgap = ai.r + bonus - 1 - (aj.r + aj.l);
        }
    } else {
        bonus = ai.l;
        if (ai.s == aj.s) {
            if (ai.s == 1) refgap = ai.r - aj.r - aj.l;
            else refgap = aj.r - ai.r - ai.l;
        } else {
            if (aj.s == -1) refgap = ai.r - aj.r + 1;
            else refgap = ai.r + ai.l - 1 - aj.r - aj.l;
        }
    }
}

// GC-exact :24828-25031. A sorted by q (stable). Returns g_max_index or -1 (bail-out, :24914).
// rmode = true: the mode-R body (mammap_noprefercloser.py:22839-23057): no coverage terms; a non-co-linear step costs the fixed
// skipcost, remembered in fixed_penatly / pre_penatly and refunded once the chain has continued co-linearly for more than skipcost bases.
int64_t chain_global_exact(const std::vector<Anchor>& A, int kmersize, double oskipcost, int omaxdiff, int maxgap,
                           std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg, int64_t* opcount_out, bool rmode) {
    const Tables& T = tables();
    const int64_t extra_size = (int64_t)T.extra.size() - 1;
    const int64_t repeat_weight = 20;
    const int64_t n = (int64_t)A.size();
    S.assign(n, 0.0); P.assign(n, 0); S_arg.assign(n, 0);
    std::vector<double> gapcost_list(omaxdiff + 1, 0.0);
    for (int g = 1; g <= omaxdiff; ++g) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
    std::vector<int64_t> cov(A[n - 1].q + 1, 0);
    for (int64_t i = 0; i < n; ++i) cov[A[i].q] = std::min(cov[A[i].q] + 1, repeat_weight);
    int64_t prereadloc = A[0].q;
    double skipcost = rmode ? oskipcost : oskipcost + (double)cov[A[0].q];
    int64_t maxdiff = rmode ? omaxdiff : std::max<int64_t>(omaxdiff - cov[A[0].q], 10);
    std::vector<double> pre_pen(rmode ? n : 0, 0.0), fixed_pen(rmode ? n : 0, 0.0);
    int64_t testspace_en = 1;
    S_arg[0] = 0;
    S[0] = (double)A[0].l; P[0] = NOPRE;
    double g_max_scores = (double)A[0].l; int64_t g_max_index = 0;
    int64_t opcount = 0;
    for (int64_t i = 1; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        if (prereadloc < A[i].q) {
            if (((double)opcount / (double)i) > 1000.0) { if (opcount_out) *opcount_out = opcount; return -1; }
            for (int64_t k = testspace_en; k < i; ++k) {
                int64_t loc = insertpoint_score(S.data(), S[k], k, S_arg.data());
                sarg_insert(S_arg.data(), loc, k);
            }
            testspace_en = i;
            if (!rmode) {
                skipcost = oskipcost + (double)cov[A[i].q];
                maxdiff = std::max<int64_t>(omaxdiff - cov[A[i].q], 10);
            }
            prereadloc = A[i].q;
        }
        const double li = (double)A[i].l;
        for (int64_t x = testspace_en - 1; x >= 0; --x) {
            const int64_t j = S_arg[x];
            if (S[j] > (max_scores - li)) {
                ++opcount;
                int64_t readgap, refgap, bonus;
                gap_geometry(A[i], A[j], readgap, refgap, bonus);
                int64_t gapcost = std::llabs(readgap - refgap);
                double test;
                if (rmode) {                                            // mammap_noprefercloser.py:22973-23001
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
                    continue;
                }
                if (A[i].s == A[j].s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                    test = S[j] + (double)bonus - gapcost_list[gapcost];
                } else {
                    if (gapcost > extra_size) gapcost = extra_size;
                    test = S[j] - skipcost + (double)bonus - (double)T.extra[gapcost];
                }
                if (test > max_scores) { max_scores = test; pre_index = j; }
            } else break;
        }
        S[i] = max_scores; P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    for (int64_t k = testspace_en; k < n; ++k) {
        int64_t loc = insertpoint_score(S.data(), S[k], k, S_arg.data());
        sarg_insert(S_arg.data(), loc, k);
    }
    if (opcount_out) *opcount_out = opcount;
    return g_max_index;
}

// -mode asm: GC-exact of the older fork, mammap_asm.py:20551-20737 (no coverage terms, gap_geometry_asm), its LINKED form :21686-21870
// (link != nullptr with n_pre > 0: S / P of the first n_pre rows are the carried state, the loop starts behind them, g_max_* and prereadloc come
// from the caller) and the linked LC :21504-21685 (lc: co-linear steps also pay readgapcost_list[readgap] :16536 = the mode-R table; no
// bail-out). A sorted by q (stable) behind the carried rows. Returns g_max_index, or -1 (bail-out :20623 / :21757).
double g_asm_max_factor = 1000.0;     // max_factor (:20623); tests lower it to drive the linked path into its GC-fast (vmo_test_asm_max_factor)
int64_t chain_exact_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, bool lc, const LinkState* link,
                        std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg) {
    const Tables& T = tables();
    const int64_t extra_size = (int64_t)T.extra.size() - 1;
    const int64_t n = (int64_t)A.size();
    S.assign(n, 0.0); P.assign(n, 0); S_arg.assign(n, 0);
    std::vector<double> gapcost_list(maxdiff + 1, 0.0);
    for (int g = 1; g <= maxdiff; ++g) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
    double g_max_scores; int64_t g_max_index, prereadloc, pre_size, testspace_en = 1;
    S_arg[0] = 0;
    if (link && link->n_pre > 0) {
        for (int64_t i = 0; i < link->n_pre; ++i) { S[i] = link->pre_S[i]; P[i] = link->pre_P[i]; }
        pre_size = link->n_pre; g_max_scores = link->g_max_scores; g_max_index = link->g_max_index; prereadloc = link->prereadloc;
    } else {
        S[0] = (double)A[0].l; P[0] = NOPRE;
        g_max_scores = (double)A[0].l; g_max_index = 0; prereadloc = A[0].q; pre_size = 1;
    }
    int64_t opcount = 0;
    for (int64_t i = pre_size; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        if (prereadloc < A[i].q) {
            if (!lc && ((double)opcount / (double)i) > g_asm_max_factor) return -1;
            for (int64_t k = testspace_en; k < i; ++k) {
                const int64_t loc = insertpoint_score(S.data(), S[k], k, S_arg.data());
                sarg_insert(S_arg.data(), loc, k);
            }
            testspace_en = i;
            prereadloc = A[i].q;
        }
        const double li = (double)A[i].l;
        for (int64_t x = testspace_en - 1; x >= 0; --x) {
            const int64_t j = S_arg[x];
            if (S[j] > (max_scores - li)) {
                ++opcount;
                int64_t readgap, refgap, bonus;
                gap_geometry_asm(A[i], A[j], readgap, refgap, bonus);
                int64_t gapcost = std::llabs(readgap - refgap);
                double test;
                if (A[i].s == A[j].s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                    test = S[j] + (double)bonus - gapcost_list[gapcost];
                    if (lc) test = test - (double)T.readgap_r[readgap];
                } else {
                    if (gapcost > extra_size) gapcost = extra_size;
                    test = S[j] - skipcost + (double)bonus - (double)T.extra[gapcost];
                }
                if (test > max_scores) { max_scores = test; pre_index = j; }
            } else break;
        }
        S[i] = max_scores; P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    for (int64_t k = testspace_en; k < n; ++k) {
        const int64_t loc = insertpoint_score(S.data(), S[k], k, S_arg.data());
        sarg_insert(S_arg.data(), loc, k);
    }
    return g_max_index;
}

int64_t chain_global_fast(const std::vector<Anchor>& A, int kmersize, double oskipcost, int omaxdiff, int maxgap,
                          std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg, bool rmode);  // vmo_chain_fast.cc

// hit2work_1 :23491-23734 + decode_hit :23981-24020 (A = output of map(), unsorted)
int decode_hit(std::vector<Anchor> A, int64_t readlen, int kmersize, const vmo_params& prm, ChainSet& out) {
    out = ChainSet();
    out.need_reverse = strand_flip(A, readlen);
    if (A.size() <= 2) return prm.mode == VMO_MODE_R ? -1 : 0;     // :23986; mode R returns an unbound `factor` here (mammap_noprefercloser.py:24417): the read raises
    const int mode = prm.mode;
    const double accept = (mode == VMO_MODE_H) ? 60.0 : 40.0;      // :23650 / mammap_ccs.py:23649
    const int64_t sec_min_span = (mode == VMO_MODE_R) ? 100 : 50;  // :23519 / mammap_noprefercloser.py:23949
    const int64_t bin_size = 100;
    const int64_t n = (int64_t)A.size();
    bool fast_enable = ((double)n / (double)readlen) > 5.0;        // :23570
    std::stable_sort(A.begin(), A.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });  // :23572
    std::vector<double> S; std::vector<int64_t> P, S_arg;
    int64_t g_max_index = 0;
    if (!fast_enable)
        g_max_index = chain_global_exact(A, kmersize, prm.global_skipcost, prm.global_maxdiff, 1000, S, P, S_arg, nullptr, mode == VMO_MODE_R);
    if (fast_enable || g_max_index == -1) {
        fast_enable = true;
        g_max_index = chain_global_fast(A, kmersize, prm.global_skipcost, prm.global_maxdiff, 1000, S, P, S_arg, mode == VMO_MODE_R);
        if (g_max_index < 0) return -2;
    }
    out.fast_used = fast_enable;
    const double scores = S[g_max_index];
    std::vector<char> used(n, 0);
    std::vector<std::vector<Anchor>> path_list;
    std::vector<double> scores_list;
    std::vector<double> S_arr;
    bool hit = false;
    double max_scores = 0;
    {
        std::vector<Anchor> path;
        int64_t take = g_max_index;
        used[take] = 1;
        double score = S[take];
        while (true) {
            path.push_back(A[take]); S_arr.push_back(S[take]);
            if (P[take] == NOPRE) break;
            take = P[take];
            used[take] = 1;
        }
        if (score > 40) { hit = true; scores_list.push_back(score); path_list.push_back(path); }
        if (scores > max_scores) max_scores = scores;
    }
    for (int64_t x = n - 1; x >= 0; --x) {                         // :23617 for take_index in S_arg[::-1]
        int64_t take = S_arg[x];
        if (used[take]) continue;
        std::vector<Anchor> path;
        used[take] = 1;
        double score = S[take];
        while (true) {
            path.push_back(A[take]);
            if (P[take] == NOPRE) break;
            take = P[take];
            if (used[take]) { score = score - S[take]; break; }
            used[take] = 1;
        }
        if (score > 40) { scores_list.push_back(score); path_list.push_back(path); }
    }
    if (!(hit && max_scores > accept)) return 0;                   // unmapped (:23711-23734)
    const int64_t m_paths = (int64_t)path_list.size();
    // order = np.argsort(scores_list)[::-1] with stable argsort
    std::vector<int64_t> order(m_paths);
    for (int64_t i = 0; i < m_paths; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return scores_list[a] < scores_list[b]; });
    std::reverse(order.begin(), order.end());
    if (order[0] != 0) {
        for (int64_t i = 0; i < m_paths; ++i) if (order[i] == 0) { order[i] = order[0]; order[0] = 0; break; }
    }
    auto binset = [&](const std::vector<Anchor>& p) { std::set<int64_t> s; for (const Anchor& a : p) s.insert(a.q / bin_size); return s; };
    std::vector<std::set<int64_t>> prim_sets;
    std::vector<std::vector<double>> prim_scores;
    prim_sets.push_back(binset(path_list[order[0]]));
    prim_scores.push_back({scores_list[order[0]]});
    for (int64_t oi = 1; oi < m_paths; ++oi) {
        int64_t iloc = order[oi];
        std::set<int64_t> b = binset(path_list[iloc]);
        double maxov = 0.; size_t prefer = 0;
        for (size_t p = 0; p < prim_sets.size(); ++p) {
            size_t inter = 0;
            for (int64_t v : b) if (prim_sets[p].count(v)) ++inter;
            double ov = (double)inter / (double)std::min(prim_sets[p].size(), b.size());
            if (ov > maxov) { maxov = ov; prefer = p; }
        }
        if (maxov < 0.5) { prim_sets.push_back(b); prim_scores.push_back({scores_list[iloc]}); }
        else prim_scores[prefer].push_back(scores_list[iloc]);
    }
    const double mlen = (double)path_list[order[0]].size();
    double f1 = prim_scores[0][0], f2 = prim_scores[0].size() < 2 ? 0.0 : prim_scores[0][1];
    {
        double v = 40 * (1 - f2 / f1);
        v = v * std::min(1.0, mlen / 10);
        v = v * std::log(f1);
        int64_t iv = (int64_t)v;   // int() truncates toward zero
        out.mapq = (int)std::min<int64_t>(iv, 60);
    }
    // select_secondary_alignment :23505-23538
    std::vector<std::vector<Anchor>> secondary;
    if (m_paths > 1) {
        std::vector<double> loc2score((size_t)readlen, 0.0);
        int64_t en_loc = readlen;
        for (size_t a = 0; a < path_list[0].size(); ++a) {
            int64_t st_loc = path_list[0][a].q;
            for (int64_t x = st_loc; x < en_loc; ++x) loc2score[x] = S_arr[a];
            en_loc = st_loc;
        }
        for (int64_t oi = 1; oi < m_paths; ++oi) {
            int64_t iloc = order[oi];
            const std::vector<Anchor>& one = path_list[iloc];
            double f2s = scores_list[iloc];
            int64_t en = one.front().q, st = one.back().q;
            if (en - st < sec_min_span) continue;
            double f1s = std::max(loc2score[en] - loc2score[st], 1.0);
            if (f2s / f1s > 0.9 || std::fabs(f1s - f2s) < 40) {
                bool skip = false;
                for (const auto& pri : secondary) {
                    int64_t pen = pri.front().q, pst = pri.back().q;
                    int64_t ov = std::max<int64_t>(std::min(en, pen) - std::max(pst, st), 0);
                    if (((double)ov / (double)(en - st)) > 0.5) { skip = true; break; }
                }
                if (!skip) secondary.push_back(one);
            }
        }
    }
    // decode_hit :24005-24020
    out.all_scores = scores_list;
    out.paths.push_back(path_list[0]);
    for (auto& s : secondary) out.paths.push_back(s);
    out.score = out.need_reverse ? -scores_list[0] : scores_list[0];
    return 0;
}

}  // namespace vmo

using namespace vmo;

extern "C" {

int vmo_strand_flip(int64_t* a, int64_t n, int64_t readlen) {
    std::vector<Anchor> v(n);
    for (int64_t i = 0; i < n; ++i) v[i] = Anchor{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    bool f = strand_flip(v, readlen);
    for (int64_t i = 0; i < n; ++i) { a[4 * i] = v[i].q; a[4 * i + 1] = v[i].r; a[4 * i + 2] = v[i].s; a[4 * i + 3] = v[i].l; }
    return f ? 1 : 0;
}

int64_t vmo_chain_global_raw(const int64_t* a, int64_t n, int mode, int kmersize, double skipcost, int maxdiff, int maxgap,
                             int which, double* S, int64_t* P, int64_t* S_arg) {
    std::vector<Anchor> v(n);
    for (int64_t i = 0; i < n; ++i) v[i] = Anchor{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    std::vector<double> s; std::vector<int64_t> p, sa;
    int64_t g;
    if (which == 0) g = chain_global_exact(v, kmersize, skipcost, maxdiff, maxgap, s, p, sa, nullptr, mode == VMO_MODE_R);
    else g = chain_global_fast(v, kmersize, skipcost, maxdiff, maxgap, s, p, sa, mode == VMO_MODE_R);
    for (int64_t i = 0; i < n && i < (int64_t)s.size(); ++i) { S[i] = s[i]; P[i] = p[i]; S_arg[i] = sa[i]; }
    return g;
}

int vmo_decode_hit(const int64_t* a, int64_t n, int64_t readlen, int kmersize, const vmo_params* prm, vmo_chains* out) {
    std::vector<Anchor> v(n);
    for (int64_t i = 0; i < n; ++i) v[i] = Anchor{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    ChainSet cs;
    int rc = decode_hit(v, readlen, kmersize, *prm, cs);
    memset(out, 0, sizeof(*out));
    if (rc < 0) return rc;
    out->need_reverse = cs.need_reverse; out->mapq = cs.mapq; out->score = cs.score; out->fast_used = cs.fast_used;
    out->n_paths = (int32_t)cs.paths.size();
    size_t tot = 0; for (auto& p : cs.paths) tot += p.size();
    out->path_off = (int64_t*)malloc(sizeof(int64_t) * (cs.paths.size() + 1));
    out->path_anchors = (int64_t*)malloc(sizeof(int64_t) * 4 * (tot ? tot : 1));
    size_t o = 0;
    for (size_t i = 0; i < cs.paths.size(); ++i) {
        out->path_off[i] = (int64_t)o;
        for (const Anchor& x : cs.paths[i]) { int64_t* r = out->path_anchors + 4 * o; r[0] = x.q; r[1] = x.r; r[2] = x.s; r[3] = x.l; ++o; }
    }
    out->path_off[cs.paths.size()] = (int64_t)o;
    out->n_all = (int32_t)cs.all_scores.size();
    out->all_scores = (double*)malloc(sizeof(double) * (cs.all_scores.size() ? cs.all_scores.size() : 1));
    for (size_t i = 0; i < cs.all_scores.size(); ++i) out->all_scores[i] = cs.all_scores[i];
    return 0;
}

void vmo_test_asm_max_factor(double f) { vmo::g_asm_max_factor = f; }

void vmo_chains_free(vmo_chains* c) { free(c->path_off); free(c->path_anchors); free(c->all_scores); memset(c, 0, sizeof(*c)); }

}  // extern "C"
