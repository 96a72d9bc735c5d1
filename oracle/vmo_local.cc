// vmo_local.cc — CPU ORACLE (test infrastructure): local 9-mer re-seeding around guide chains + local chain DP.
//
// Restates the LIVE definitions in /root/reference/src/vacmap/mammap_clrnano.py:
//   get_localmap_multi_all_forDP_inv_guide_list (+ nested merge_chain, drop_somechains)   :28479-28589  (L1)
//   get_localmap_multi_all_forDP_inv_guide_1 (+ nested seq2hashtable_multi_test)          :23069-23345  (L2)
//   findClosest_1                                                                         :17560-17582
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list        ("LC-exact")   :27305-27528  (L3)
//   get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_mismatch ("LC-mm")    :28250-28476  (L4)
//   smallorequal2target_1d_point                                                          :13229-13265
// Mode L deltas: at most 3 guide chains (mammap_ccs.py:28581), LC-mm skipcost = min(skipcost, 40)
// (mammap_ccs.py:28587), maxgap 50 (mammap_ccs.py:24061).
//
// The reference keys its local tables by Python hash(str) of the 9-mer text, i.e. exact string identity; here the
// key is the 18-bit 2-bit-packed 9-mer. Documented deviation D1: a 9-mer holding a non-ACGT base never matches
// (the reference would match it against an identical N-bearing 9-mer; all-N 9-mers are skipped there too, :23074).
// Table entries of one 9-mer are visited in ascending reference position; duplicates caused by overlapping
// windows are no-ops in the reference (same read position, same diagonal -> bouns == 0), so they are dropped.
#include "vmo_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace vmo {

static const int64_t NOPRE = -9999999;

int pos2contig(const vmo_index* mi, int64_t gpos) {   // :51-59
    int n = index_nseq(mi);
    int pre = 0;
    for (int c = 0; c < n; ++c) { if (gpos < index_offset(mi, c)) break; pre = c; }
    return pre;
}

// :17560-17582
static void findClosest_1(const std::vector<int64_t>& arr, int64_t target, int64_t& b0, int64_t& b1, int64_t& i0, int64_t& i1) {
    int64_t n = (int64_t)arr.size();
    if (target <= arr[0]) { b0 = b1 = arr[0] - target; i0 = i1 = 0; return; }
    if (target >= arr[n - 1]) { b0 = b1 = target - arr[n - 1]; i0 = i1 = n - 1; return; }
    int64_t i = 0, j = n, mid = 0;
    while (i < j) {
        mid = (i + j) / 2;
        if (arr[mid] == target) { b0 = b1 = 0; i0 = i1 = mid; return; }
        if (target < arr[mid]) j = mid; else i = mid + 1;
    }
    b0 = std::llabs(arr[j - 1] - target); b1 = std::llabs(arr[j] - target); i0 = j - 1; i1 = j;
}

struct PointEntry { int64_t q, r, s, l; };

// L2 :23069-23345. `guide` = one guide chain (any order). Appends run-merged local anchors to out.
// r_st >= 0: collect_second_round_anchors (mammap_asm.py:22477-22756) — the same body, but the read positions looked up are [r_st, r_en - k)
// (:22580, :22589-22593) instead of the guide's span
void local_seed_one(const vmo_index* mi, const std::string& read, Path guide, int k, int64_t look_span,
                    int64_t read_span, std::vector<Anchor>& out, int64_t r_st, int64_t r_en) {
    const int64_t L = (int64_t)read.size();
    // :23095-23102
    int64_t readgap = 0;
    for (size_t i = 1; i < guide.size(); ++i) readgap = std::max<int64_t>(readgap, std::llabs(guide[i].q - guide[i - 1].q));
    readgap += 1000; readgap = std::max<int64_t>(readgap, 5000);
    std::stable_sort(guide.begin(), guide.end(), [](const Anchor& a, const Anchor& b) { return a.r < b.r; });   // :23103
    auto windows = [&](bool split_contig) {
        std::vector<std::pair<int64_t, int64_t>> se;
        se.emplace_back(guide[0].r, guide[0].r);
        int cur = pos2contig(mi, guide[0].r);
        for (size_t i = 1; i < guide.size(); ++i) {
            int64_t r = guide[i].r;
            bool same = (r - se.back().second) < readgap;
            if (split_contig) same = same && (cur == pos2contig(mi, r));
            if (same) se.back().second = r;
            else {
                if (se.back().first == se.back().second) se.pop_back();
                se.emplace_back(r, r);
                if (split_contig) cur = pos2contig(mi, r);
            }
        }
        if (se.back().first == se.back().second) se.pop_back();
        return se;
    };
    // build the union of padded windows as disjoint intervals of k-mer START positions (global coords)
    std::vector<std::pair<int64_t, int64_t>> ivs;  // [lo, hi) of k-mer starts
    auto build = [&](const std::vector<std::pair<int64_t, int64_t>>& se) -> bool {
        ivs.clear();
        for (auto& w : se) {
            int64_t min_ref = w.first, max_ref = w.second;
            int c = pos2contig(mi, min_ref);
            if (c != pos2contig(mi, max_ref)) return false;       // retry_diffcontig
            int64_t cst = index_offset(mi, c), clen = (int64_t)index_seq(mi, c).size();
            int64_t lookfurther = std::min<int64_t>(look_span, min_ref - cst);
            min_ref -= lookfurther; max_ref += look_span;
            int64_t lo = min_ref - cst, hi = std::min<int64_t>(max_ref - cst, clen);   // slice clips at contig end
            int64_t nk = (hi - lo) - k + 1;                                             // range(0, len(seq)-k+1)
            if (nk <= 0) continue;
            int64_t a = cst + lo, b = cst + lo + nk;
            if (!ivs.empty() && a < ivs.back().second) { if (b > ivs.back().second) ivs.back().second = b; }
            else ivs.emplace_back(a, b);
        }
        return true;
    };
    if (!build(windows(false))) build(windows(true));              // :23142-23180
    // 18-bit direct-address table (CSR), positions ascending
    const int nkey = 1 << (2 * k);
    std::vector<int32_t> cnt(nkey + 1, 0);
    std::vector<std::pair<int32_t, int64_t>> kp;
    for (auto& iv : ivs) {
        int c = pos2contig(mi, iv.first);
        const std::string& cs = index_seq(mi, c);
        int64_t cst = index_offset(mi, c);
        int64_t lo = iv.first - cst, hi = iv.second - cst;   // k-mer starts [lo,hi)
        uint32_t v = 0; int l = 0; const uint32_t mask = (uint32_t)nkey - 1;
        for (int64_t x = lo; x < hi + k - 1; ++x) {
            int cde = NT4[(uint8_t)cs[x]];
            if (cde < 4) { v = ((v << 2) | (uint32_t)cde) & mask; ++l; } else l = 0;
            if (l >= k) kp.emplace_back((int32_t)v, cst + x - k + 1);
        }
    }
    for (auto& e : kp) cnt[e.first + 1]++;
    for (int i = 0; i < nkey; ++i) cnt[i + 1] += cnt[i];
    std::vector<int64_t> tpos(kp.size());
    {
        std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
        for (auto& e : kp) tpos[fill[e.first]++] = e.second;   // kp is in ascending position order already
    }
    // :23183-23191
    std::stable_sort(guide.begin(), guide.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
    int64_t readstart = std::max<int64_t>(0, guide.front().q - read_span);
    int64_t readend = std::min<int64_t>(L - k + 1, guide.back().q + read_span);
    if (r_st >= 0) { readstart = r_st; readend = r_en - k; }
    std::vector<int64_t> readpos(guide.size());
    for (size_t i = 0; i < guide.size(); ++i) readpos[i] = guide[i].q;
    std::unordered_map<int64_t, PointEntry> pointdict;
    std::vector<int64_t> pointkeys;
    auto on_hit = [&](int64_t iloc, int64_t refloc, int64_t strand) {
        int64_t point = strand == 1 ? (refloc - iloc) : -(refloc + iloc);       // :23233 / :23292
        auto it = pointdict.find(point);
        if (it != pointdict.end()) {
            PointEntry c = it->second;
            if (c.q + c.l >= iloc) {
                int64_t bouns = iloc - (c.q + c.l) + k;
                if (bouns > 0) {
                    if (c.l + bouns < 20) {
                        if (strand == 1) it->second = PointEntry{c.q, c.r, 1, c.l + bouns};
                        else it->second = PointEntry{c.q, refloc, -1, c.l + bouns};
                    } else {
                        out.push_back(Anchor{c.q, c.r, c.s, c.l});
                        if (strand == 1) it->second = PointEntry{c.q + c.l, c.r + c.l, 1, bouns};
                        else it->second = PointEntry{c.q + c.l, refloc, -1, bouns};
                    }
                }
            } else {
                out.push_back(Anchor{c.q, c.r, c.s, c.l});
                it->second = PointEntry{iloc, refloc, strand, (int64_t)k};
            }
        } else {
            pointdict[point] = PointEntry{iloc, refloc, strand, (int64_t)k};
            pointkeys.push_back(point);
        }
    };
    const uint32_t mask = (uint32_t)nkey - 1;
    // rolling forward / reverse-complement 9-mers of the read
    uint32_t fw = 0, rv = 0; int l = 0;
    int64_t x0 = readstart;
    for (int64_t x = x0; x < readend + k - 1 && x < L; ++x) {
        int cde = NT4[(uint8_t)read[x]];
        if (cde < 4) { fw = ((fw << 2) | (uint32_t)cde) & mask; rv = (rv >> 2) | ((uint32_t)(3 - cde) << (2 * (k - 1))); ++l; }
        else l = 0;
        int64_t iloc = x - k + 1;
        if (iloc < readstart || iloc >= readend) continue;
        if (l < k) continue;                       // D1: non-ACGT 9-mer never matches
        if (fw == rv) continue;                    // :23213 (impossible for odd k)
        int64_t b0, b1, c0, c1;
        findClosest_1(readpos, iloc, b0, b1, c0, c1);
        int64_t interval = std::min<int64_t>(b0 + b1 + 500, 2000);
        int64_t ref1 = guide[c0].r, ref2 = guide[c1].r;
        int64_t rgap = std::llabs(iloc - guide[c0].q);
        auto accept = [&](int64_t refloc) {
            int64_t refgap = std::llabs(refloc - ref1);
            int64_t diff = std::llabs(rgap - refgap);
            return (diff < 500) || (ref1 + interval >= refloc && ref1 - interval <= refloc) ||
                   (ref2 + interval >= refloc && ref2 - interval <= refloc);
        };
        for (int32_t t = cnt[fw]; t < cnt[fw + 1]; ++t) if (accept(tpos[t])) on_hit(iloc, tpos[t], 1);
        if (iloc > 0)                              // rc_testseq[-(iloc+k):-iloc] is '' at iloc == 0 (:23212)
            for (int32_t t = cnt[rv]; t < cnt[rv + 1]; ++t) if (accept(tpos[t])) on_hit(iloc, tpos[t], -1);
    }
    for (int64_t key : pointkeys) { const PointEntry& c = pointdict[key]; out.push_back(Anchor{c.q, c.r, c.s, c.l}); }   // :23343
}

// :13229-13265 literal port
static int64_t smallorequal2target_1d_point(const double* arr, double target, int64_t n, const int64_t* point) {
    if (target < arr[point[0]]) return -1;
    if (target >= arr[point[n - 1]]) return n - 1;
    int64_t i = 0, j = n, mid = 0;
    while (i < j) {
        mid = (i + j) / 2;
        if (target == arr[point[mid]]) {
            if (mid < n - 1) { if (arr[point[mid + 1]] > target) return mid; else i = mid + 1; }
            else return mid;
        } else if (target < arr[point[mid]]) {
            if (mid > 0 && target >= arr[point[mid - 1]]) return mid - 1;
            j = mid;
        } else {
            if (mid < n - 1 && target < arr[point[mid + 1]]) return mid;
            i = mid + 1;
        }
    }
    return mid;
}

int local_chain_fast(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, bool mismatch,
                     int mode, double* score, Path& path);   // vmo_chain_fast.cc

// LC-exact :27305-27528 (mismatch=false) and LC-mm :28250-28476 (mismatch=true). A sorted by q+l (stable).
static int local_chain_dp(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, bool mismatch,
                          int mode, double* score, Path& path, std::vector<double>* S_out, std::vector<int64_t>* P_out) {
    const Tables& T = tables();
    const int64_t extra_size = (int64_t)T.extra.size() - 1;
    const int64_t log2cache_size = (int64_t)T.log2cache.size() - 1;
    const int64_t n = (int64_t)A.size();
    if (n == 0) return -1;   // np.array([]) indexing raises in the reference
    std::vector<double> gapcost_list(maxdiff + 1, 0.0);
    for (int g = 1; g <= maxdiff; ++g) {
        if (g <= 10) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
        else gapcost_list[g] = (0.01 * kmersize * g + 2 * T.log2int[g]);
    }
    const std::vector<float>& readgapcost = mismatch ? T.large_readgap : (mode == VMO_MODE_R ? T.readgap_r : T.readgap_h);
    std::vector<double> S(n); std::vector<int64_t> P(n), S_arg(n);
    int64_t opcount = 0;
    int64_t prereadloc = A[0].q + A[0].l;
    int64_t testspace_en = 1;
    S_arg[0] = 0; S[0] = (double)A[0].l; P[0] = NOPRE;
    double g_max_scores = (double)A[0].l; int64_t g_max_index = 0;
    for (int64_t i = 1; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        if (prereadloc < A[i].q + A[i].l) {
            if (opcount > 100000 && ((double)opcount / (double)prereadloc) > 1000.0)      // :27380
                return local_chain_fast(A, kmersize, skipcost, maxdiff, maxgap, mismatch, mode, score, path);
            for (int64_t k = testspace_en; k < i; ++k) {
                int64_t loc = smallorequal2target_1d_point(S.data(), S[k], k, S_arg.data()) + 1;
                memmove(S_arg.data() + loc + 1, S_arg.data() + loc, sizeof(int64_t) * (size_t)(k - loc));
                S_arg[loc] = k;
            }
            testspace_en = i;
            prereadloc = A[i].q + A[i].l;
        }
        const double li = (double)A[i].l;
        for (int64_t x = testspace_en - 1; x >= 0; --x) {
            const int64_t j = S_arg[x];
            ++opcount;
            if (S[j] < (max_scores - li)) break;
            const Anchor &ai = A[i], &aj = A[j];
            int64_t readgap = ai.q - aj.q - aj.l, refgap, bonus;
            if (readgap < 0) {
                bonus = ai.q + ai.l - aj.q - aj.l;
                if (bonus <= 0) continue;
                readgap = 0;
                int64_t overlap = aj.q + aj.l - ai.q;
                if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r + overlap - (aj.r + aj.l); else refgap = aj.r - (ai.r + bonus); }
                else { if (aj.s == -1) refgap = ai.r + overlap - aj.r + 1; else refgap = ai.r + bonus - 1 - (aj.r + aj.l); }
            } else {
                bonus = ai.l;
                if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r - aj.r - aj.l; else refgap = aj.r - ai.r - ai.l; }
                else { if (aj.s == -1) refgap = ai.r - aj.r + 1; else refgap = ai.r + ai.l - 1 - aj.r - aj.l; }
            }
            int64_t gapcost = std::llabs(readgap - refgap);
            double test;
            if (ai.s == aj.s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                test = S[j] + (double)bonus - gapcost_list[gapcost] - (double)readgapcost[readgap];
            } else if (!mismatch) {
                if (gapcost > extra_size) gapcost = extra_size;
                double pen;
                if (ai.s != aj.s) pen = std::min(50.0, skipcost) + (double)T.extra[gapcost];
                else pen = skipcost + (double)T.extra[gapcost];
                test = S[j] + (double)bonus - pen;
            } else {
                double pen = skipcost + T.log2cache[std::min<int64_t>(log2cache_size, gapcost)];
                test = S[j] + (double)bonus - pen;
            }
            if (test > max_scores) { max_scores = test; pre_index = j; }
        }
        S[i] = max_scores; P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    // traceback with overlap trimming :27508-27526
    path.clear();
    int64_t take = g_max_index;
    path.push_back(A[take]);
    Anchor preitem = A[take];
    while (true) {
        if (P[take] == NOPRE) break;
        take = P[take];
        const Anchor& now = A[take];
        if (preitem.q < now.q + now.l) {
            int64_t ov = now.q + now.l - preitem.q;
            if (preitem.s == 1) path.back() = Anchor{preitem.q + ov, preitem.r + ov, preitem.s, preitem.l - ov};
            else path.back() = Anchor{preitem.q + ov, preitem.r, preitem.s, preitem.l - ov};
        }
        path.push_back(now);
        preitem = now;
    }
    *score = g_max_scores;
    if (S_out) *S_out = S;
    if (P_out) *P_out = P;
    return 0;
}

// -mode asm: the older LC DP, mammap_asm.py:16540-16730. A sorted by read START (stable, :18237); the candidate window advances on read starts;
// `break` on S[j] < max - l is tested before anything else; no `bonus <= 0` skip; gap_geometry_asm; co-linear gap cost 0.5*log2 for every size;
// read-gap cost 0.1*log2(r) (:16536 = the mode-R table); a non-co-linear step costs skipcost + extra[gapcost]; no opcount switch. The traceback
// trims the EARLIER anchor's end where two chained anchors overlap on the read (:16720-16726).
int local_chain_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, double* score, Path& path) {
    const Tables& T = tables();
    const int64_t extra_size = (int64_t)T.extra.size() - 1;
    const int64_t n = (int64_t)A.size();
    if (n == 0) return -1;
    std::vector<double> gapcost_list(maxdiff + 1, 0.0);
    for (int g = 1; g <= maxdiff; ++g) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
    std::vector<double> S(n); std::vector<int64_t> P(n), S_arg(n);
    int64_t prereadloc = A[0].q;
    int64_t testspace_en = 1;
    S_arg[0] = 0; S[0] = (double)A[0].l; P[0] = NOPRE;
    double g_max_scores = (double)A[0].l; int64_t g_max_index = 0;
    for (int64_t i = 1; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        if (prereadloc < A[i].q) {
            for (int64_t k = testspace_en; k < i; ++k) {
                int64_t loc = smallorequal2target_1d_point(S.data(), S[k], k, S_arg.data()) + 1;
                memmove(S_arg.data() + loc + 1, S_arg.data() + loc, sizeof(int64_t) * (size_t)(k - loc));
                S_arg[loc] = k;
            }
            testspace_en = i;
            prereadloc = A[i].q;
        }
        const double li = (double)A[i].l;
        for (int64_t x = testspace_en - 1; x >= 0; --x) {
            const int64_t j = S_arg[x];
            if (S[j] < (max_scores - li)) break;
            int64_t readgap, refgap, bonus;
            gap_geometry_asm(A[i], A[j], readgap, refgap, bonus);
            int64_t gapcost = std::llabs(readgap - refgap);
            double test;
            if (A[i].s == A[j].s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                test = S[j] + (double)bonus - gapcost_list[gapcost] - (double)T.readgap_r[readgap];
            } else {
                if (gapcost > extra_size) gapcost = extra_size;
                test = S[j] - skipcost + (double)bonus - (double)T.extra[gapcost];
            }
            if (test > max_scores) { max_scores = test; pre_index = j; }
        }
        S[i] = max_scores; P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    path.clear();
    int64_t take = g_max_index;
    path.push_back(A[take]);
    Anchor preitem = A[take];
    while (P[take] != NOPRE) {
        take = P[take];
        const Anchor& now = A[take];
        if (preitem.q >= now.q + now.l) path.push_back(now);
        else if (now.s == 1) path.push_back(Anchor{now.q, now.r, now.s, preitem.q - now.q});
        else path.push_back(Anchor{now.q, now.r + now.l - preitem.q + now.q, now.s, preitem.q - now.q});
        preitem = now;
    }
    *score = g_max_scores;
    return 0;
}

// mode R: get_optimal_chain_sortbyreadpos_forSV_inv_test_merged_fine_list_scar (mammap_noprefercloser.py:23419-23628). A sorted by read
// START (stable); the candidate window still advances on read ENDS (:23484). Co-linear gap cost 0.5*log2 for every size (:23432), read-gap
// cost from the R table (:16534), a non-co-linear step costs the fixed skipcost, refunded after skipcost co-linear bases (fixed_penatly /
// pre_penatly). No opcount switch.
static int local_chain_scar(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, double* score, Path& path) {
    const Tables& T = tables();
    const int64_t n = (int64_t)A.size();
    if (n == 0) return -1;
    std::vector<double> gapcost_list(maxdiff + 1, 0.0);
    for (int g = 1; g <= maxdiff; ++g) gapcost_list[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
    const std::vector<float>& readgapcost = T.readgap_r;
    std::vector<double> S(n), pre_pen(n, 0.0), fixed_pen(n, 0.0); std::vector<int64_t> P(n), S_arg(n);
    int64_t prereadloc = A[0].q + A[0].l;
    int64_t testspace_en = 1;
    S_arg[0] = 0; S[0] = (double)A[0].l; P[0] = NOPRE;
    double g_max_scores = (double)A[0].l; int64_t g_max_index = 0;
    for (int64_t i = 1; i < n; ++i) {
        double max_scores = (double)A[i].l;
        int64_t pre_index = NOPRE;
        if (prereadloc < A[i].q + A[i].l) {
            for (int64_t k = testspace_en; k < i; ++k) {
                int64_t loc = smallorequal2target_1d_point(S.data(), S[k], k, S_arg.data()) + 1;
                memmove(S_arg.data() + loc + 1, S_arg.data() + loc, sizeof(int64_t) * (size_t)(k - loc));
                S_arg[loc] = k;
            }
            testspace_en = i;
            prereadloc = A[i].q + A[i].l;
        }
        const double li = (double)A[i].l;
        for (int64_t x = testspace_en - 1; x >= 0; --x) {
            const int64_t j = S_arg[x];
            if (S[j] < (max_scores - li)) break;
            const Anchor &ai = A[i], &aj = A[j];
            int64_t readgap = ai.q - aj.q - aj.l, refgap, bonus;
            if (readgap < 0) {
                bonus = ai.q + ai.l - aj.q - aj.l;
                if (bonus <= 0) continue;
                readgap = 0;
                int64_t overlap = aj.q + aj.l - ai.q;
                if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r + overlap - (aj.r + aj.l); else refgap = aj.r - (ai.r + bonus); }
                else { if (aj.s == -1) refgap = ai.r + overlap - aj.r + 1; else refgap = ai.r + bonus - 1 - (aj.r + aj.l); }
            } else {
                bonus = ai.l;
                if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r - aj.r - aj.l; else refgap = aj.r - ai.r - ai.l; }
                else { if (aj.s == -1) refgap = ai.r - aj.r + 1; else refgap = ai.r + ai.l - 1 - aj.r - aj.l; }
            }
            const int64_t gapcost = std::llabs(readgap - refgap);
            if (ai.s == aj.s && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                double test = S[j] + (double)bonus - gapcost_list[gapcost] - (double)readgapcost[readgap];
                if (fixed_pen[j] < 0 && (fixed_pen[j] + (double)bonus) >= 0) test += pre_pen[j];
                if (test > max_scores) {
                    max_scores = test; pre_index = j;
                    if (fixed_pen[j] < 0 && (fixed_pen[j] + (double)bonus) < 0) { fixed_pen[i] = fixed_pen[j] + (double)bonus; pre_pen[i] = pre_pen[j]; }
                    else { fixed_pen[i] = 0; pre_pen[i] = 0; }
                }
            } else {
                const double tmp_penalty = skipcost;
                const double test = S[j] + (double)bonus - tmp_penalty;
                if (test > max_scores) { max_scores = test; pre_index = j; fixed_pen[i] = -tmp_penalty + (double)bonus; pre_pen[i] = tmp_penalty; }
            }
        }
        S[i] = max_scores; P[i] = pre_index;
        if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
    }
    path.clear();
    int64_t take = g_max_index;
    path.push_back(A[take]);
    Anchor preitem = A[take];
    while (P[take] != NOPRE) {
        take = P[take];
        const Anchor& now = A[take];
        if (preitem.q < now.q + now.l) {
            int64_t ov = now.q + now.l - preitem.q;
            if (preitem.s == 1) path.back() = Anchor{preitem.q + ov, preitem.r + ov, preitem.s, preitem.l - ov};
            else path.back() = Anchor{preitem.q + ov, preitem.r, preitem.s, preitem.l - ov};
        }
        path.push_back(now);
        preitem = now;
    }
    *score = g_max_scores;
    return 0;
}

// L1 :28479-28589
int local_chain(const vmo_index* mi, const std::string& read, const std::string& rc, const std::vector<Path>& guides_in,
                const vmo_params& prm, double* score, Path& chain_desc, std::vector<Anchor>* raw_out, int* variant) {
    (void)rc;
    const int mode = prm.mode;
    const int k = prm.local_kmersize;
    const int maxgap = (mode == VMO_MODE_L) ? 50 : 99;           // :24061 / mammap_ccs.py:24061
    const int max_chains = (mode == VMO_MODE_L) ? 3 : 5;         // :28581 / mammap_ccs.py:28581 (S: unlimited -> see below)
    if (guides_in.empty()) return -1;
    if (mode == VMO_MODE_ASM) {
        // mammap_asm.py:19714-19719 -> get_localmap_multi_all_forDP_inv_guide :17960-18238: the primary path only, reference window +-2000, read
        // window +-500, anchors sorted by read start (:18237), the older LC DP
        std::vector<Anchor> raw;
        local_seed_one(mi, read, guides_in[0], k, 2000, 500, raw);
        if (raw_out) *raw_out = raw;
        if (raw.empty()) return -16;       // np.array([])[:, 0] raises IndexError (:18237)
        std::stable_sort(raw.begin(), raw.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
        if (variant) *variant = 3;
        return local_chain_asm(raw, k, prm.local_skipcost, prm.local_maxdiff, 99, score, chain_desc);
    }
    if (mode == VMO_MODE_R) {
        // mammap_noprefercloser.py:23902-23914: every chain is re-seeded in the order given (no merge / drop / cap), reference window
        // +-2000, read window +-500 (:23631+); anchors sorted by read start; one DP variant
        std::vector<Anchor> raw;
        for (const Path& g : guides_in) local_seed_one(mi, read, g, k, 2000, 500, raw);
        if (raw_out) *raw_out = raw;
        std::stable_sort(raw.begin(), raw.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
        if (variant) *variant = 2;
        return local_chain_scar(raw, k, prm.local_skipcost, prm.local_maxdiff, maxgap, score, chain_desc);
    }
    // merge_chain :28529-28569
    std::vector<Path> chains(guides_in.begin() + 1, guides_in.end());
    std::stable_sort(chains.begin(), chains.end(), [](const Path& a, const Path& b) { return a.back().q < b.back().q; });
    for (size_t iloc = 0; iloc + 1 < chains.size(); ++iloc) {
        size_t jloc = iloc + 1;
        while (jloc < chains.size()) {
            const Anchor& ie = chains[iloc].front();   // chain end (highest q)
            const Anchor& js = chains[jloc].back();    // chain start (lowest q)
            if (ie.q + ie.l <= js.q && ie.s == js.s) {
                int64_t readgap = js.q - ie.q - ie.l, refgap;
                if (ie.s == 1) refgap = js.r - ie.r - ie.l; else refgap = ie.r - js.r - js.l;
                if (std::llabs(readgap - refgap) < 500) {
                    Path merged = chains[jloc];
                    merged.insert(merged.end(), chains[iloc].begin(), chains[iloc].end());
                    chains[iloc] = merged;
                    chains.erase(chains.begin() + jloc);
                    continue;
                }
            }
            ++jloc;
        }
    }
    std::stable_sort(chains.begin(), chains.end(), [](const Path& a, const Path& b) { return a.size() < b.size(); });
    std::vector<Path> lst; lst.push_back(guides_in[0]);
    for (auto& c : chains) lst.push_back(c);
    // drop_somechains :28482-28528
    {
        size_t ns = lst.size() - 1;
        std::vector<int64_t> iloclist(ns, 0), distance(ns, INT64_MAX);
        std::vector<double> sc0(ns, 0), sc1(ns, 0), cc0(ns, 0), cc1(ns, 0);
        for (const Anchor& item : lst[0]) {
            for (size_t c = 0; c < ns; ++c) {
                const Path& chain = lst[c + 1];
                if (item.q >= chain.back().q && item.q <= chain.front().q) { if (item.s == 1) sc0[c] += 1; else sc1[c] += 1; }
                while (chain[iloclist[c]].q > item.q) { if (iloclist[c] < (int64_t)chain.size() - 1) iloclist[c]++; else break; }
                const Anchor& t = chain[iloclist[c]];
                if (std::llabs(item.r - t.r) < distance[c]) distance[c] = std::llabs(item.r - t.r);
            }
        }
        for (size_t c = 0; c < ns; ++c) for (const Anchor& item : lst[c + 1]) { if (item.s == 1) cc0[c] += 1; else cc1[c] += 1; }
        std::vector<Path> kept; kept.push_back(lst[0]);
        for (size_t c = 0; c < ns; ++c) {
            bool keep;
            if (sc0[c] > sc1[c] && cc0[c] > cc1[c]) keep = true;
            else if (sc0[c] < sc1[c] && cc0[c] < cc1[c]) keep = true;
            else keep = false;
            const Path& ch = lst[c + 1];
            if ((!keep && distance[c] < 500) || (ch.front().q - ch.back().q) < 100) continue;
            kept.push_back(ch);
        }
        lst.swap(kept);
    }
    // sort by 1/len ascending (= length descending), stable :28574
    std::stable_sort(lst.begin(), lst.end(), [](const Path& a, const Path& b) { return a.size() > b.size(); });
    std::vector<Anchor> raw;
    int used = 0;
    for (const Path& g : lst) {
        if (used >= max_chains && mode != VMO_MODE_S) break;
        local_seed_one(mi, read, g, k, 7000, 7000, raw);
        ++used;
    }
    if (raw_out) *raw_out = raw;
    std::stable_sort(raw.begin(), raw.end(), [](const Anchor& a, const Anchor& b) { return a.q + a.l < b.q + b.l; });   // :28585
    if (lst.size() > 1) {
        double sk = prm.local_skipcost;
        if (mode == VMO_MODE_L) sk = std::min(sk, 40.0);             // mammap_ccs.py:28587
        if (variant) *variant = 1;
        return local_chain_dp(raw, k, sk, prm.local_maxdiff, maxgap, true, mode, score, chain_desc, nullptr, nullptr);
    }
    if (variant) *variant = 0;
    return local_chain_dp(raw, k, prm.local_skipcost, prm.local_maxdiff, maxgap, false, mode, score, chain_desc, nullptr, nullptr);
}

}  // namespace vmo

using namespace vmo;

extern "C" int vmo_local_chain(const vmo_index* mi, const char* read, int64_t readlen, int n_paths, const int64_t* path_off,
                               const int64_t* pa, const vmo_params* p, double* score, int64_t** chain, int64_t* n_chain,
                               int64_t** raw, int64_t* n_raw, int32_t* variant) {
    std::vector<Path> guides(n_paths);
    for (int i = 0; i < n_paths; ++i)
        for (int64_t x = path_off[i]; x < path_off[i + 1]; ++x) guides[i].push_back(Anchor{pa[4 * x], pa[4 * x + 1], pa[4 * x + 2], pa[4 * x + 3]});
    std::string rd(read, (size_t)readlen);
    for (char& c : rd) if (c >= 'a' && c <= 'z') c -= 32;
    Path ch; std::vector<Anchor> rw; int var = 0; double sc = 0;
    int rc = local_chain(mi, rd, revcomp(rd), guides, *p, &sc, ch, &rw, &var);
    auto dump = [](const std::vector<Anchor>& v, int64_t** o, int64_t* n) {
        *o = (int64_t*)malloc(sizeof(int64_t) * 4 * (v.size() ? v.size() : 1));
        for (size_t i = 0; i < v.size(); ++i) { (*o)[4 * i] = v[i].q; (*o)[4 * i + 1] = v[i].r; (*o)[4 * i + 2] = v[i].s; (*o)[4 * i + 3] = v[i].l; }
        *n = (int64_t)v.size();
    };
    dump(ch, chain, n_chain); dump(rw, raw, n_raw);
    if (score) *score = sc;
    if (variant) *variant = var;
    return rc;
}
