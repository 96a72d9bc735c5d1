// vmo_extend.cc — CPU ORACLE (test infrastructure): segment surgery, divergence filter, edge extension, gap-fill,
// record assembly, and the per-read driver.
//
// Restates the LIVE definitions in /root/reference/src/vacmap/mammap_clrnano.py:
//   extend_func                      :19238-19303        rebuild_chain_break            :23437-23484  (E1)
//   get_query_target_for_cigar       :5802-5818          extend_edge_test               :2302-2525    (E3)
//   drop_misplaced_alignment_test    :726-787            merge_conjacent_alignment      :16736-16780  (E4)
//   getdupiloc_numba                 :16680-16734        List_merge                     :283-288
//   fix_simple_inv                   :24226-24312        split_alignment_test           :21505-21617  (E5)
//   get_onemapinfolist               :20731-20838 (E6)   pairedindel                    :5604-5650
//   get_readmap_DP_test              :24023-24084        (per-read driver)
// A Python exception inside the per-read path makes the reference skip the read (:24116-24125); here every such
// site returns a negative status instead (documented next to each `return -...`).
#include "vmo_internal.h"
#include <malloc.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

namespace vmo {

thread_local std::vector<DpCall>* g_dplog = nullptr;
static std::atomic<int64_t> g_surgery[4];     // tests: drop_misplaced removals, merges, fix_simple_inv shifts (left / right branch)

typedef std::vector<Anchor> Seg;

// Python s[a:b] (step 1) including negative-index wrap-around
static std::string pyslice(const std::string& s, int64_t a, int64_t b) {
    int64_t n = (int64_t)s.size();
    if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
    if (b <= a) return std::string();
    return s.substr((size_t)a, (size_t)(b - a));
}
static std::string reversed(std::string s) { std::reverse(s.begin(), s.end()); return s; }

static bool same_contig(const vmo_index* mi, int64_t a, int64_t b) { return pos2contig(mi, a) == pos2contig(mi, b); }

// E1 :23437-23484. returns <0 where the reference raises IndexError (alignment_list[-1] on an empty list)
// asmv: the -mode asm fork (mammap_asm.py:13256-13295) joins only when refgap >= 0 (no tolerance for a 20-base back-step)
static int rebuild_chain_break(const vmo_index* mi, const Path& raw, int64_t large_cost, int64_t small_alignment, std::vector<Seg>& al, bool asmv = false) {
    al.clear();
    Anchor pre = raw[0];
    al.push_back(Seg{pre});
    for (size_t x = 1; x < raw.size(); ++x) {
        const Anchor& now = raw[x];
        if (pre.s == now.s) {
            int64_t readgap = now.q - pre.q - pre.l, refgap;
            if (pre.s == 1) refgap = now.r - pre.r - pre.l; else refgap = pre.r - now.r - now.l;
            if (asmv) {
                if (std::llabs(readgap - refgap) <= large_cost && refgap >= 0 && readgap < 100 && same_contig(mi, pre.r, now.r)) { al.back().push_back(now); pre = now; continue; }
            } else if (std::llabs(readgap - refgap) <= large_cost && refgap >= -20 && readgap < 100) {
                if (same_contig(mi, pre.r, now.r)) {
                    if (refgap >= 0) { al.back().push_back(now); pre = now; continue; }
                    else { if (readgap <= 20) continue; al.back().push_back(now); pre = now; continue; }
                }
            }
        }
        if (al.back().size() == 1) al.pop_back();
        if (!al.empty()) {
            const Seg& s = al.back();
            if ((s.back().q + s.back().l - s.front().q) < small_alignment) al.pop_back();
        }
        al.push_back(Seg{now});
        pre = now;
    }
    if (al.back().size() == 1) al.pop_back();
    if (al.empty()) return -11;   // IndexError at :23480
    {
        const Seg& s = al.back();
        if ((s.back().q + s.back().l - s.front().q) < small_alignment) al.pop_back();
    }
    return 0;
}

// :5802-5818
void get_query_target_for_cigar(const vmo_index* mi, const Anchor& pre, const Anchor& now, const std::string& read,
                                       const std::string& rc, int64_t L, std::string& target, std::string& query) {
    if (pre.s == 1) {
        int c = pos2contig(mi, pre.r); int64_t bias = index_offset(mi, c);
        query = pyslice(read, pre.q, now.q);
        target = pyslice(index_seq(mi, c), pre.r - bias, now.r - bias);
    } else {
        int c = pos2contig(mi, now.r); int64_t bias = index_offset(mi, c);
        query = pyslice(rc, L - now.q, L - pre.q);
        target = pyslice(index_seq(mi, c), now.r + now.l - bias, pre.r + pre.l - bias);
    }
}

static void ext_call(const std::string& target, const std::string& query, int32_t& t_e, int32_t& q_e) {
    if (g_dplog) g_dplog->push_back(DpCall{1, target, query});
    // mp.k_cigar(target, query, match=2, mismatch=-4, 4,4,4,4, bw=100, zdropvalue=50)  (:2381)
    k_extend(target.data(), (int64_t)target.size(), query.data(), (int64_t)query.size(), 2, -4, 4, 4, 100, 50, &t_e, &q_e);
}

// E3 :2302-2525 (san = 1)
static void extend_edge_test(const vmo_index* mi, const std::string& read, int64_t L, std::vector<Seg>& al) {
    const int64_t max_extend_size = 20000;
    const int64_t san = 1;
    for (int64_t idx = 0; idx < (int64_t)al.size(); ++idx) {
        Seg& one = al[idx];
        if (one[0].q > 0) {
            int64_t looksize;
            int64_t pre_idx = std::max<int64_t>(idx - san, 0);
            if (idx == 0 || idx - san < 0) looksize = one[0].q - 0;
            else looksize = one[0].q - (al[pre_idx].back().q + al[pre_idx].back().l);
            const Anchor pre = one[0];
            int c = pos2contig(mi, pre.r);
            int64_t cst = index_offset(mi, c);
            const std::string& cs = index_seq(mi, c);
            if (pre.s == 1) {
                int64_t target_st = pre.r, query_st = pre.q;
                looksize = std::min(looksize, target_st - cst);
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize != 0) {
                    std::string query = reversed(pyslice(read, std::max<int64_t>(query_st - looksize, 0), query_st));
                    std::string target = reversed(pyslice(cs, target_st - cst - (int64_t)query.size(), target_st - cst));
                    int32_t t_e, q_e; ext_call(target, query, t_e, q_e);
                    one[0] = Anchor{query_st - q_e, target_st - t_e, 1, 0};
                }
            } else {
                int64_t target_en = pre.r + pre.l, query_st = pre.q;
                looksize = std::min(looksize, cst + (int64_t)cs.size() - (target_en - 1));
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize != 0) {
                    std::string query = reversed(pyslice(read, std::max<int64_t>(query_st - looksize, 0), query_st));
                    std::string target = reversed(revcomp(pyslice(cs, target_en - cst, target_en + (int64_t)query.size() - cst)));
                    int32_t t_e, q_e; ext_call(target, query, t_e, q_e);
                    one[0] = Anchor{query_st - q_e, target_en + t_e, -1, 0};
                }
            }
        } else {
            Anchor t = one[0];
            if (t.s == 1) one[0] = Anchor{t.q, t.r, 1, 0}; else one[0] = Anchor{t.q, t.r + t.l, -1, 0};
        }
        if ((one.back().q + one.back().l) < L) {
            int64_t looksize;
            int64_t nxt = std::min<int64_t>(idx + san, (int64_t)al.size());
            if (nxt == (int64_t)al.size()) looksize = L - (one.back().q + one.back().l);
            else looksize = al[nxt][0].q - (one.back().q + one.back().l);
            const Anchor pre = one[one.size() - 2];
            const Anchor now = one.back();
            int c = pos2contig(mi, pre.r);
            int64_t cst = index_offset(mi, c);
            const std::string& cs = index_seq(mi, c);
            if (pre.s == 1) {
                int64_t target_en = now.r + now.l, query_en = now.q + now.l;
                looksize = std::min(looksize, cst + (int64_t)cs.size() - (target_en - 1));
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize != 0) {
                    std::string query = pyslice(read, query_en, query_en + looksize);
                    std::string target = pyslice(cs, target_en - cst, target_en + (int64_t)query.size() - cst);
                    int32_t t_e, q_e; ext_call(target, query, t_e, q_e);
                    one.back() = Anchor{query_en + q_e, target_en + t_e, 1, 0};
                }
            } else {
                int64_t target_st = now.r, query_en = now.q + now.l;
                looksize = std::min(looksize, target_st - cst);
                if (looksize > max_extend_size) looksize = max_extend_size;
                if (looksize != 0) {
                    std::string query = pyslice(read, query_en, query_en + looksize);
                    std::string target = revcomp(pyslice(cs, target_st - cst - (int64_t)query.size(), target_st - cst));
                    int32_t t_e, q_e; ext_call(target, query, t_e, q_e);
                    one.back() = Anchor{query_en + q_e, target_st - t_e, -1, 0};
                }
            }
        } else {
            Anchor t = one.back();
            if (t.s == 1) one.back() = Anchor{t.q + t.l, t.r + t.l, 1, 0}; else one.back() = Anchor{t.q + t.l, t.r, -1, 0};
        }
    }
}

// :726-787
static bool drop_misplaced_alignment_test(std::vector<Seg>& al, int64_t iloc) {
    if (al[iloc][0].s == al[iloc + 1][0].s && al[iloc][0].s == al[iloc + 2][0].s) {
        int64_t mid_size = al[iloc + 1].back().q + al[iloc + 1].back().l - al[iloc + 1][0].q;
        if (mid_size > 1000) return false;
        Anchor pre = al[iloc].back(), now = al[iloc + 1][0];
        int64_t readgap = now.q - pre.q - pre.l, refgap;
        if (pre.s == 1) refgap = now.r - pre.r - pre.l; else refgap = pre.r - now.r - now.l;
        if (std::llabs(refgap) < 100000) {
            int DEL = 0, INS = 0;
            if ((readgap - refgap) < -30) DEL += 1; else if ((readgap - refgap) > 30) INS += 1; else return false;
            int64_t gap_1 = std::llabs(readgap - refgap);
            pre = al[iloc + 1].back(); now = al[iloc + 2][0];
            readgap = now.q - pre.q - pre.l;
            if (pre.s == 1) refgap = now.r - pre.r - pre.l; else refgap = pre.r - now.r - now.l;
            if (std::llabs(refgap) < 100000) {
                if ((readgap - refgap) < -30) DEL += 1; else if ((readgap - refgap) > 30) INS += 1; else return false;
                int64_t gap_2 = std::llabs(readgap - refgap);
                if (DEL == 1 && INS == 1 && (mid_size < 500 || ((double)std::max(gap_1, gap_2) / (double)mid_size) > 0.5)) {
                    al.erase(al.begin() + iloc + 1);
                    g_surgery[0].fetch_add(1);
                    return true;
                }
            }
        }
    }
    return false;
}

// :16680-16734 (Q7: adds the strand field at :16705, preserved)
static std::vector<int64_t> getdupiloc(const std::vector<Seg>& al) {
    std::vector<int64_t> dup;
    if (al.size() >= 2) {
        int64_t iloc = 0;
        while ((iloc + 1) < (int64_t)al.size()) {
            int64_t readpos_1 = al[iloc].back().q + al[iloc].back().l;
            int64_t refpos_1; int strand_1;
            if (al[iloc].back().s == 1) { refpos_1 = al[iloc].back().r + al[iloc].back().l; strand_1 = 1; }
            else { refpos_1 = al[iloc].back().r; strand_1 = -1; }
            int64_t jloc = iloc; bool hit = false; int64_t dupsize = 0, new_iloc = 0, readpos_2 = 0;
            while ((jloc + 1) < (int64_t)al.size()) {
                jloc += 1;
                int64_t refpos_2; int strand_2;
                if (al[jloc].back().s == 1) { refpos_2 = al[jloc][0].r; strand_2 = 1; }
                else { refpos_2 = al[jloc][0].r + al[jloc][0].s; strand_2 = -1; }
                if (strand_1 != strand_2) continue;
                if (strand_1 == 1) {
                    if ((refpos_2 - refpos_1) < 50) { new_iloc = jloc; dupsize = refpos_2 - refpos_1; readpos_2 = al[jloc][0].q; hit = true; }
                } else {
                    if ((refpos_1 - refpos_2) < 50) { new_iloc = jloc; dupsize = refpos_1 - refpos_2; readpos_2 = al[jloc][0].q; hit = true; }
                }
            }
            if (hit) {
                int64_t readgap = readpos_2 - readpos_1;
                if (((iloc + 1) < new_iloc) || (((dupsize - readgap) < -30) && (readgap < 30)))
                    for (int64_t s = iloc; s < new_iloc; ++s) dup.push_back(s);
                iloc = new_iloc;
            } else iloc += 1;
        }
    }
    return dup;
}

// :16736-16780
static void merge_conjacent_alignment(const vmo_index* mi, std::vector<Seg>& al) {
    if (al.size() >= 2) {
        int64_t iloc = 0;
        std::vector<int64_t> duplist = getdupiloc(al);
        while ((iloc + 1) < (int64_t)al.size()) {
            if (std::find(duplist.begin(), duplist.end(), iloc) != duplist.end()) { iloc += 1; continue; }
            const Anchor pre = al[iloc].back(), now = al[iloc + 1][0];
            if (pre.s != now.s || !same_contig(mi, pre.r, now.r)) { iloc += 1; continue; }
            int64_t readgap = now.q - pre.q - pre.l, refgap;
            if (pre.s == 1) refgap = now.r - pre.r - pre.l; else refgap = pre.r - now.r - now.l;
            if (refgap < 0) { iloc += 1; continue; }
            if (std::min(readgap, refgap) < 50 && std::llabs(readgap - refgap) < 10000) {
                g_surgery[1].fetch_add(1);
                al[iloc].insert(al[iloc].end(), al[iloc + 1].begin(), al[iloc + 1].end());   // List_merge :283
                al.erase(al.begin() + iloc + 1);
            } else iloc += 1;
        }
    }
}

// :24226-24312. returns <0 where the reference raises (assert / IndexError on an emptied segment)
// rmode: mode R keeps an older body (mammap_noprefercloser.py:17155-17200) whose `refen_0 > refst_1` branch compares nothing and changes nothing
static int fix_simple_inv(const vmo_index* mi, std::vector<Seg>& al, const std::string& read, bool rmode) {
    if (al.size() > 2) {
        int64_t iloc = 0;
        while (iloc + 2 < (int64_t)al.size()) {
            if (al[iloc][0].s == al[iloc + 2][0].s && al[iloc][0].s != al[iloc + 1][0].s) {
                if (al[iloc][0].s == 1) {
                    int c = pos2contig(mi, al[iloc][0].r);
                    int64_t bias = index_offset(mi, c);
                    const std::string& cs = index_seq(mi, c);
                    int64_t refen_0 = al[iloc].back().r + al[iloc].back().l - bias;
                    int64_t readen_0 = al[iloc].back().q + al[iloc].back().l;
                    int64_t refst_1 = al[iloc + 1].back().r - bias;
                    int64_t readst_1 = al[iloc + 1][0].q;
                    int64_t refen_1 = al[iloc + 1][0].r + al[iloc + 1][0].l - bias;
                    int64_t readen_1 = al[iloc + 1].back().q + al[iloc + 1].back().l;
                    int64_t refst_2 = al[iloc + 2][0].r - bias;
                    int64_t readst_2 = al[iloc + 2][0].q;
                    if (refst_2 - refen_0 == refen_1 - refst_1 && readst_1 - readen_0 + readst_2 - readen_1 == 0) {
                        if (refst_1 - refen_0 != 0 && refst_1 - refen_0 + refst_2 - refen_1 == 0) {
                            if (refen_0 > refst_1) {
                                if (rmode) { iloc += 1; continue; }
                                std::string tempref = revcomp(pyslice(cs, refen_1, refen_1 + refen_0 - refst_1));
                                std::string tempquery = pyslice(read, readen_0 - refen_0 + refst_1, readen_0);
                                if (tempref == tempquery) {
                                    g_surgery[2].fetch_add(1);
                                    int64_t b = refen_0 - refst_1;
                                    al[iloc + 2][0] = Anchor{readst_2 - b, refst_2 - b + bias, 1, 0};
                                    Anchor ins{readst_2 - b, refen_0 + bias, -1, 0};
                                    while (true) {
                                        if (al[iloc + 1].empty()) return -12;   // IndexError
                                        if (ins.q <= (al[iloc + 1].back().q + al[iloc + 1].back().l)) al[iloc + 1].pop_back(); else break;
                                    }
                                    al[iloc + 1].push_back(ins);
                                }
                            } else {
                                std::string tempref = pyslice(cs, refen_0, refen_0 - refen_0 + refst_1);
                                std::string tempquery = pyslice(read, readen_0, readen_0 - refen_0 + refst_1);
                                if (tempref == tempquery) {
                                    g_surgery[3].fetch_add(1);
                                    al[iloc].back() = Anchor{readen_0 - refen_0 + refst_1, refen_0 - refen_0 + refst_1 + bias, 1, 0};
                                    Anchor ins{readen_0 - refen_0 + refst_1, refen_1 + refen_0 - refst_1 + bias, -1, 0};
                                    while (true) {
                                        if (al[iloc + 1].empty()) return -12;
                                        if (ins.q >= al[iloc + 1][0].q) al[iloc + 1].erase(al[iloc + 1].begin()); else break;
                                    }
                                    al[iloc + 1].insert(al[iloc + 1].begin(), ins);
                                }
                            }
                        }
                    }
                }
            }
            iloc += 1;
        }
    }
    return 0;
}

static int gap_fill(const std::string& target, const std::string& query, int eqx, std::string& cigar) {
    if (g_dplog) g_dplog->push_back(DpCall{0, target, query});
    // mp.k_cigar(target, query, 2, -4, 4, 2, 24, 1, bw=-1, zdropvalue=-1, eqx)   (:21554)
    return k_cigar_global(target.data(), (int64_t)target.size(), query.data(), (int64_t)query.size(), 2, -4, 4, 2, 24, 1, eqx, cigar, nullptr);
}

// E5 :21505-21617. out_alignment = new_alignment[0], cigars = cigarlist[0]
// link_cigar :22365-22410 (mode asm): joins two CIGAR strings, adding the counts when the last operator of the first equals the first of the second
static std::string link_cigar(const std::string& c1, const std::string& c2) {
    int64_t iloc = (int64_t)c1.size() - 1;
    const char last = c1[iloc];
    int64_t num_1 = 0, factor = 1;
    while (iloc > 0) {
        iloc -= 1;
        if (c1[iloc] >= '0' && c1[iloc] <= '9') { num_1 = num_1 + (c1[iloc] - '0') * factor; factor *= 10; continue; }
        else break;
    }
    char first = 0; int64_t jloc = 0, num_2 = 0;
    while (jloc < (int64_t)c2.size()) {
        first = c2[jloc];
        if (c2[jloc] >= '0' && c2[jloc] <= '9') { num_2 = num_2 * 10 + (c2[jloc] - '0'); jloc += 1; continue; }
        else break;
    }
    if (last == first) {
        const std::string mid = std::to_string(num_1 + num_2) + last;
        const bool whole1 = iloc == 0, whole2 = (jloc + 1) == (int64_t)c2.size();
        if (whole1 && whole2) return mid;
        if (whole1) return mid + c2.substr((size_t)jloc + 1);
        if (whole2) return c1.substr(0, (size_t)iloc + 1) + mid;
        return c1.substr(0, (size_t)iloc + 1) + mid + c2.substr((size_t)jloc + 1);
    }
    return c1 + c2;
}

// asmv (mammap_asm.py:22197-22316): the short-anchor / short-gap skip applies only while max(readgap, refgap) < 2000; an empty CIGAR raises; the
// CIGAR pieces of a segment are joined with link_cigar into ONE string
static int split_alignment_test(const vmo_index* mi, Seg alignment, const std::string& read, const std::string& rc, int64_t L,
                                int eqx, Seg& out_alignment, std::vector<std::string>& cigars, bool asmv = false) {
    const int64_t min_gap_forcigar = 200;
    auto add_cigar = [&](const std::string& cg) -> int {
        if (!asmv) { cigars.push_back(cg); return 0; }
        if (cg.empty()) return -13;                                   // :22244
        if (!cigars.empty()) cigars[0] = link_cigar(cigars[0], cg); else cigars.push_back(cg);
        return 0;
    };
    out_alignment.clear(); cigars.clear();
    if (alignment[0].s == 1) {
        Anchor& last = alignment.back();
        if (last.l != 0) last = Anchor{last.q + last.l, last.r + last.l, 1, 0};
        Anchor pre = alignment[0];
        out_alignment.push_back(pre);
        size_t iloc = 1;
        while (iloc < alignment.size()) {
            const Anchor now = alignment[iloc];
            int64_t readgap = now.q - pre.q - pre.l, refgap = now.r - pre.r - pre.l;
            if ((!asmv || std::max(readgap, refgap) < 2000) && (now.l < 19 || std::min(readgap, refgap) < min_gap_forcigar)) {
                if (iloc + 1 != alignment.size()) { iloc += 1; continue; }
            }
            std::string target, query;
            get_query_target_for_cigar(mi, pre, now, read, rc, L, target, query);
            if (target.size() > 0 && query.size() > 0) {
                std::string cg; gap_fill(target, query, eqx, cg);
                out_alignment.push_back(now); if (add_cigar(cg) < 0) return -13;
            } else return -13;   // raise Exception("ERROR: Failed to compute CIGAR") :21562
            pre = now; iloc += 1;
        }
        if (cigars.empty()) return -13;   // :21566
        return 0;
    } else {
        if (alignment[0].l != 0) alignment[0] = Anchor{alignment[0].q, alignment[0].r + alignment[0].l, -1, 0};
        if (alignment.back().l != 0) { Anchor& b = alignment.back(); b = Anchor{b.q + b.l, b.r, -1, 0}; }
        std::reverse(alignment.begin(), alignment.end());
        Anchor pre = alignment[0];
        out_alignment.push_back(pre);
        size_t iloc = 1;
        while (iloc < alignment.size()) {
            const Anchor now = alignment[iloc];
            int64_t readgap = pre.q - now.q - now.l, refgap = now.r - pre.r - pre.l;
            if ((!asmv || std::max(readgap, refgap) < 2000) && (now.l < 19 || std::min(readgap, refgap) < min_gap_forcigar)) {
                if (iloc + 1 != alignment.size()) { iloc += 1; continue; }
            }
            std::string target, query;
            get_query_target_for_cigar(mi, now, pre, read, rc, L, target, query);
            if (target.size() > 0 && query.size() > 0) {
                std::string cg; gap_fill(target, query, eqx, cg);
                out_alignment.push_back(now); if (add_cigar(cg) < 0) return -13;
            } else return -13;
            pre = now; iloc += 1;
        }
        if (cigars.empty()) return -13;
        return 0;
    }
}

static int64_t cigar_qlen(const std::string& c) {   // len(Cigar(s)): M, I, S, =, X
    int64_t n = 0, tot = 0;
    for (char ch : c) {
        if (ch >= '0' && ch <= '9') n = n * 10 + (ch - '0');
        else { if (ch == 'M' || ch == 'I' || ch == 'S' || ch == '=' || ch == 'X') tot += n; n = 0; }
    }
    return tot;
}

// E6 :20731-20838
static int get_onemapinfolist(const vmo_index* mi, const std::vector<Seg>& nal, const std::vector<std::vector<std::string>>& cigarlist,
                              int mapq, int64_t L, bool need_reverse, bool hardclip, std::vector<Record>& out) {
    out.clear();
    const char clip = hardclip ? 'H' : 'S';
    for (size_t i = 0; i < nal.size(); ++i) {
        const Seg& a = nal[i];
        int c = pos2contig(mi, a[0].r);
        int64_t bias = index_offset(mi, c);
        Record r; r.contig = c; r.mapq = mapq;
        std::string cg; for (const std::string& s : cigarlist[i]) cg += s;
        std::string top, tail;
        if (a[0].s == 1) {
            r.q_st = a[0].q; r.q_en = a.back().q + a.back().l;
            r.r_st = a[0].r - bias; r.r_en = a.back().r + a.back().l - bias;
            if (r.q_st > 0) top = std::to_string(r.q_st) + clip;
            if ((L - r.q_en) > 0) tail = std::to_string(L - r.q_en) + clip;
            if (a.back().l > 0) tail = std::to_string(a.back().l) + "M" + tail;
            r.strand = need_reverse ? -1 : 1;
        } else {
            r.q_st = L - a[0].q - a[0].l; r.q_en = L - a.back().q;
            r.r_st = a[0].r - bias; r.r_en = a.back().r + a.back().l - bias;
            if (r.q_st > 0) top = std::to_string(r.q_st) + clip;
            if ((L - r.q_en) > 0) tail = std::to_string(L - r.q_en) + clip;
            r.strand = need_reverse ? 1 : -1;
        }
        r.cigar = top + cg + tail;
        out.push_back(r);
    }
    for (const Record& r : out) {
        if (!hardclip) { if (L != cigar_qlen(r.cigar)) return -14; }            // :20779-20781
        else { if ((r.q_en - r.q_st) != cigar_qlen(r.cigar)) return -14; }      // :20784-20786
    }
    if (need_reverse) std::reverse(out.begin(), out.end());
    return 0;
}

// :5604-5650
bool pairedindel(const std::vector<std::string>& cigars, double indelsize) {
    std::vector<double> indel;
    for (const std::string& cg : cigars) {
        double number = 0.;
        for (char ch : cg) {
            int item = ch - '0';
            if (item < 10) number = number * 10. + item;   // :5624 (chars below '0' never occur in a CIGAR)
            else {
                if (ch != 'I' && ch != 'S' && ch != 'H' && ch != 'P') { if (ch == 'D' && number > indelsize) indel.push_back(number); number = 0.; }
                else { if (ch == 'I' && number > indelsize) indel.push_back(number); number = 0.; }
            }
        }
    }
    std::sort(indel.begin(), indel.end());
    double pre = 0; int clustersize = 1;
    for (double now : indel) {
        if ((std::min(pre, now) / std::max(pre, now)) > 0.7) { clustersize += 1; if (clustersize > 1) return true; }
        else clustersize = 1;
        pre = now;
    }
    return false;
}

}  // namespace vmo

extern "C" void vmo_surgery_counters(int64_t out[4], int reset) {
    for (int i = 0; i < 4; ++i) { out[i] = vmo::g_surgery[i].load(); if (reset) vmo::g_surgery[i].store(0); }
}

// Stage entry for golden vectors V4 (segment surgery): rows are (segment index, q, r, s, l).
//   fn 0 rebuild_chain_break(raw chain; arg = large_cost)   fn 1 drop_misplaced_alignment_test(segments; arg = iloc) -> *ret = removed
//   fn 2 merge_conjacent_alignment (getdupiloc inside)      fn 3 fix_simple_inv(segments, read; arg = 1 for mode R)
// returns 0, or the negative code of the condition under which the reference raises
extern "C" int vmo_stage_v4(const vmo_index* mi, int fn, const int64_t* rows_in, int64_t n_in, int64_t arg, const char* read, int64_t readlen,
                            int64_t** rows_out, int64_t* n_out, int* ret) {
    using namespace vmo;
    std::vector<Seg> al;
    for (int64_t i = 0; i < n_in; ++i) {
        const int64_t* r = rows_in + 5 * i;
        if ((int64_t)al.size() <= r[0]) al.resize((size_t)r[0] + 1);
        al[(size_t)r[0]].push_back(Anchor{r[1], r[2], r[3], r[4]});
    }
    int rc = 0; *ret = 0;
    if (fn == 0) { Path raw = al.empty() ? Path() : al[0]; std::vector<Seg> out; rc = raw.empty() ? -1 : rebuild_chain_break(mi, raw, arg, 50, out); al.swap(out); }
    else if (fn == 1) *ret = drop_misplaced_alignment_test(al, arg) ? 1 : 0;
    else if (fn == 2) merge_conjacent_alignment(mi, al);
    else if (fn == 3) rc = fix_simple_inv(mi, al, std::string(read, (size_t)readlen), arg != 0);
    else return -1;
    size_t tot = 0; for (auto& sg : al) tot += sg.size();
    int64_t* o = (int64_t*)malloc(sizeof(int64_t) * 5 * (tot ? tot : 1));
    size_t x = 0;
    for (size_t si = 0; si < al.size(); ++si) for (const Anchor& a : al[si]) { o[5 * x] = (int64_t)si; o[5 * x + 1] = a.q; o[5 * x + 2] = a.r; o[5 * x + 3] = a.s; o[5 * x + 4] = a.l; ++x; }
    *rows_out = o; *n_out = (int64_t)tot;
    return rc;
}

namespace vmo {

// extend_func :19238-19303
int extend_func(const vmo_index* mi, const std::string& read, const std::string& rc, Path chain_asc, int mapq,
                bool need_reverse, bool nofilter, const vmo_params& prm, std::vector<Record>& recs, bool* filtered_out) {
    const int64_t L = (int64_t)read.size();
    recs.clear();
    std::vector<Seg> al;
    const bool asmv = prm.mode == VMO_MODE_ASM;      // mammap_asm.py:22317-22363: small_alignment 40, no drop_misplaced pass, filtered = False
    int rcode = rebuild_chain_break(mi, chain_asc, prm.local_maxdiff, asmv ? 40 : 50, al, asmv);
    if (rcode < 0) return rcode;
    // divergence filter :19246-19254
    for (int64_t t = 0; t < (int64_t)al.size(); ++t) {
        std::string target, query;
        get_query_target_for_cigar(mi, al[t].front(), al[t].back(), read, rc, L, target, query);
        if (g_dplog) g_dplog->push_back(DpCall{2, target, query});
        size_t mn = std::min(target.size(), query.size());
        if (mn == 0) return -15;   // ZeroDivisionError
        double diffratio = (double)edit_distance_str(query, target) / (double)mn;
        if (diffratio > prm.maxdivergence) { al.erase(al.begin() + t); --t; }
    }
    extend_edge_test(mi, read, L, al);
    size_t o_len = al.size();
    bool filtered = false;
    if (al.size() > 2 && !nofilter && !asmv) {
        int64_t iloc = 0;
        while (iloc < (int64_t)al.size() - 2) { if (drop_misplaced_alignment_test(al, iloc)) continue; else iloc += 1; }
    }
    if (al.size() < o_len) { filtered = true; extend_edge_test(mi, read, L, al); }
    merge_conjacent_alignment(mi, al);
    rcode = fix_simple_inv(mi, al, read, prm.mode == VMO_MODE_R || asmv);      // mammap_asm.py:17158-17207 = the mode-R body
    if (rcode < 0) return rcode;
    std::vector<Seg> nal; std::vector<std::vector<std::string>> cigarlist;
    for (const Seg& a : al) {
        Seg oa; std::vector<std::string> cg;
        rcode = split_alignment_test(mi, a, read, rc, L, prm.eqx, oa, cg, asmv);
        if (rcode < 0) return rcode;
        nal.push_back(oa); cigarlist.push_back(cg);
    }
    rcode = get_onemapinfolist(mi, nal, cigarlist, mapq, L, need_reverse, prm.hardclip != 0, recs);
    if (rcode < 0) { recs.clear(); return rcode; }
    if (filtered_out) *filtered_out = filtered;
    return 0;
}

// ass_extend_func mammap_asm.py:23423-23460: large_cost 50, small_alignment 30, no divergence filter, MAPQ 60, forward orientation
int ass_extend_func(const vmo_index* mi, const std::string& read, const std::string& rc, Path chain_asc, const vmo_params& prm, std::vector<Record>& recs) {
    const int64_t L = (int64_t)read.size();
    recs.clear();
    std::vector<Seg> al;
    int rcode = rebuild_chain_break(mi, chain_asc, 50, 30, al, true);
    if (rcode < 0) return rcode;
    extend_edge_test(mi, read, L, al);
    merge_conjacent_alignment(mi, al);
    rcode = fix_simple_inv(mi, al, read, true);
    if (rcode < 0) return rcode;
    std::vector<Seg> nal; std::vector<std::vector<std::string>> cigarlist;
    for (const Seg& a : al) {
        Seg oa; std::vector<std::string> cg;
        rcode = split_alignment_test(mi, a, read, rc, L, prm.eqx, oa, cg, true);
        if (rcode < 0) return rcode;
        nal.push_back(oa); cigarlist.push_back(cg);
    }
    rcode = get_onemapinfolist(mi, nal, cigarlist, 60, L, false, prm.hardclip != 0, recs);
    if (rcode < 0) { recs.clear(); return rcode; }
    return 0;
}

// get_readmap_DP_test :24023-24084
int align_read(const vmo_index* mi, const std::string& read_in, const vmo_params& prm, std::vector<Record>& recs) {
    recs.clear();
    if (prm.mode == VMO_MODE_ASM) return align_asm(mi, read_in, prm, 0, 0, 0, recs);
    std::string read = read_in;
    for (char& c : read) if (c >= 'a' && c <= 'z') c -= 32;   // driver upper-cases reads (src/vacmap/vacmap:449)
    const int64_t L = (int64_t)read.size();
    std::string rc = revcomp(read);
    std::vector<Anchor> A;
    map_read(mi, read.data(), L, prm.check_num, prm.mid_occ, A);
    ChainSet cs;
    int rcode = decode_hit(A, L, vmo_index_k(mi), prm, cs);
    if (rcode < 0) return rcode;
    if (cs.score == 0.) return 0;
    bool need_reverse = cs.score < 0.;
    double lscore; Path chain_desc;
    if (need_reverse) std::swap(read, rc);
    rcode = local_chain(mi, read, rc, cs.paths, prm, &lscore, chain_desc, nullptr, nullptr);
    if (rcode < 0) return rcode;
    if (chain_desc.size() <= 1) return 0;
    Path chain_asc(chain_desc.rbegin(), chain_desc.rend());
    bool filtered = false;
    rcode = extend_func(mi, read, rc, chain_asc, cs.mapq, need_reverse, prm.nodiscard != 0, prm, recs, &filtered);
    if (rcode < 0) { recs.clear(); return rcode; }
    if (recs.empty()) return 0;
    if (!prm.nodiscard && filtered) {
        std::vector<std::string> cg; for (const Record& r : recs) cg.push_back(r.cigar);
        if (pairedindel(cg, 30)) {
            rcode = extend_func(mi, read, rc, chain_asc, cs.mapq, need_reverse, true, prm, recs, &filtered);
            if (rcode < 0) { recs.clear(); return rcode; }
        }
    }
    return 0;
}

}  // namespace vmo

using namespace vmo;

static void pack_records(const std::vector<std::vector<Record>>& per_read, vmo_record** recs, int64_t* n_recs, char** blob) {
    size_t nr = 0, nb = 0;
    for (auto& v : per_read) for (auto& r : v) { ++nr; nb += r.cigar.size() + 1; }
    *recs = (vmo_record*)malloc(sizeof(vmo_record) * (nr ? nr : 1));
    *blob = (char*)malloc(nb ? nb : 1);
    size_t ri = 0, bo = 0;
    for (size_t i = 0; i < per_read.size(); ++i)
        for (auto& r : per_read[i]) {
            vmo_record& o = (*recs)[ri++];
            o.read_idx = (int32_t)i; o.contig = r.contig; o.strand = r.strand; o.mapq = r.mapq;
            o.q_st = r.q_st; o.q_en = r.q_en; o.r_st = r.r_st; o.r_en = r.r_en;
            o.cigar_off = (int64_t)bo; o.cigar_len = (int64_t)r.cigar.size();
            memcpy(*blob + bo, r.cigar.c_str(), r.cigar.size() + 1); bo += r.cigar.size() + 1;
        }
    *n_recs = (int64_t)nr;
}

extern "C" {

int vmo_extend(const vmo_index* mi, const char* read, int64_t readlen, const int64_t* ch, int64_t n_chain, int mapq,
               int need_reverse, int nofilter, const vmo_params* p, vmo_record** recs, int64_t* n_recs, char** blob, int32_t* filtered) {
    std::string rd(read, (size_t)readlen);
    Path chain(n_chain);
    for (int64_t i = 0; i < n_chain; ++i) chain[i] = Anchor{ch[4 * i], ch[4 * i + 1], ch[4 * i + 2], ch[4 * i + 3]};
    std::vector<std::vector<Record>> pr(1); bool f = false;
    int rc = extend_func(mi, rd, revcomp(rd), chain, mapq, need_reverse != 0, nofilter != 0, *p, pr[0], &f);
    if (rc < 0) pr[0].clear();
    pack_records(pr, recs, n_recs, blob);
    if (filtered) *filtered = f;
    return rc;
}

int vmo_align_read(const vmo_index* mi, const char* read, int64_t readlen, const vmo_params* p, vmo_record** recs, int64_t* n_recs, char** blob) {
    std::vector<std::vector<Record>> pr(1);
    int rc = align_read(mi, std::string(read, (size_t)readlen), *p, pr[0]);
    if (rc < 0) pr[0].clear();
    pack_records(pr, recs, n_recs, blob);
    return rc;
}

int vmo_align_batch(const vmo_index* mi, const vmo_params* p, int64_t n_reads, const char* seqs, const int64_t* offsets, int nthreads,
                    vmo_record** recs, int64_t* n_recs, char** blob, int32_t* status) {
    // many threads: keep the per-read megabyte-sized scratch vectors (9-mer tables, DP matrices) inside the malloc arenas — with the
    // default 128 KiB mmap threshold every one of them is an mmap/munmap pair and the threads queue on the process's address-space lock
    if (nthreads > 1 && !getenv("VMO_NO_MALLOPT")) { mallopt(M_MMAP_THRESHOLD, 256 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 64 << 20); }
    std::vector<std::vector<Record>> pr((size_t)n_reads);
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        while (true) {
            int64_t i = next.fetch_add(1);
            if (i >= n_reads) break;
            int rc = align_read(mi, std::string(seqs + offsets[i], (size_t)(offsets[i + 1] - offsets[i])), *p, pr[i]);
            if (rc < 0) pr[i].clear();
            if (status) status[i] = rc;
        }
    };
    if (nthreads <= 1) work();
    else { std::vector<std::thread> th; for (int t = 0; t < nthreads; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
    pack_records(pr, recs, n_recs, blob);
    return 0;
}

// DP-call log for golden V5 / tests: begin -> run vmo_extend / vmo_align_read on this thread -> fetch
static thread_local std::vector<DpCall> t_log;
void vmo_dplog_begin(void) { t_log.clear(); g_dplog = &t_log; }
int64_t vmo_dplog_end(void) { g_dplog = nullptr; return (int64_t)t_log.size(); }
int vmo_dplog_get(int64_t i, int32_t* kind, const char** t, int64_t* tl, const char** q, int64_t* ql) {
    if (i < 0 || i >= (int64_t)t_log.size()) return -1;
    *kind = t_log[i].kind; *t = t_log[i].t.data(); *tl = (int64_t)t_log[i].t.size(); *q = t_log[i].q.data(); *ql = (int64_t)t_log[i].q.size();
    return 0;
}

}  // extern "C"
