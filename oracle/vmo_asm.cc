// vmo_asm.cc — CPU ORACLE (test infrastructure): the -mode asm path.
//
// `-mode asm` runs /root/reference/src/vacmap/mammap_asm.py, an OLDER FORK of the per-read path. Restated here, from the
// LIVE definitions of that file:
//   assembly_get_readmap_DP_test              :23204-23422   contigs of 500 000 bases and more: linked chain DPs over seeding batches
//   yield_mapinfo                             :22411-22443   100 kb seeding windows, batches of more than 500 000 anchors
//   yield_second_mapinfo                      :22444-22476   second-round batches along the first-round path
//   get_readmap_DP_test                       :19681-19739   contigs below 500 000 bases
//   decode_hit                                :21280-21348   (+ hit2work_1 :18435-18625, return_main_alignment_size :21244-21279)
// and, next to the functions they fork from: chain_exact_asm (vmo_chain.cc; GC-exact :20551, linked :21686, linked LC :21504), fast_dp variant 4
// (vmo_chain_fast.cc; GC-fast :20738, linked :21871), local_chain_asm + local_seed_one (vmo_local.cc; :16540, :17960, collect_second_round_anchors
// :22477), rebuild_chain_break / split_alignment_test / link_cigar / extend_func / ass_extend_func (vmo_extend.cc; :13256, :22197, :22365,
// :22317, :23423). extend_edge_test, merge_conjacent_alignment, getdupiloc_numba, get_onemapinfolist, pairedindel and
// get_reversed_chain_numpy_rough are byte-identical to mode H's; fix_simple_inv is mode R's body.
//
// The reference spills every batch's (anchors, P) to <workdir>/<n>.npz and reads them back for the traceback (:23273, :23281); the oracle keeps
// them in memory — the values are the same. A Python exception makes the reference skip the contig (:23493-23498): negative status here.
#include "vmo_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

namespace vmo {

static const int64_t NOPRE = -9999999;

// :21244-21279 on a path in ascending read order: the longest co-linear stretch (first, last anchor)
static void return_main_alignment_size(const vmo_index* mi, const Path& raw, Anchor& st_out, Anchor& en_out) {
    Anchor pre = raw[0];
    st_out = pre; en_out = pre;
    int64_t size = 0;
    Anchor st_item = pre;
    for (size_t x = 1; x < raw.size(); ++x) {
        const Anchor& now = raw[x];
        if (pre.s == now.s) {
            const int64_t readgap = now.q - pre.q - pre.l;
            if (readgap < 0) continue;
            int64_t refgap;
            if (pre.s == 1) refgap = now.r - pre.r - pre.l; else refgap = pre.r - now.r - now.l;
            if (std::llabs(readgap - refgap) <= 30 && refgap >= 0) {
                if (pos2contig(mi, pre.r) == pos2contig(mi, now.r)) { pre = now; continue; }
            }
        }
        if (pre.q - st_item.q > size) { size = pre.q - st_item.q; st_out = st_item; en_out = pre; }
        pre = now;
        st_item = pre;
    }
    if (pre.q - st_item.q > size) { size = pre.q - st_item.q; st_out = st_item; en_out = pre; }
}

// decode_hit :21280-21348 + hit2work_1 :18435-18625. read / rc: the contig as given and its reverse complement.
static int decode_hit_asm(const vmo_index* mi, const std::string& read, const std::string& rc, std::vector<Anchor> A, int kmersize,
                          const vmo_params& prm, ChainSet& out) {
    out = ChainSet();
    const int64_t readlen = (int64_t)read.size();
    out.need_reverse = strand_flip(A, readlen);
    if (A.size() <= 2) return 0;                                      // :21285 scores = 0
    const int64_t bin_size = 100;
    const int64_t n = (int64_t)A.size();
    bool fast_enable = ((double)n / (double)readlen) > 5.0;           // :18479
    std::stable_sort(A.begin(), A.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
    std::vector<double> S; std::vector<int64_t> P, S_arg;
    int64_t g_max_index = 0;
    if (!fast_enable) g_max_index = chain_exact_asm(A, kmersize, prm.global_skipcost, prm.global_maxdiff, 1000, false, nullptr, S, P, S_arg);
    if (fast_enable || g_max_index == -1) {
        fast_enable = true;
        g_max_index = chain_global_fast_asm(A, kmersize, prm.global_skipcost, prm.global_maxdiff, 1000, nullptr, S, P, S_arg);
        if (g_max_index < 0) return -2;
    }
    out.fast_used = fast_enable;
    const double scores = S[g_max_index];
    std::vector<char> used(n, 0);
    std::vector<Path> path_list;
    std::vector<double> scores_list;
    bool hit = false;
    double max_scores = 0;
    {
        Path path;
        int64_t take = g_max_index;
        used[take] = 1;
        const double score = S[take];
        while (true) { path.push_back(A[take]); if (P[take] == NOPRE) break; take = P[take]; used[take] = 1; }
        if (score > 40) { hit = true; scores_list.push_back(score); path_list.push_back(path); }
        if (scores > max_scores) max_scores = scores;
    }
    for (int64_t x = n - 1; x >= 0; --x) {                            // :18521
        int64_t take = S_arg[x];
        if (used[take]) continue;
        Path path;
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
    if (!(hit && max_scores > 40)) return 0;                          // :18551
    const int64_t m_paths = (int64_t)path_list.size();
    std::vector<int64_t> order(m_paths);
    for (int64_t i = 0; i < m_paths; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return scores_list[a] < scores_list[b]; });
    std::reverse(order.begin(), order.end());                         // no "path 0 first" swap in this fork
    auto binset = [&](const Path& p) { std::set<int64_t> s; for (const Anchor& a : p) s.insert(a.q / bin_size); return s; };
    std::vector<std::set<int64_t>> prim_sets;
    std::vector<std::vector<double>> prim_scores;
    std::vector<std::vector<int64_t>> all_index;
    prim_sets.push_back(binset(path_list[order[0]]));
    prim_scores.push_back({scores_list[order[0]]});
    all_index.push_back({order[0]});
    for (int64_t oi = 1; oi < m_paths; ++oi) {
        const int64_t iloc = order[oi];
        std::set<int64_t> b = binset(path_list[iloc]);
        double maxov = 0.; size_t prefer = 0;
        for (size_t p = 0; p < prim_sets.size(); ++p) {
            size_t inter = 0;
            for (int64_t v : b) if (prim_sets[p].count(v)) ++inter;
            const double ov = (double)inter / (double)std::min(prim_sets[p].size(), b.size());
            if (ov > maxov) { maxov = ov; prefer = p; }
        }
        if (maxov < 0.5) { prim_sets.push_back(b); prim_scores.push_back({scores_list[iloc]}); all_index.push_back({iloc}); }
        else { prim_scores[prefer].push_back(scores_list[iloc]); all_index[prefer].push_back(iloc); }
    }
    const double mlen = (double)path_list[order[0]].size();
    const double f1 = prim_scores[0][0], f2 = prim_scores[0].size() < 2 ? 0.0 : prim_scores[0][1];
    {
        double v = 40 * (1 - f2 / f1);
        v = v * std::min(1.0, mlen / 10);
        v = v * std::log(f1);
        out.mapq = (int)std::min<int64_t>((int64_t)v, 60);
    }
    int64_t base_iloc = order[0];
    if (out.mapq == 0) {                                              // :21302-21326: among the (near-)equal chains take the least divergent one
        const double base_score = prim_scores[0][0];
        double min_diff = 10;
        for (size_t t = 0; t < prim_scores[0].size(); ++t) {
            if (prim_scores[0][t] / base_score < 0.999) break;
            const Path& pd = path_list[all_index[0][t]];
            Path asc(pd.rbegin(), pd.rend());
            Anchor pre, now;
            return_main_alignment_size(mi, asc, pre, now);
            if (pre.s != now.s || pre.q == now.q) continue;
            std::string target, query;
            if (!out.need_reverse) get_query_target_for_cigar(mi, pre, now, read, rc, readlen, target, query);
            else get_query_target_for_cigar(mi, pre, now, rc, read, readlen, target, query);
            const size_t mn = std::min(target.size(), query.size());
            if (mn == 0) return -15;                                  // ZeroDivisionError
            if (g_dplog) g_dplog->push_back(DpCall{2, target, query});
            const double diff = (double)edit_distance_str(query, target) / (double)mn;
            if (diff <= min_diff) { min_diff = diff; base_iloc = all_index[0][t]; }
        }
    }
    out.all_scores = scores_list;
    out.paths.push_back(path_list[base_iloc]);
    out.score = out.need_reverse ? -scores_list[base_iloc] : scores_list[base_iloc];
    return 0;
}

// get_readmap_DP_test :19681-19739 (check_num = -1 from :23206)
static int align_small(const vmo_index* mi, std::string read, const vmo_params& prm, std::vector<Record>& recs) {
    const int64_t L = (int64_t)read.size();
    std::string rc = revcomp(read);
    std::vector<Anchor> A;
    map_read(mi, read.data(), L, prm.check_num, prm.mid_occ, A);
    ChainSet cs;
    int rcode = decode_hit_asm(mi, read, rc, A, vmo_index_k(mi), prm, cs);
    if (rcode < 0) return rcode;
    if (cs.score == 0.) return 0;
    const bool need_reverse = cs.score < 0.;
    double lscore; Path chain_desc;
    if (need_reverse) std::swap(read, rc);
    rcode = local_chain(mi, read, rc, cs.paths, prm, &lscore, chain_desc, nullptr, nullptr);
    if (rcode < 0) return rcode;
    if (chain_desc.size() <= 1) return 0;
    Path chain_asc(chain_desc.rbegin(), chain_desc.rend());
    bool filtered = false;
    rcode = extend_func(mi, read, rc, chain_asc, cs.mapq, need_reverse, prm.nodiscard != 0, prm, recs, &filtered);
    if (rcode < 0) { recs.clear(); return rcode; }
    return 0;
}

// stage capture for the tests of the device's long-contig loop: 1 = first-round path (descending read order), 2 = the chain handed to
// ass_extend_func (ascending, overlaps trimmed), 3 = the second-round anchor batches (rows, batch after batch)
thread_local int g_trace_which = 0;
thread_local std::vector<Anchor>* g_trace_rows = nullptr;
thread_local std::vector<int64_t>* g_trace_off = nullptr;

struct SavedBatch { std::vector<Anchor> rows; std::vector<int64_t> P; };

// the body shared by the two rounds of assembly_get_readmap_DP_test (:23228-23275 / :23328-23373): link, chain, carry, "save"
struct Linker {
    int kmersize; double skipcost; int maxdiff; int maxgap; bool lc;
    double g_max_scores = 0.; int64_t g_max_index = 0;
    std::vector<double> pre_S; std::vector<int64_t> pre_P; std::vector<Anchor> pre_rows;
    std::vector<SavedBatch> saved;
    int64_t pre_g_max_index = 0; bool have = false;
    int feed(const std::vector<Anchor>& one) {
        if (one.empty()) return 0;
        std::vector<Anchor> linked; int64_t prereadloc;
        if (!pre_rows.empty()) {
            linked = pre_rows; linked.insert(linked.end(), one.begin(), one.end());
            prereadloc = 0;
            for (const Anchor& a : pre_rows) if (prereadloc < a.q) prereadloc = a.q;
        } else { linked = one; prereadloc = one[0].q; }
        LinkState ls; ls.pre_S = pre_S.data(); ls.pre_P = pre_P.data(); ls.n_pre = (int64_t)pre_S.size();
        ls.g_max_scores = g_max_scores; ls.g_max_index = g_max_index; ls.prereadloc = prereadloc;
        std::vector<double> S; std::vector<int64_t> P, S_arg;
        int64_t g = chain_exact_asm(linked, kmersize, skipcost, maxdiff, maxgap, lc, &ls, S, P, S_arg);
        if (g == -1) {                                                // :23246-23247 (first round only; the linked LC never bails out)
            g = chain_global_fast_asm(linked, kmersize, skipcost, maxdiff, maxgap, &ls, S, P, S_arg);
            if (g < 0) return -2;
        }
        pre_g_max_index = g; have = true;
        if (P[g] < 0) return 0;                                       // :23250 `continue`: nothing carried, nothing saved
        const int64_t n = (int64_t)S.size();
        g_max_scores = S[S_arg[n - 1]];
        const double lowestscores = g_max_scores - skipcost - 36 - 20;
        int64_t sliceiloc = n - 1;
        if (sliceiloc > 0) { while (lowestscores < S[S_arg[sliceiloc]]) { sliceiloc -= 1; if (sliceiloc == 0) break; } }
        else return -17;                                              // raise Exception("ERROR: ") :23266
        const double base = S[S_arg[sliceiloc]];
        pre_S.clear(); pre_P.clear(); pre_rows.clear();
        for (int64_t x = sliceiloc; x < n; ++x) {
            const int64_t j = S_arg[x];
            pre_S.push_back(S[j] - base + 1000);
            pre_P.push_back(-P[j]);
            pre_rows.push_back(linked[j]);
        }
        g_max_index = (int64_t)pre_S.size() - 1;
        g_max_scores = pre_S.back();
        saved.push_back(SavedBatch{std::move(linked), std::move(P)});
        return 0;
    }
    // :23277-23292 / :23379-23396. start = the pre_g_max_index in force (the second round falls back to the first round's when it fed nothing)
    int traceback(int64_t start, Path& path) const {
        path.clear();
        int64_t gi = start;
        for (int64_t d = (int64_t)saved.size() - 1; d >= 0; --d) {
            const SavedBatch& b = saved[d];
            const int64_t n = (int64_t)b.rows.size();
            int64_t take = gi;
            if (take < 0 || take >= n) return -19;                    // IndexError
            path.push_back(b.rows[take]);
            while (true) {
                if (b.P[take] < 0) break;
                take = b.P[take];
                if (take >= n) return -19;
                path.push_back(b.rows[take]);
            }
            gi = std::llabs(b.P[take]);
        }
        return 0;
    }
};

int align_asm(const vmo_index* mi, const std::string& contig_in, const vmo_params& prm, int64_t split_len, int64_t batch_anchors, int64_t window,
              std::vector<Record>& recs) {
    recs.clear();
    if (split_len <= 0) split_len = 500000;
    if (batch_anchors <= 0) batch_anchors = 500000;
    if (window <= 0) window = 100000;
    std::string seq = contig_in;
    for (char& c : seq) if (c >= 'a' && c <= 'z') c -= 32;
    const int64_t L = (int64_t)seq.size();
    if (L < split_len) return align_small(mi, seq, prm, recs);         // :23205
    const std::string rc = revcomp(seq);
    // ---- first round :23214-23292
    Linker r1; r1.kmersize = vmo_index_k(mi); r1.skipcost = prm.global_skipcost; r1.maxdiff = prm.global_maxdiff; r1.maxgap = 1000; r1.lc = false;
    {
        // yield_mapinfo :22411-22443 (including its final flush, which appends the last window's anchors a second time when that window had
        // already gone into the cache)
        std::vector<std::vector<Anchor>> cache; int64_t cache_size = 0;
        std::vector<Anchor> one;
        auto flush = [&](std::vector<Anchor>& batch) -> int {
            std::stable_sort(batch.begin(), batch.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
            return r1.feed(batch);
        };
        for (int64_t st = 0; st < L; st += window) {
            const int64_t en = std::min(st + window, L);
            one.clear();
            map_read(mi, seq.data() + st, en - st, -1, -1, one);
            for (Anchor& a : one) a.q += st;
            if ((int64_t)one.size() + cache_size > batch_anchors) {
                if (cache_size > 0) {
                    if (!one.empty()) cache.push_back(one);
                    std::vector<Anchor> all; for (auto& c : cache) all.insert(all.end(), c.begin(), c.end());
                    one.swap(all); cache_size = 0; cache.clear();
                }
                std::vector<Anchor> batch = one;
                const int rcode = flush(batch);
                if (rcode < 0) return rcode;
            } else if (!one.empty()) { cache.push_back(one); cache_size += (int64_t)one.size(); }
        }
        if (cache_size > 0) {
            if (!one.empty()) cache.push_back(one);
            std::vector<Anchor> all; for (auto& c : cache) all.insert(all.end(), c.begin(), c.end());
            const int rcode = flush(all);
            if (rcode < 0) return rcode;
        }
    }
    if (!r1.have) return -18;                                          // NameError: pre_g_max_index (:23278)
    Path path;
    int rcode = r1.traceback(r1.pre_g_max_index, path);
    if (rcode < 0) return rcode;
    if (g_trace_which == 1 && g_trace_rows) *g_trace_rows = path;
    if (path.size() <= 1) return 0;
    // ---- second round :23309-23396
    const int k2 = prm.local_kmersize;
    Linker r2; r2.kmersize = k2; r2.skipcost = prm.local_skipcost; r2.maxdiff = prm.local_maxdiff; r2.maxgap = 99; r2.lc = true;
    {
        const Path raw(path.rbegin(), path.rend());                    // ascending read order
        const int64_t np_ = (int64_t)raw.size();
        auto collect = [&](int64_t st_read, int64_t en_read, int64_t lo, int64_t hi) -> int {   // collect_second_round_anchors on raw[lo:hi]
            if (hi <= lo) return -16;                                  // raw_alignment_array[0] on an empty slice
            Path guide(raw.begin() + lo, raw.begin() + hi);
            std::vector<Anchor> out;
            local_seed_one(mi, seq, guide, k2, 2000, 500, out, st_read, en_read);
            if (out.empty()) return -16;                               // np.array([])[:, 0] (:22755)
            std::stable_sort(out.begin(), out.end(), [](const Anchor& a, const Anchor& b) { return a.q < b.q; });
            if (g_trace_which == 3 && g_trace_rows) { g_trace_rows->insert(g_trace_rows->end(), out.begin(), out.end()); g_trace_off->push_back((int64_t)g_trace_rows->size()); }
            return r2.feed(out);
        };
        // yield_second_mapinfo :22444-22476
        int64_t st_read = 0, st_path = 0, iloc_path = 0;
        for (int64_t x = 1; x < np_; ++x) {
            const Anchor& now = raw[x];
            iloc_path += 1;
            if (iloc_path == np_ - 1 || (iloc_path < np_ - 1 && raw[iloc_path + 1].q > raw[iloc_path].q)) {
                if ((now.q + now.l) > (st_read + window) && (iloc_path - st_path) > 300) {
                    const int64_t en_read = raw[iloc_path].q;
                    rcode = collect(st_read, en_read, std::max<int64_t>(0, st_path - 20), std::min<int64_t>(iloc_path + 20, np_));
                    if (rcode < 0) return rcode;
                    st_path = iloc_path + 1;
                    st_read = en_read;
                }
            }
        }
        if (st_read < L) {
            rcode = collect(st_read, L, std::max<int64_t>(0, st_path - 20), std::min<int64_t>(iloc_path + 20, np_));
            if (rcode < 0) return rcode;
        }
    }
    Path path2;
    rcode = r2.traceback(r2.have ? r2.pre_g_max_index : r1.pre_g_max_index, path2);
    if (rcode < 0) return rcode;
    if (path2.size() <= 1) return 0;
    // :23400-23413 trim read overlaps (against the UNtrimmed neighbour), ascending order, extension
    {
        Anchor pre = path2[0];
        for (size_t x = 1; x < path2.size(); ++x) {
            const Anchor now = path2[x];
            if (!(pre.q >= now.q + now.l)) {
                if (now.s == 1) path2[x] = Anchor{now.q, now.r, now.s, pre.q - now.q};
                else path2[x] = Anchor{now.q, now.r + now.l - pre.q + now.q, now.s, pre.q - now.q};
            }
            pre = now;
        }
    }
    Path asc(path2.rbegin(), path2.rend());
    if (g_trace_which == 2 && g_trace_rows) *g_trace_rows = asc;
    rcode = ass_extend_func(mi, seq, rc, asc, prm, recs);
    if (rcode < 0) { recs.clear(); return rcode; }
    return 0;
}

}  // namespace vmo

using namespace vmo;

extern "C" {

int vmo_align_asm(const vmo_index* mi, const char* contig, int64_t len, const vmo_params* p, int64_t split_len, int64_t batch_anchors,
                  int64_t window, vmo_record** recs, int64_t* n_recs, char** blob) {
    std::vector<Record> r;
    int rc = align_asm(mi, std::string(contig, (size_t)len), *p, split_len, batch_anchors, window, r);
    if (rc < 0) r.clear();
    size_t nb = 0; for (auto& x : r) nb += x.cigar.size() + 1;
    *recs = (vmo_record*)malloc(sizeof(vmo_record) * (r.size() ? r.size() : 1));
    *blob = (char*)malloc(nb ? nb : 1);
    size_t bo = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        vmo_record& o = (*recs)[i];
        o.read_idx = 0; o.contig = r[i].contig; o.strand = r[i].strand; o.mapq = r[i].mapq;
        o.q_st = r[i].q_st; o.q_en = r[i].q_en; o.r_st = r[i].r_st; o.r_en = r[i].r_en;
        o.cigar_off = (int64_t)bo; o.cigar_len = (int64_t)r[i].cigar.size();
        memcpy(*blob + bo, r[i].cigar.c_str(), r[i].cigar.size() + 1); bo += r[i].cigar.size() + 1;
    }
    *n_recs = (int64_t)r.size();
    return rc;
}

int vmo_asm_trace(const vmo_index* mi, const char* contig, int64_t len, const vmo_params* p, int64_t split_len, int64_t batch_anchors, int64_t window,
                  int which, int64_t** rows, int64_t* n_rows, int64_t** off, int64_t* n_off) {
    std::vector<Anchor> tr; std::vector<int64_t> to(1, 0);
    g_trace_which = which; g_trace_rows = &tr; g_trace_off = &to;
    std::vector<Record> r;
    const int rc = align_asm(mi, std::string(contig, (size_t)len), *p, split_len, batch_anchors, window, r);
    g_trace_which = 0; g_trace_rows = nullptr; g_trace_off = nullptr;
    *rows = (int64_t*)malloc(32 * (tr.size() ? tr.size() : 1));
    for (size_t i = 0; i < tr.size(); ++i) { (*rows)[4 * i] = tr[i].q; (*rows)[4 * i + 1] = tr[i].r; (*rows)[4 * i + 2] = tr[i].s; (*rows)[4 * i + 3] = tr[i].l; }
    *n_rows = (int64_t)tr.size();
    *off = (int64_t*)malloc(8 * to.size());
    for (size_t i = 0; i < to.size(); ++i) (*off)[i] = to[i];
    *n_off = (int64_t)to.size();
    return rc;
}

int vmo_decode_hit_asm(const vmo_index* mi, const char* contig, int64_t len, const vmo_params* prm, vmo_chains* out) {
    std::string seq(contig, (size_t)len);
    for (char& c : seq) if (c >= 'a' && c <= 'z') c -= 32;
    std::vector<Anchor> A;
    map_read(mi, seq.data(), len, prm->check_num, prm->mid_occ, A);
    ChainSet cs;
    const int rc = decode_hit_asm(mi, seq, revcomp(seq), A, vmo_index_k(mi), *prm, cs);
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

int64_t vmo_chain_linked_raw(const int64_t* a, int64_t n, int which, int kmersize, double skipcost, int maxdiff, int maxgap,
                             double g_max_scores, int64_t g_max_index, const double* pre_S, const int64_t* pre_P, int64_t n_pre,
                             int64_t prereadloc, double* S, int64_t* P, int64_t* S_arg) {
    std::vector<Anchor> v(n);
    for (int64_t i = 0; i < n; ++i) v[i] = Anchor{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    LinkState ls; ls.pre_S = pre_S; ls.pre_P = pre_P; ls.n_pre = n_pre; ls.g_max_scores = g_max_scores; ls.g_max_index = g_max_index; ls.prereadloc = prereadloc;
    std::vector<double> s; std::vector<int64_t> p, sa;
    int64_t g;
    if (which == 1) g = chain_global_fast_asm(v, kmersize, skipcost, maxdiff, maxgap, &ls, s, p, sa);
    else g = chain_exact_asm(v, kmersize, skipcost, maxdiff, maxgap, which == 2, &ls, s, p, sa);
    for (int64_t i = 0; i < n && i < (int64_t)s.size(); ++i) { S[i] = s[i]; P[i] = p[i]; S_arg[i] = sa[i]; }
    return g;
}

}  // extern "C"
