// vmo_seed.cc — CPU ORACLE (test infrastructure): minimizer sketch, index, map().
//
// Replaces the un-vendored C extension vacmap-index==0.0.3 (`Aligner`, `.map`, `.seq`, `.seq_offset`;
// call sites /root/reference/src/vacmap/mammap_clrnano.py:23985, src/vacmap/vacmap:344-367).
// Its source is NOT under /root/reference, so this file implements the build's own normative spec
// "VMX-S1" (DESIGN.md §Spec), which follows the published minimap2 design (2-bit k-mers, canonical
// strand, invertible 64-bit mix, (w,k) window minimizers, occurrence cap) and the clustering that
// the reference's superseded in-repo code shows (mammap_clrnano.py:1359-1383; prefix property :10285).
// PARITY UNPINNED for this file (no golden vector exists in the reference for map()).
#include "vmo_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace vmo {

Nt4Table::Nt4Table() {
    for (int i = 0; i < 256; ++i) t[i] = 4;
    t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
}
const Nt4Table NT4T;

std::string revcomp(const std::string& s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i], o = 'N';
        switch (c) { case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break; default: o = 'N'; }
        r[i] = o;
    }
    return r;
}

// Invertible integer mix (Thomas Wang), masked to 2k bits: the published minimap2 `hash64`.
static inline uint64_t hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// VMX-S1 sketch. For every k-mer start p in [0, L-k]: invalid if it holds a non-ACGT base or equals its
// reverse complement. canon = min(fwd, rc); strand = (rc < fwd); h = hash64(canon).
// p is a minimizer iff some window of w consecutive k-mer starts containing p (windows lie inside
// [0, L-k]; a single short window if fewer than w starts exist) has min hash == h(p) (all ties kept).
void sketch(const char* seq, int64_t len, int k, int w, std::vector<Mz>& out) {
    out.clear();
    int64_t P = len - k + 1;
    if (P <= 0) return;
    const uint64_t mask = (k < 32) ? ((1ULL << (2 * k)) - 1) : ~0ULL;
    const uint64_t INF = ~0ULL;
    std::vector<uint64_t> h(P, INF);
    std::vector<int8_t> z(P, 0);
    uint64_t fwd = 0, rc = 0;
    int l = 0;
    const int shift = 2 * (k - 1);
    for (int64_t i = 0; i < len; ++i) {
        int c = NT4[(uint8_t)seq[i]];
        if (c < 4) {
            fwd = ((fwd << 2) | (uint64_t)c) & mask;
            rc = (rc >> 2) | ((uint64_t)(3 - c) << shift);
            ++l;
        } else {
            l = 0;
        }
        if (l >= k) {
            int64_t p = i - k + 1;
            if (fwd != rc) {
                uint64_t canon = fwd < rc ? fwd : rc;
                z[p] = rc < fwd ? 1 : 0;
                h[p] = hash64(canon, mask);
                // a valid hash can be all-ones only if mask is all ones; k<=28 so never INF
            }
        }
    }
    int64_t nwin = P >= w ? P - w + 1 : 1;
    int64_t wl = P >= w ? w : P;
    // window minima
    std::vector<uint64_t> wmin(nwin);
    for (int64_t a = 0; a < nwin; ++a) {
        uint64_t m = INF;
        for (int64_t j = a; j < a + wl; ++j) m = h[j] < m ? h[j] : m;
        wmin[a] = m;
    }
    for (int64_t p = 0; p < P; ++p) {
        if (h[p] == INF) continue;
        int64_t a0 = p - wl + 1; if (a0 < 0) a0 = 0;
        int64_t a1 = p; if (a1 > nwin - 1) a1 = nwin - 1;
        bool sel = false;
        for (int64_t a = a0; a <= a1 && !sel; ++a) sel = (wmin[a] == h[p]);
        if (sel) out.push_back(Mz{h[p], (int32_t)p, z[p]});
    }
}

}  // namespace vmo

using namespace vmo;

struct vmo_index {
    int k, w, mid_occ;
    std::vector<std::string> names;
    std::vector<std::string> seqs;      // upper-cased
    std::vector<int64_t> offsets;       // global start of each contig
    std::vector<uint64_t> hashes;       // sorted (hash, pos)
    std::vector<uint64_t> positions;    // gpos<<1 | strand
    std::vector<uint64_t> dkeys;        // distinct hashes ascending
    std::vector<uint64_t> dstart;       // start into hashes/positions (size n_distinct+1)
};

static thread_local std::string g_err;
extern "C" const char* vmo_last_error(void) { return g_err.c_str(); }
namespace vmo { void set_error(const std::string& s) { g_err = s; } }

// threads the oracle may use for its one-off index build: the machine's cores, cut to the container's CPU quota (cgroup v2)
static unsigned build_threads() {
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long long per = 0;
        if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { long long c = (atoll(q) + per - 1) / per; if (c >= 1 && (unsigned)c < nt) nt = (unsigned)c; }
        fclose(f);
    }
    if (const char* e = getenv("VMO_THREADS")) { int v = atoi(e); if (v >= 1) nt = (unsigned)v; }
    return nt > 64 ? 64 : nt;
}
template <class F> static void parallel_jobs(size_t njobs, unsigned nt, F fn) {
    if (nt > njobs) nt = (unsigned)(njobs ? njobs : 1);
    std::atomic<size_t> next(0);
    auto work = [&]() { while (true) { size_t j = next.fetch_add(1); if (j >= njobs) break; fn(j); } };
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> th; for (unsigned t = 0; t < nt; ++t) th.emplace_back(work); for (auto& t : th) t.join();
}

// Index = every contig's minimizers (sketch() per contig, windows never cross contigs) as (hash, gpos<<1|strand) pairs in ascending
// order. Large references are sketched in chunks of k-mer starts by several threads: a chunk [st, en) is sketched together with
// w - 1 starts of context on each side, which shows sketch() every window of the contig that contains one of the chunk's starts, so
// the union over chunks equals sketch() of the whole contig; the pairs are then sorted by buckets of their leading hash bits.
static void index_finish(vmo_index* mi) {
    const int k = mi->k, w = mi->w;
    typedef std::pair<uint64_t, uint64_t> HP;
    struct Job { size_t contig; int64_t st, en; };
    std::vector<Job> jobs;
    int64_t off = 0;
    mi->offsets.clear();
    const int64_t CH = 4 << 20;
    for (size_t c = 0; c < mi->seqs.size(); ++c) {
        mi->offsets.push_back(off);
        const int64_t P = (int64_t)mi->seqs[c].size() - k + 1;
        for (int64_t s = 0; s < P; s += CH) jobs.push_back(Job{c, s, std::min<int64_t>(s + CH, P)});
        off += (int64_t)mi->seqs[c].size();
    }
    const unsigned nt = build_threads();
    const int NB = 256, hb = 2 * k >= 8 ? 2 * k - 8 : 0;           // bucket = leading 8 bits of the 2k-bit hash
    std::vector<std::vector<HP>> parts(jobs.size());
    std::vector<std::vector<size_t>> hist(jobs.size(), std::vector<size_t>(NB, 0));
    parallel_jobs(jobs.size(), nt, [&](size_t j) {
        const Job& jb = jobs[j];
        const std::string& sq = mi->seqs[jb.contig];
        const int64_t P = (int64_t)sq.size() - k + 1, goff = mi->offsets[jb.contig];
        const int64_t lo = std::max<int64_t>(jb.st - (w - 1), 0), hi = std::min<int64_t>(jb.en + (w - 1), P);
        std::vector<Mz> mz;
        sketch(sq.data() + lo, (hi - lo) + k - 1, k, w, mz);
        for (const Mz& m : mz) {
            const int64_t p = lo + m.pos;
            if (p < jb.st || p >= jb.en) continue;
            parts[j].emplace_back(m.h, ((uint64_t)(goff + p) << 1) | (uint64_t)m.strand);
            hist[j][(m.h >> hb) & (NB - 1)]++;
        }
    });
    size_t total = 0; std::vector<size_t> bstart(NB + 1, 0);
    for (int b = 0; b < NB; ++b) { bstart[b] = total; for (size_t j = 0; j < jobs.size(); ++j) { const size_t c = hist[j][b]; hist[j][b] = total; total += c; } }
    bstart[NB] = total;
    std::vector<HP> all(total);
    parallel_jobs(jobs.size(), nt, [&](size_t j) {
        for (const HP& e : parts[j]) all[hist[j][(e.first >> hb) & (NB - 1)]++] = e;
        std::vector<HP>().swap(parts[j]);
    });
    parallel_jobs((size_t)NB, nt, [&](size_t b) { std::sort(all.begin() + bstart[b], all.begin() + bstart[b + 1]); });
    mi->hashes.resize(all.size());
    mi->positions.resize(all.size());
    parallel_jobs((all.size() + (1 << 22) - 1) >> 22, nt, [&](size_t c) {
        const size_t e = std::min(all.size(), (c + 1) << 22);
        for (size_t i = c << 22; i < e; ++i) { mi->hashes[i] = all[i].first; mi->positions[i] = all[i].second; }
    });
    std::vector<HP>().swap(all);
    mi->dkeys.clear(); mi->dstart.clear();
    const std::vector<uint64_t>& H = mi->hashes;
    for (size_t i = 0; i < H.size(); ++i)
        if (i == 0 || H[i] != H[i - 1]) { mi->dkeys.push_back(H[i]); mi->dstart.push_back(i); }
    mi->dstart.push_back(H.size());
    // default occurrence cap: max(10, (count at the (1 - 2e-4) quantile of distinct minimizers) + 1)
    size_t nd = mi->dkeys.size();
    int occ = 10;
    if (nd > 0) {
        std::vector<uint32_t> cnt(nd);
        for (size_t i = 0; i < nd; ++i) cnt[i] = (uint32_t)(mi->dstart[i + 1] - mi->dstart[i]);
        size_t kth = (size_t)((1.0 - 2e-4) * (double)nd);
        if (kth >= nd) kth = nd - 1;
        std::nth_element(cnt.begin(), cnt.begin() + kth, cnt.end());
        int v = (int)cnt[kth] + 1;
        if (v > occ) occ = v;
    }
    mi->mid_occ = occ;
}

extern "C" {

void vmo_params_default(vmo_params* p, int mode) {
    memset(p, 0, sizeof(*p));
    p->mode = mode;
    p->check_num = 100; p->mid_occ = -1;
    p->global_maxdiff = 50; p->local_maxdiff = 30; p->local_kmersize = 9;
    p->eqx = 0; p->hardclip = 0;
    // src/vacmap/vacmap:257-296
    if (mode == VMO_MODE_L) { p->local_skipcost = 59.; p->global_skipcost = 40.; p->maxdivergence = 0.1; }
    else if (mode == VMO_MODE_H) { p->local_skipcost = 40.; p->global_skipcost = 40.; p->maxdivergence = 0.2; }
    else { p->local_skipcost = 30.; p->global_skipcost = 30.; p->maxdivergence = 0.5; }
    p->nodiscard = !(mode == VMO_MODE_L || mode == VMO_MODE_H);
    // -mode asm: --eqx forced (vacmap:246), maxdivergence forced to 1 by the worker (mammap_asm.py:23483), check_num = -1 (:23206, :22419)
    if (mode == VMO_MODE_ASM) { p->eqx = 1; p->maxdivergence = 1.0; p->check_num = -1; }
}

vmo_index* vmo_index_build_mem(int nseq, const char* const* names, const char* const* seqs, const int64_t* lens,
                               int k, int w) {
    if (k < 1 || k > 28 || w < 1 || w > 255) { set_error("bad k/w"); return nullptr; }
    vmo_index* mi = new vmo_index();
    mi->k = k; mi->w = w;
    for (int i = 0; i < nseq; ++i) {
        mi->names.emplace_back(names[i]);
        std::string s(seqs[i], (size_t)lens[i]);
        for (char& c : s) if (c >= 'a' && c <= 'z') c -= 32;
        mi->seqs.push_back(std::move(s));
    }
    index_finish(mi);
    return mi;
}

vmo_index* vmo_index_build_fasta(const char* path, int k, int w) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return nullptr; }
    std::vector<std::string> names, seqs;
    std::string line;
    char buf[1 << 16];
    bool inhdr = false;
    std::string cur;
    auto flush_line = [&](const std::string& ln) {
        if (ln.empty()) return;
        if (ln[0] == '>') {
            std::string nm = ln.substr(1);
            size_t e = nm.find_first_of(" \t");
            if (e != std::string::npos) nm.resize(e);
            names.push_back(nm); seqs.emplace_back();
        } else if (!seqs.empty()) {
            seqs.back() += ln;
        }
    };
    (void)inhdr;
    while (fgets(buf, sizeof buf, f)) {
        size_t n = strlen(buf);
        bool eol = n && buf[n - 1] == '\n';
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
        cur.append(buf, n);
        if (eol) { flush_line(cur); cur.clear(); }
    }
    flush_line(cur);
    fclose(f);
    std::vector<const char*> np, sp; std::vector<int64_t> ls;
    for (size_t i = 0; i < names.size(); ++i) { np.push_back(names[i].c_str()); sp.push_back(seqs[i].data()); ls.push_back((int64_t)seqs[i].size()); }
    return vmo_index_build_mem((int)names.size(), np.data(), sp.data(), ls.data(), k, w);
}

void vmo_index_free(vmo_index* mi) { delete mi; }
int vmo_index_k(const vmo_index* mi) { return mi->k; }
int vmo_index_w(const vmo_index* mi) { return mi->w; }
int vmo_index_nseq(const vmo_index* mi) { return (int)mi->seqs.size(); }
int vmo_index_mid_occ(const vmo_index* mi) { return mi->mid_occ; }
int64_t vmo_index_n_minimizers(const vmo_index* mi) { return (int64_t)mi->hashes.size(); }
int64_t vmo_index_n_distinct(const vmo_index* mi) { return (int64_t)mi->dkeys.size(); }
const char* vmo_index_seq_name(const vmo_index* mi, int i) { return mi->names[i].c_str(); }
int64_t vmo_index_seq_len(const vmo_index* mi, int i) { return (int64_t)mi->seqs[i].size(); }
int64_t vmo_index_seq_offset(const vmo_index* mi, int i) { return mi->offsets[i]; }
int64_t vmo_index_seq(const vmo_index* mi, int i, int64_t st, int64_t en, char* out) {
    int64_t L = (int64_t)mi->seqs[i].size();
    if (st < 0) st = 0; if (en > L) en = L;
    if (en <= st) return 0;
    memcpy(out, mi->seqs[i].data() + st, (size_t)(en - st));
    return en - st;
}
const uint64_t* vmo_index_hashes(const vmo_index* mi) { return mi->hashes.data(); }
const uint64_t* vmo_index_positions(const vmo_index* mi) { return mi->positions.data(); }

int64_t vmo_sketch(const char* seq, int64_t len, int k, int w, uint64_t* hash, int32_t* pos, int8_t* strand) {
    std::vector<Mz> mz;
    sketch(seq, len, k, w, mz);
    for (size_t i = 0; i < mz.size(); ++i) { hash[i] = mz[i].h; pos[i] = mz[i].pos; strand[i] = mz[i].strand; }
    return (int64_t)mz.size();
}

void vmo_free(void* p) { free(p); }

}  // extern "C"

namespace vmo {

const std::string& index_seq(const vmo_index* mi, int c) { return mi->seqs[c]; }
int64_t index_offset(const vmo_index* mi, int c) { return mi->offsets[c]; }
int index_nseq(const vmo_index* mi) { return (int)mi->seqs.size(); }

// VMX-S1 map(): hits -> sort by (r, q, s) -> clusters cut at ref gap > 5000 -> rank by
// (size desc, first r asc) -> keep first check_num clusters (all if <= 0) -> emit cluster by cluster.
void map_read(const vmo_index* mi, const char* seq, int64_t len, int check_num, int mid_occ, std::vector<Anchor>& out) {
    out.clear();
    std::vector<Mz> mz;
    sketch(seq, len, mi->k, mi->w, mz);
    if (mid_occ <= 0) mid_occ = mi->mid_occ;
    std::vector<Anchor> hits;
    for (const Mz& m : mz) {
        auto it = std::lower_bound(mi->dkeys.begin(), mi->dkeys.end(), m.h);
        if (it == mi->dkeys.end() || *it != m.h) continue;
        size_t d = (size_t)(it - mi->dkeys.begin());
        uint64_t s0 = mi->dstart[d], s1 = mi->dstart[d + 1];
        if ((int64_t)(s1 - s0) > mid_occ) continue;
        for (uint64_t j = s0; j < s1; ++j) {
            uint64_t pv = mi->positions[j];
            int zr = (int)(pv & 1);
            hits.push_back(Anchor{(int64_t)m.pos, (int64_t)(pv >> 1), zr == m.strand ? 1 : -1, (int64_t)mi->k});
        }
    }
    std::sort(hits.begin(), hits.end(), [](const Anchor& a, const Anchor& b) {
        if (a.r != b.r) return a.r < b.r;
        if (a.q != b.q) return a.q < b.q;
        return a.s < b.s;
    });
    struct Cl { int64_t st, n, r0; };
    std::vector<Cl> cls;
    for (size_t i = 0; i < hits.size(); ++i) {
        if (i == 0 || hits[i].r - hits[i - 1].r > 5000) cls.push_back(Cl{(int64_t)i, 0, hits[i].r});
        cls.back().n++;
    }
    std::stable_sort(cls.begin(), cls.end(), [](const Cl& a, const Cl& b) {
        if (a.n != b.n) return a.n > b.n;
        return a.r0 < b.r0;
    });
    size_t keep = cls.size();
    if (check_num > 0 && (size_t)check_num < keep) keep = (size_t)check_num;
    for (size_t c = 0; c < keep; ++c)
        for (int64_t i = cls[c].st; i < cls[c].st + cls[c].n; ++i) out.push_back(hits[i]);
}

}  // namespace vmo

extern "C" int64_t vmo_map(const vmo_index* mi, const char* seq, int64_t len, int check_num, int mid_occ, int64_t** anchors) {
    std::vector<Anchor> a;
    map_read(mi, seq, len, check_num, mid_occ, a);
    int64_t* o = (int64_t*)malloc(sizeof(int64_t) * 4 * (a.size() ? a.size() : 1));
    for (size_t i = 0; i < a.size(); ++i) { o[4 * i] = a[i].q; o[4 * i + 1] = a[i].r; o[4 * i + 2] = a[i].s; o[4 * i + 3] = a[i].l; }
    *anchors = o;
    return (int64_t)a.size();
}
