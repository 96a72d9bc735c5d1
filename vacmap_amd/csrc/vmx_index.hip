// vmx_index.hip — the minimizer index: host-side build (not on the timed path), HBM-resident layout, save/load, and the
// seed-stage entry points vm_sketch_batch / vm_map_batch (kernels in k_seed.hip).
//
// Replaces `mp.Aligner(path, w=, k=)` and its accessors `.k`, `.seq_offset`, `.seq(name)` (/root/reference/src/vacmap/vacmap:344-367,
// mammap_clrnano.py:24024, :24098). Layout in HBM (replicated per GPU):
//   codes[total_len]          1 byte per base, A0 C1 G2 T3 other 4, contigs concatenated on one global axis
//   positions[n_minimizers]   uint64 (global pos << 1 | strand), grouped by hash, ascending inside a group
//   table[2^bits]             16-byte slots {hash, start, count}, open addressing (load factor <= 0.5)
//   offsets[nseq+1]           global start of every contig
#include "vmx_host.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

using namespace vmx;

struct vm_index {
    vm_ctx* ctx = nullptr;
    int k = 0, w = 0, mid_occ = 10, table_bits = 0;
    std::vector<std::string> names;
    std::vector<int64_t> lens, offsets;     // offsets has nseq+1 entries
    std::string bases;                      // upper-case concatenation (host copy for Aligner.seq)
    std::vector<uint64_t> hashes, positions;
    int64_t n_distinct = 0;
    DevBuf d_codes, d_pos, d_table, d_off;
    bool has_host_seq = true;
};

__global__ void k_sketch(const uint8_t* codes, const int64_t* roff, int n_reads, int k, int w, uint64_t* mz_hash, uint32_t* mz_ps,
                         const int64_t* mz_off, int32_t* mz_cnt);
__global__ void k_lookup(const uint64_t* mz_hash, const int64_t* mz_off, const int32_t* mz_cnt, int n_reads, const vmx_slot* tab, int bits,
                         int mid_occ, uint32_t* m_start, uint32_t* m_cnt, uint32_t* m_hoff, int64_t* nhits);
__global__ void k_fill_hits(const uint32_t* mz_ps, const int64_t* mz_off, const int32_t* mz_cnt, int n_reads, const uint32_t* m_start,
                            const uint32_t* m_cnt, const uint32_t* m_hoff, const uint64_t* idx_pos, uint64_t* keys, const int64_t* key_off,
                            const int64_t* nhits);
__global__ void k_cluster(uint64_t* keys, uint64_t* cl_keys, const int64_t* key_off, const int64_t* nhits, int n_reads, int check_num, int kmer,
                          int64_t* rows, int32_t* n_anchors);
__global__ void k_scan_i64(const int64_t* in, int64_t* out, int64_t n, int pow2_round);

static inline uint64_t h_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask; key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8)) & mask; key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask; key = key ^ key >> 28; key = (key + (key << 31)) & mask;
    return key;
}

// host sketch of one contig (spec VMX-S1), appends (hash, gpos<<1|strand)
static void host_sketch(const char* s, int64_t len, int64_t goff, int k, int w, std::vector<std::pair<uint64_t, uint64_t>>& out) {
    int64_t P = len - k + 1;
    if (P <= 0) return;
    const uint64_t mask = (1ULL << (2 * k)) - 1, INF = ~0ULL;
    const int shift = 2 * (k - 1);
    std::vector<uint64_t> h((size_t)P, INF); std::vector<uint8_t> z((size_t)P, 0);
    uint64_t fwd = 0, rc = 0; int l = 0;
    for (int64_t i = 0; i < len; ++i) {
        int c = vmx_code((uint8_t)s[i]);
        if (c < 4) { fwd = ((fwd << 2) | (uint64_t)c) & mask; rc = (rc >> 2) | ((uint64_t)(3 - c) << shift); ++l; } else l = 0;
        if (l >= k && fwd != rc) { int64_t p = i - k + 1; z[p] = rc < fwd; h[p] = h_hash64(fwd < rc ? fwd : rc, mask); }
    }
    int64_t nwin = P >= w ? P - w + 1 : 1, wl = P >= w ? w : P;
    // sliding-window minimum via the "last position where window min changes" trick is unnecessary: w is small
    std::vector<uint64_t> wmin((size_t)nwin);
    for (int64_t a = 0; a < nwin; ++a) { uint64_t m = INF; for (int64_t j = a; j < a + wl; ++j) m = h[j] < m ? h[j] : m; wmin[a] = m; }
    for (int64_t p = 0; p < P; ++p) {
        if (h[p] == INF) continue;
        int64_t a0 = p - wl + 1 < 0 ? 0 : p - wl + 1, a1 = p > nwin - 1 ? nwin - 1 : p;
        for (int64_t a = a0; a <= a1; ++a) if (wmin[a] == h[p]) { out.emplace_back(h[p], ((uint64_t)(goff + p) << 1) | z[p]); break; }
    }
}

static int index_finish_upload(vm_index* mi) {
    vm_ctx* c = mi->ctx;
    const int64_t n = (int64_t)mi->hashes.size();
    // distinct keys, occurrence cap (max(10, count at the (1 - 2e-4) quantile + 1)), hash table
    std::vector<uint64_t> dk; std::vector<uint32_t> ds, dc;
    for (int64_t i = 0; i < n;) { int64_t j = i; while (j < n && mi->hashes[j] == mi->hashes[i]) ++j; dk.push_back(mi->hashes[i]); ds.push_back((uint32_t)i); dc.push_back((uint32_t)(j - i)); i = j; }
    mi->n_distinct = (int64_t)dk.size();
    int occ = 10;
    if (!dk.empty()) {
        std::vector<uint32_t> cnt = dc;
        size_t kth = (size_t)((1.0 - 2e-4) * (double)cnt.size()); if (kth >= cnt.size()) kth = cnt.size() - 1;
        std::nth_element(cnt.begin(), cnt.begin() + kth, cnt.end());
        occ = std::max(occ, (int)cnt[kth] + 1);
    }
    mi->mid_occ = occ;
    int bits = 4; while ((1ULL << bits) < 2 * dk.size() + 1) ++bits;
    mi->table_bits = bits;
    std::vector<vmx_slot> tab((size_t)1 << bits, vmx_slot{~0ULL, 0, 0});
    const uint64_t m = ((uint64_t)1 << bits) - 1;
    for (size_t d = 0; d < dk.size(); ++d) {
        uint64_t i = (dk[d] * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
        while (tab[i].key != ~0ULL) i = (i + 1) & m;
        tab[i] = vmx_slot{dk[d], ds[d], dc[d]};
    }
    // upload
    const int64_t tot = mi->offsets.back();
    std::vector<uint8_t> codes((size_t)tot + 64, 4);
    for (int64_t i = 0; i < tot; ++i) codes[i] = vmx_code((uint8_t)mi->bases[i]);
    VMX_TRY(upload(mi->d_codes, codes.data(), codes.size(), c->stream));
    VMX_TRY(upload(mi->d_pos, mi->positions.data(), (size_t)n, c->stream));
    VMX_TRY(upload(mi->d_table, tab.data(), tab.size(), c->stream));
    VMX_TRY(upload(mi->d_off, mi->offsets.data(), mi->offsets.size(), c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" {

int vm_index_build_mem(vm_ctx* c, int nseq, const char* const* names, const char* const* seqs, const int64_t* lens, int k, int w, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (k < 1 || k > 28 || w < 1 || w > 255) { set_error("k must be in [1,28], w in [1,255]"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(c->device));
    vm_index* mi = new vm_index();
    mi->ctx = c; mi->k = k; mi->w = w;
    int64_t off = 0;
    for (int i = 0; i < nseq; ++i) { mi->names.emplace_back(names[i]); mi->lens.push_back(lens[i]); mi->offsets.push_back(off); off += lens[i]; }
    mi->offsets.push_back(off);
    if (off >= (1LL << 35)) { delete mi; set_error("reference longer than 2^35 bases"); return VM_ERR_UNSUPPORTED; }
    mi->bases.resize((size_t)off);
    for (int i = 0; i < nseq; ++i) { char* d = &mi->bases[(size_t)mi->offsets[i]]; for (int64_t x = 0; x < lens[i]; ++x) { char ch = seqs[i][x]; d[x] = (ch >= 'a' && ch <= 'z') ? ch - 32 : ch; } }
    // sketch contigs in parallel chunks (host threads; index build is outside the timed path)
    struct Job { int contig; int64_t st, en; };
    std::vector<Job> jobs;
    const int64_t CH = 4 << 20;
    for (int i = 0; i < nseq; ++i) for (int64_t s = 0; s < std::max<int64_t>(lens[i] - k + 1, 0); s += CH) jobs.push_back(Job{i, s, std::min<int64_t>(s + CH, lens[i] - k + 1)});
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> parts(jobs.size());
    std::atomic<size_t> next(0);
    auto work = [&]() {
        while (true) {
            size_t j = next.fetch_add(1); if (j >= jobs.size()) break;
            const Job& jb = jobs[j];
            // a chunk of k-mer starts [st,en) needs w-1 starts of context on both sides to decide its window minima
            int64_t lo = std::max<int64_t>(jb.st - (w - 1), 0), hi = std::min<int64_t>(jb.en + (w - 1), lens[jb.contig] - k + 1);
            std::vector<std::pair<uint64_t, uint64_t>> tmp;
            const int64_t goff = mi->offsets[jb.contig];
            // sketch the padded piece as if it were a sequence; drop selections outside [st,en) and selections that used a
            // clipped window: windows are clipped only at true contig ends because the padding is w-1 on each side
            host_sketch(mi->bases.data() + goff + lo, (hi - lo) + k - 1, goff + lo, k, w, tmp);
            for (auto& e : tmp) { int64_t p = (int64_t)(e.second >> 1) - goff; if (p >= jb.st && p < jb.en) parts[j].push_back(e); }
        }
    };
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 32) nt = 32;
    if (jobs.size() < nt) nt = (unsigned)std::max<size_t>(jobs.size(), 1);
    { std::vector<std::thread> th; for (unsigned t = 0; t < nt; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
    std::vector<std::pair<uint64_t, uint64_t>> all;
    size_t tot = 0; for (auto& p : parts) tot += p.size();
    all.reserve(tot);
    for (auto& p : parts) { all.insert(all.end(), p.begin(), p.end()); std::vector<std::pair<uint64_t, uint64_t>>().swap(p); }
    std::sort(all.begin(), all.end());
    mi->hashes.resize(all.size()); mi->positions.resize(all.size());
    for (size_t i = 0; i < all.size(); ++i) { mi->hashes[i] = all[i].first; mi->positions[i] = all[i].second; }
    std::vector<std::pair<uint64_t, uint64_t>>().swap(all);
    int rc = index_finish_upload(mi);
    if (rc < 0) { delete mi; return rc; }
    *out = mi;
    return VM_OK;
}

int vm_index_build_fasta(vm_ctx* c, const char* path, int k, int w, vm_index** out) {
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
    std::vector<std::string> names, seqs; std::string cur; char buf[1 << 16];
    auto flush_line = [&](const std::string& ln) {
        if (ln.empty()) return;
        if (ln[0] == '>') { std::string nm = ln.substr(1); size_t e = nm.find_first_of(" \t"); if (e != std::string::npos) nm.resize(e); names.push_back(nm); seqs.emplace_back(); }
        else if (!seqs.empty()) seqs.back() += ln;
    };
    while (fgets(buf, sizeof buf, f)) {
        size_t n = strlen(buf); bool eol = n && buf[n - 1] == '\n';
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
        cur.append(buf, n);
        if (eol) { flush_line(cur); cur.clear(); }
    }
    flush_line(cur); fclose(f);
    std::vector<const char*> np, sp; std::vector<int64_t> ls;
    for (size_t i = 0; i < names.size(); ++i) { np.push_back(names[i].c_str()); sp.push_back(seqs[i].data()); ls.push_back((int64_t)seqs[i].size()); }
    return vm_index_build_mem(c, (int)names.size(), np.data(), sp.data(), ls.data(), k, w, out);
}

void vm_index_free(vm_index* mi) {
    if (!mi) return;
    mi->d_codes.release(); mi->d_pos.release(); mi->d_table.release(); mi->d_off.release();
    delete mi;
}
int vm_index_k(const vm_index* mi) { return mi->k; }
int vm_index_w(const vm_index* mi) { return mi->w; }
int vm_index_nseq(const vm_index* mi) { return (int)mi->names.size(); }
int vm_index_mid_occ(const vm_index* mi) { return mi->mid_occ; }
int64_t vm_index_n_minimizers(const vm_index* mi) { return (int64_t)mi->positions.size(); }
int vm_index_seq_info(const vm_index* mi, int i, const char** name, int64_t* len, int64_t* offset) {
    if (i < 0 || i >= (int)mi->names.size()) return VM_ERR_ARG;
    if (name) *name = mi->names[i].c_str(); if (len) *len = mi->lens[i]; if (offset) *offset = mi->offsets[i];
    return VM_OK;
}
int64_t vm_index_seq(const vm_index* mi, int i, int64_t st, int64_t en, char* out) {
    if (i < 0 || i >= (int)mi->names.size() || !mi->has_host_seq) return VM_ERR_ARG;
    if (st < 0) st = 0; if (en > mi->lens[i]) en = mi->lens[i];
    if (en <= st) return 0;
    memcpy(out, mi->bases.data() + mi->offsets[i] + st, (size_t)(en - st));
    return en - st;
}
int vm_index_minimizers(const vm_index* mi, uint64_t** hashes, uint64_t** positions, int64_t* n) {
    *n = (int64_t)mi->positions.size();
    *hashes = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(*n, 1));
    *positions = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(*n, 1));
    memcpy(*hashes, mi->hashes.data(), sizeof(uint64_t) * (size_t)*n);
    // positions come back FROM THE DEVICE (what the kernels read)
    if (hipMemcpy(*positions, mi->d_pos.p, sizeof(uint64_t) * (size_t)*n, hipMemcpyDeviceToHost) != hipSuccess) return VM_ERR_HIP;
    return VM_OK;
}

// ---- save / load: own format "<ref>.w<w>_k<k>.vmx"
static const char VMX_MAGIC[8] = {'V', 'M', 'X', 'I', 'D', 'X', '0', '1'};
int vm_index_save(const vm_index* mi, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return VM_ERR_IO; }
    int64_t hdr[6] = {mi->k, mi->w, (int64_t)mi->names.size(), (int64_t)mi->positions.size(), mi->offsets.back(), mi->mid_occ};
    fwrite(VMX_MAGIC, 1, 8, f); fwrite(hdr, 8, 6, f);
    for (size_t i = 0; i < mi->names.size(); ++i) { int64_t nl = (int64_t)mi->names[i].size(); fwrite(&nl, 8, 1, f); fwrite(mi->names[i].data(), 1, (size_t)nl, f); fwrite(&mi->lens[i], 8, 1, f); }
    fwrite(mi->bases.data(), 1, mi->bases.size(), f);
    fwrite(mi->hashes.data(), 8, mi->hashes.size(), f); fwrite(mi->positions.data(), 8, mi->positions.size(), f);
    bool ok = !ferror(f); fclose(f);
    if (!ok) { set_error("write error"); return VM_ERR_IO; }
    return VM_OK;
}
int vm_index_load(vm_ctx* c, const char* path, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
    char mg[8]; int64_t hdr[6];
    if (fread(mg, 1, 8, f) != 8 || memcmp(mg, VMX_MAGIC, 8) || fread(hdr, 8, 6, f) != 6) { fclose(f); set_error("not a .vmx index"); return VM_ERR_IO; }
    vm_index* mi = new vm_index(); mi->ctx = c; mi->k = (int)hdr[0]; mi->w = (int)hdr[1];
    int64_t off = 0; bool ok = true;
    for (int64_t i = 0; i < hdr[2] && ok; ++i) {
        int64_t nl = 0, ln = 0; ok = fread(&nl, 8, 1, f) == 1 && nl >= 0 && nl < 65536;
        std::string nm((size_t)(ok ? nl : 0), ' '); ok = ok && fread(&nm[0], 1, (size_t)nl, f) == (size_t)nl && fread(&ln, 8, 1, f) == 1;
        mi->names.push_back(nm); mi->lens.push_back(ln); mi->offsets.push_back(off); off += ln;
    }
    mi->offsets.push_back(off);
    ok = ok && off == hdr[4];
    if (ok) { mi->bases.resize((size_t)off); ok = fread(&mi->bases[0], 1, (size_t)off, f) == (size_t)off; }
    if (ok) { mi->hashes.resize((size_t)hdr[3]); mi->positions.resize((size_t)hdr[3]); ok = fread(mi->hashes.data(), 8, (size_t)hdr[3], f) == (size_t)hdr[3] && fread(mi->positions.data(), 8, (size_t)hdr[3], f) == (size_t)hdr[3]; }
    fclose(f);
    if (!ok) { delete mi; set_error("truncated .vmx index"); return VM_ERR_IO; }
    VMX_HIP(hipSetDevice(c->device));
    int rc = index_finish_upload(mi);
    if (rc < 0) { delete mi; return rc; }
    *out = mi;
    return VM_OK;
}

// ---- pieces of the HBM-resident index for the multi-GPU broadcast (RCCL over xGMI; bench.py / driver)
int vm_index_blob_count(const vm_index*) { return 4; }
int vm_index_blob(const vm_index* mi, int i, void** dev_ptr, int64_t* bytes) {
    const DevBuf* b[4] = {&mi->d_codes, &mi->d_pos, &mi->d_table, &mi->d_off};
    int64_t sz[4] = {mi->offsets.back() + 64, (int64_t)mi->positions.size() * 8, (int64_t)sizeof(vmx_slot) << mi->table_bits, (int64_t)mi->offsets.size() * 8};
    if (i < 0 || i > 3) return VM_ERR_ARG;
    *dev_ptr = b[i]->p; *bytes = sz[i];
    return VM_OK;
}
// metadata needed to allocate an empty replica on another GPU/process: fixed header + names + lens
int vm_index_meta_size(const vm_index* mi, int64_t* bytes) {
    int64_t n = 8 * 8; for (auto& s : mi->names) n += 16 + (int64_t)s.size();
    *bytes = n; return VM_OK;
}
int vm_index_meta_get(const vm_index* mi, void* buf, int64_t bytes) {
    int64_t need; vm_index_meta_size(mi, &need); if (bytes < need) return VM_ERR_ARG;
    char* p = (char*)buf;
    int64_t hdr[8] = {mi->k, mi->w, (int64_t)mi->names.size(), (int64_t)mi->positions.size(), mi->offsets.back(), mi->mid_occ, mi->table_bits, mi->n_distinct};
    memcpy(p, hdr, 64); p += 64;
    for (size_t i = 0; i < mi->names.size(); ++i) { int64_t nl = (int64_t)mi->names[i].size(); memcpy(p, &nl, 8); p += 8; memcpy(p, &mi->lens[i], 8); p += 8; memcpy(p, mi->names[i].data(), (size_t)nl); p += nl; }
    return VM_OK;
}
int vm_index_from_meta(vm_ctx* c, const void* buf, int64_t bytes, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (bytes < 64) return VM_ERR_ARG;
    const char* p = (const char*)buf; int64_t hdr[8]; memcpy(hdr, p, 64); p += 64;
    vm_index* mi = new vm_index(); mi->ctx = c; mi->k = (int)hdr[0]; mi->w = (int)hdr[1]; mi->mid_occ = (int)hdr[5]; mi->table_bits = (int)hdr[6]; mi->n_distinct = hdr[7];
    mi->has_host_seq = false;
    int64_t off = 0;
    for (int64_t i = 0; i < hdr[2]; ++i) { int64_t nl, ln; memcpy(&nl, p, 8); p += 8; memcpy(&ln, p, 8); p += 8; mi->names.emplace_back(p, (size_t)nl); p += nl; mi->lens.push_back(ln); mi->offsets.push_back(off); off += ln; }
    mi->offsets.push_back(off);
    mi->positions.resize(0);
    VMX_HIP(hipSetDevice(c->device));
    int rc = 0;
    if ((rc = mi->d_codes.reserve((size_t)off + 64)) < 0 || (rc = mi->d_pos.reserve((size_t)hdr[3] * 8 + 8)) < 0 ||
        (rc = mi->d_table.reserve(sizeof(vmx_slot) << mi->table_bits)) < 0 || (rc = mi->d_off.reserve(mi->offsets.size() * 8)) < 0) { delete mi; return rc; }
    mi->hashes.clear();
    mi->positions.assign(0, 0);
    // remember the minimizer count for vm_index_n_minimizers / blob sizes
    mi->positions.shrink_to_fit();
    mi->positions.resize((size_t)hdr[3]);   // host shadow (zeros): only its size is used on replicas
    *out = mi;
    return VM_OK;
}

// ------------------------------------------------------------------------------------------------ seed stage entries
static int upload_reads(vm_ctx* c, int64_t n, const char* seqs, const int64_t* off, DevBuf& raw, DevBuf& codes, DevBuf& doff) {
    const int64_t tot = off[n];
    VMX_TRY(upload(raw, seqs, (size_t)tot, c->stream));
    VMX_TRY(codes.reserve((size_t)tot + 64));
    VMX_TRY(upload(doff, off, (size_t)n + 1, c->stream));
    if (tot) hipLaunchKernelGGL(k_encode, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 4096)), dim3(256), 0, c->stream, raw.as<char>(), codes.as<uint8_t>(), tot);
    return 0;
}

int vm_sketch_batch(vm_ctx* c, int k, int w, int64_t n, const char* seqs, const int64_t* off, uint64_t** hash, int32_t** pos, int8_t** strand, int64_t** ooff) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (k < 1 || k > 28 || w < 1 || w > 255) { set_error("bad k/w"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_reads(c, n, seqs, off, c->b[0], c->b[1], c->b[2]));
    const int64_t tot = off[n];
    VMX_TRY(c->b[3].reserve(8 * (size_t)(tot + 1))); VMX_TRY(c->b[4].reserve(4 * (size_t)(tot + 1))); VMX_TRY(c->b[5].reserve(4 * (size_t)(n + 1)));
    if (n) hipLaunchKernelGGL(k_sketch, dim3((unsigned)std::min<int64_t>(n, c->num_cu * 4)), dim3(256), 0, c->stream, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(),
                              (int)n, k, w, c->b[3].as<uint64_t>(), c->b[4].as<uint32_t>(), c->b[2].as<int64_t>(), c->b[5].as<int32_t>());
    std::vector<int32_t> cnt((size_t)n); std::vector<uint64_t> hh((size_t)tot); std::vector<uint32_t> ps((size_t)tot);
    VMX_TRY(download(cnt.data(), c->b[5].p, (size_t)n, c->stream)); VMX_TRY(download(hh.data(), c->b[3].p, (size_t)tot, c->stream));
    VMX_TRY(download(ps.data(), c->b[4].p, (size_t)tot, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream)); VMX_HIP(hipGetLastError());
    int64_t m = 0; for (int64_t r = 0; r < n; ++r) m += cnt[r];
    *hash = (uint64_t*)malloc(8 * (size_t)std::max<int64_t>(m, 1)); *pos = (int32_t*)malloc(4 * (size_t)std::max<int64_t>(m, 1));
    *strand = (int8_t*)malloc((size_t)std::max<int64_t>(m, 1)); *ooff = (int64_t*)malloc(8 * (size_t)(n + 1));
    int64_t o = 0;
    for (int64_t r = 0; r < n; ++r) { (*ooff)[r] = o; for (int x = 0; x < cnt[r]; ++x) { (*hash)[o] = hh[off[r] + x]; (*pos)[o] = (int32_t)(ps[off[r] + x] >> 1); (*strand)[o] = (int8_t)(ps[off[r] + x] & 1); ++o; } }
    (*ooff)[n] = o;
    return VM_OK;
}

// device-side seed stage shared by vm_map_batch and vm_align_batch: codes/roff already on the device.
// leaves rows (int64 x4) at key_off[r] with n_anchors[r] valid rows; returns host copies of key_off / nhits totals
int vmx_seed_stage(vm_ctx* c, const vm_index* mi, int check_num, int mid_occ, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, int64_t total_bases,
                   DevBuf* B /* >= 12 buffers */, std::vector<int64_t>& h_koff, std::vector<int64_t>& h_nhits) {
    if (mid_occ <= 0) mid_occ = mi->mid_occ;
    DevBuf &mzh = B[0], &mzp = B[1], &mzc = B[2], &mst = B[3], &mcn = B[4], &mho = B[5], &nh = B[6], &koff = B[7], &keys = B[8], &ckeys = B[9], &rows = B[10], &nanc = B[11];
    VMX_TRY(mzh.reserve(8 * (size_t)(total_bases + 1))); VMX_TRY(mzp.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mzc.reserve(4 * (size_t)(n + 1)));
    VMX_TRY(mst.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mcn.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mho.reserve(4 * (size_t)(total_bases + 1)));
    VMX_TRY(nh.reserve(8 * (size_t)(n + 2))); VMX_TRY(koff.reserve(8 * (size_t)(n + 2))); VMX_TRY(nanc.reserve(4 * (size_t)(n + 1)));
    const unsigned grid = (unsigned)std::max<int64_t>(std::min<int64_t>(n, (int64_t)c->num_cu * 4), 1);
    hipLaunchKernelGGL(k_sketch, dim3(grid), dim3(256), 0, c->stream, d_codes, d_roff, (int)n, mi->k, mi->w, mzh.as<uint64_t>(), mzp.as<uint32_t>(), d_roff, mzc.as<int32_t>());
    hipLaunchKernelGGL(k_lookup, dim3(grid), dim3(256), 0, c->stream, mzh.as<uint64_t>(), d_roff, mzc.as<int32_t>(), (int)n, mi->d_table.as<vmx_slot>(), mi->table_bits, mid_occ,
                       mst.as<uint32_t>(), mcn.as<uint32_t>(), mho.as<uint32_t>(), nh.as<int64_t>());
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, nh.as<int64_t>(), koff.as<int64_t>(), n, 1);
    h_koff.resize((size_t)n + 1); h_nhits.resize((size_t)n);
    VMX_TRY(download(h_koff.data(), koff.p, (size_t)n + 1, c->stream)); VMX_TRY(download(h_nhits.data(), nh.p, (size_t)n, c->stream));
    std::vector<int32_t> h_mzc((size_t)n);
    VMX_TRY(download(h_mzc.data(), mzc.p, (size_t)n, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));   // sizing sync #1: total (power-of-two padded) hits of the batch
    c->last_n_minimizers = 0; for (int64_t r = 0; r < n; ++r) c->last_n_minimizers += h_mzc[r];
    const int64_t ktot = h_koff[n];
    VMX_TRY(keys.reserve(8 * (size_t)(ktot + 1))); VMX_TRY(ckeys.reserve(8 * (size_t)(ktot + 1))); VMX_TRY(rows.reserve(32 * (size_t)(ktot + 1)));
    hipLaunchKernelGGL(k_fill_hits, dim3(grid), dim3(256), 0, c->stream, mzp.as<uint32_t>(), d_roff, mzc.as<int32_t>(), (int)n, mst.as<uint32_t>(), mcn.as<uint32_t>(),
                       mho.as<uint32_t>(), mi->d_pos.as<uint64_t>(), keys.as<uint64_t>(), koff.as<int64_t>(), nh.as<int64_t>());
    hipLaunchKernelGGL(k_cluster, dim3(grid), dim3(256), 0, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(), koff.as<int64_t>(), nh.as<int64_t>(), (int)n, check_num,
                       mi->k, rows.as<int64_t>(), nanc.as<int32_t>());
    return 0;
}

int vm_map_batch(vm_ctx* c, const vm_index* mi, int check_num, int mid_occ, int64_t n, const char* seqs, const int64_t* off, int64_t** anchors, int64_t** anchor_off) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_reads(c, n, seqs, off, c->b[0], c->b[1], c->b[2]));
    std::vector<int64_t> koff, nhits;
    if (n) VMX_TRY(vmx_seed_stage(c, mi, check_num, mid_occ, n, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(), off[n], &c->b[3], koff, nhits));
    else koff.assign(1, 0);
    std::vector<int32_t> nanc((size_t)n);
    std::vector<int64_t> rows((size_t)koff[n] * 4);
    VMX_TRY(download(nanc.data(), c->b[14].p, (size_t)n, c->stream));
    VMX_TRY(download(rows.data(), c->b[13].p, rows.size(), c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream)); VMX_HIP(hipGetLastError());
    int64_t tot = 0; for (int64_t r = 0; r < n; ++r) tot += nanc[r];
    *anchors = (int64_t*)malloc(32 * (size_t)std::max<int64_t>(tot, 1)); *anchor_off = (int64_t*)malloc(8 * (size_t)(n + 1));
    int64_t o = 0;
    for (int64_t r = 0; r < n; ++r) { (*anchor_off)[r] = o; memcpy(*anchors + 4 * o, rows.data() + 4 * koff[r], 32 * (size_t)nanc[r]); o += nanc[r]; }
    (*anchor_off)[n] = o;
    return VM_OK;
}

int vm_map(vm_ctx* c, const vm_index* mi, const char* seq, int64_t len, int check_num, int mid_occ, int64_t** anchors, int64_t* n) {
    int64_t off[2] = {0, len}; int64_t* ao = nullptr;
    int rc = vm_map_batch(c, mi, check_num, mid_occ, 1, seq, off, anchors, &ao);
    if (rc < 0) return rc;
    *n = ao[1]; free(ao);
    return VM_OK;
}

}  // extern "C"

#include "vmx_stage.h"
void vmx_index_view(const vm_index* mi, vm_index_view* v) {
    v->codes = mi->d_codes.as<uint8_t>(); v->coff = mi->d_off.as<int64_t>(); v->nseq = (int)mi->names.size(); v->total_len = mi->offsets.back();
    v->pos = mi->d_pos.as<uint64_t>(); v->table = mi->d_table.as<vmx_slot>(); v->table_bits = mi->table_bits; v->k = mi->k; v->w = mi->w; v->mid_occ = mi->mid_occ;
}
