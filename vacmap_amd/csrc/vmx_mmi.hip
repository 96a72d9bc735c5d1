// vmx_mmi.hip — minimap2 index files (`minimap2 -d`, on-disk format v3 "MMI\2"): the reference builds and reuses
// `<ref>.w<w>_k<k>.mmi` (/root/reference/src/vacmap/vacmap:324-344, index.py:26).
//
//   vm_index_load_mmi   reads such a file into the HBM-resident layout of vmx_index.hip: the minimizer SET is the file's (minimap2's own
//                       selection), positions are converted from "last base of the k-mer" to forward-strand starts on the global axis,
//                       the sequence is unpacked from the 4-bit store. Every stored minimizer is then re-derived on the GPU from the
//                       sequence (canonical k-mer, strand bit, hash64) and compared with the stored hash — a file whose hashes do not
//                       follow the published minimap2 mix is rejected, which makes this entry the cross-check of spec VMX-S1's hash /
//                       strand conventions against a real minimap2.
//   vm_index_save_mmi   writes the index in the same format (bucketed by the low `b` hash bits, singletons inline, others in the
//                       bucket's position array, 4-bit packed sequence) for tools that read `.mmi`.
// File layout (minimap2 index.c, mm_idx_dump / mm_idx_load): magic, uint32 {w, k, b, n_seq, flag}; per sequence: uint8 name length,
// name, uint32 length; per bucket (2^b): int32 n, uint64 p[n], uint32 size, size x {uint64 key, uint64 val} with
// key = (hash >> b) << 1 | singleton, val = the position word (singleton) or start << 32 | count into p;
// position word = rid << 32 | last_pos << 1 | strand; then uint32 S[(sum_len + 7) / 8], 4 bits per base (A0 C1 G2 T3, 4 = N).
// Host-side format code plus the same device finishing steps as the build; not on the per-read path.
#include "vmx_index_priv.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <sys/stat.h>

using namespace vmx;

static const char MMI_MAGIC[4] = {'M', 'M', 'I', 2};
enum { MMI_F_HPC = 1, MMI_F_NO_SEQ = 2, MMI_F_NO_NAME = 4 };

namespace {
struct Reader {
    FILE* f; int64_t left;
    bool get(void* p, size_t n) { if ((int64_t)n > left || fread(p, 1, n, f) != n) return false; left -= (int64_t)n; return true; }
};
}

static int load_mmi_impl(vm_ctx* c, const char* path, vm_index** out) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
    struct Close { FILE* f; ~Close() { fclose(f); } } cl{f};
    struct stat sb; if (fstat(fileno(f), &sb) != 0) { set_error("cannot stat index file"); return VM_ERR_IO; }
    Reader R{f, (int64_t)sb.st_size};
    char mg[4]; uint32_t x[5];
    if (!R.get(mg, 4) || memcmp(mg, MMI_MAGIC, 4) != 0 || !R.get(x, 20)) { set_error("not a minimap2 index (format v3)"); return VM_ERR_IO; }
    const int w = (int)x[0], k = (int)x[1], b = (int)x[2]; const uint32_t nseq = x[3], flag = x[4];
    if (k < 1 || k > 28 || w < 1 || w > 255 || b < 0 || b > 28 || b > 2 * k || nseq > (1u << 24)) { set_error("corrupt .mmi header"); return VM_ERR_IO; }
    if (flag & MMI_F_HPC) { set_error(".mmi built with homopolymer compression (-H) is not supported"); return VM_ERR_UNSUPPORTED; }
    if (flag & MMI_F_NO_SEQ) { set_error(".mmi without sequence (--idx-no-seq) is not supported: the extension stages need the reference bases"); return VM_ERR_UNSUPPORTED; }
    vm_index* mi = new vm_index();
    struct Guard { vm_index* m; ~Guard() { if (m) vm_index_free(m); } } g{mi};
    mi->ctx = c; mi->k = k; mi->w = w;
    int64_t off = 0;
    for (uint32_t i = 0; i < nseq; ++i) {
        uint8_t l = 0; uint32_t len = 0; char nm[256];
        if (!R.get(&l, 1) || (l && !R.get(nm, l)) || !R.get(&len, 4)) { set_error("truncated .mmi contig table"); return VM_ERR_IO; }
        mi->names.emplace_back(nm, (size_t)l); mi->lens.push_back((int64_t)len); mi->offsets.push_back(off); off += len;
    }
    mi->offsets.push_back(off);
    if (off >= (1LL << 35)) { set_error("reference longer than 2^35 bases"); return VM_ERR_UNSUPPORTED; }
    const int64_t tot = off, s_words = (tot + 7) / 8;
    if (4 * s_words > R.left) { set_error("truncated .mmi (sequence store)"); return VM_ERR_IO; }
    // buckets -> (hash, gpos << 1 | strand) pairs
    std::vector<uint64_t> hs, vs, p, ent;
    const uint64_t nb = 1ULL << b;
    auto conv = [&](uint64_t y, uint64_t& v) -> bool {
        const uint64_t rid = y >> 32; const int64_t last = (int64_t)((y & 0xffffffffULL) >> 1);
        if (rid >= nseq) return false;
        const int64_t st = last - (k - 1);
        if (st < 0 || last >= mi->lens[rid]) return false;
        v = ((uint64_t)(mi->offsets[rid] + st) << 1) | (y & 1);
        return true;
    };
    for (uint64_t bi = 0; bi < nb; ++bi) {
        int32_t n = 0; uint32_t size = 0;
        if (!R.get(&n, 4) || n < 0 || 8 * (int64_t)n > R.left) { set_error("corrupt .mmi bucket"); return VM_ERR_IO; }
        p.resize((size_t)n);
        if (n && !R.get(p.data(), 8 * (size_t)n)) { set_error("truncated .mmi bucket"); return VM_ERR_IO; }
        if (!R.get(&size, 4) || 16 * (int64_t)size > R.left) { set_error("corrupt .mmi bucket"); return VM_ERR_IO; }
        ent.resize(2 * (size_t)size);
        if (size && !R.get(ent.data(), 16 * (size_t)size)) { set_error("truncated .mmi bucket"); return VM_ERR_IO; }
        for (uint32_t e = 0; e < size; ++e) {
            const uint64_t key = ent[2 * e], val = ent[2 * e + 1];
            const uint64_t h = ((key >> 1) << b) | bi;
            uint64_t v;
            if (key & 1) {
                if (!conv(val, v)) { set_error("corrupt .mmi: minimizer position outside its contig"); return VM_ERR_IO; }
                hs.push_back(h); vs.push_back(v);
            } else {
                const uint64_t st = val >> 32, cn = val & 0xffffffffULL;
                if (st + cn > (uint64_t)n) { set_error("corrupt .mmi: bucket entry outside its position array"); return VM_ERR_IO; }
                for (uint64_t j = st; j < st + cn; ++j) { if (!conv(p[j], v)) { set_error("corrupt .mmi: minimizer position outside its contig"); return VM_ERR_IO; } hs.push_back(h); vs.push_back(v); }
            }
        }
        if ((int64_t)hs.size() >= (1LL << 32)) { set_error("index: 2^32 or more minimizers"); return VM_ERR_UNSUPPORTED; }
    }
    // 4-bit sequence store -> upper-case bases
    {
        std::vector<uint32_t> S((size_t)s_words);
        if (s_words && !R.get(S.data(), 4 * (size_t)s_words)) { set_error("truncated .mmi (sequence store)"); return VM_ERR_IO; }
        mi->bases.resize((size_t)tot);
        for (int64_t i = 0; i < tot; ++i) { const uint32_t cc = (S[(size_t)(i >> 3)] >> ((i & 7) << 2)) & 0xf; mi->bases[(size_t)i] = cc < 4 ? "ACGT"[cc] : 'N'; }
    }
    if (R.left != 0) { set_error("multi-part .mmi (built with a small -I) is not supported: build with -I larger than the reference, as the reference's driver does"); return VM_ERR_UNSUPPORTED; }
    const int64_t n = (int64_t)hs.size();
    VMX_HIP(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    std::vector<const char*> sp; for (size_t i = 0; i < mi->names.size(); ++i) sp.push_back(mi->bases.data() + mi->offsets[i]);
    VMX_TRY(vmx_index_upload_codes(mi, sp.data()));
    VMX_TRY(upload(mi->d_off, mi->offsets.data(), mi->offsets.size(), st));
    DevBuf d_h0, d_v0, d_h1, d_v1, d_tmp, d_keys, d_err;
    struct Rel { DevBuf* b[7]; ~Rel() { for (auto* q : b) q->release(); } } rel{{&d_h0, &d_v0, &d_h1, &d_v1, &d_tmp, &d_keys, &d_err}};
    VMX_TRY(mi->d_pos.reserve(8 * (size_t)n + 8));
    VMX_TRY(d_h1.reserve(8 * (size_t)std::max<int64_t>(n, 1)));
    if (n) {
        VMX_TRY(upload(d_h0, hs.data(), (size_t)n, st)); VMX_TRY(upload(d_v0, vs.data(), (size_t)n, st));
        VMX_TRY(d_v1.reserve(8 * (size_t)n));
        // order by (hash, position): stable radix sort by position (37 bits) first, by the 2k hash bits second
        size_t tb = 0, tb2 = 0;
        VMX_PRIM(vmx_prim_sort_pairs_u64(nullptr, &tb, d_v0.as<uint64_t>(), d_v1.as<uint64_t>(), d_h0.as<uint64_t>(), d_h1.as<uint64_t>(), (size_t)n, 37, st));
        VMX_PRIM(vmx_prim_sort_pairs_u64(nullptr, &tb2, d_h1.as<uint64_t>(), d_h0.as<uint64_t>(), d_v1.as<uint64_t>(), mi->d_pos.as<uint64_t>(), (size_t)n, 2 * k, st));
        VMX_TRY(d_tmp.reserve(std::max(tb, tb2) + 256));
        VMX_PRIM(vmx_prim_sort_pairs_u64(d_tmp.p, &tb, d_v0.as<uint64_t>(), d_v1.as<uint64_t>(), d_h0.as<uint64_t>(), d_h1.as<uint64_t>(), (size_t)n, 37, st));
        VMX_PRIM(vmx_prim_sort_pairs_u64(d_tmp.p, &tb2, d_h1.as<uint64_t>(), d_h0.as<uint64_t>(), d_v1.as<uint64_t>(), mi->d_pos.as<uint64_t>(), (size_t)n, 2 * k, st));
        // d_h0 = the file's hashes in final order; d_keys = the hashes re-derived from the sequence
        VMX_TRY(d_keys.reserve(8 * (size_t)n)); VMX_TRY(d_err.reserve(64)); VMX_HIP(hipMemsetAsync(d_err.p, 0, 8, st));
        hipLaunchKernelGGL(k_idx_pos_keys, dim3(grid1d(n)), dim3(256), 0, st, mi->d_codes.as<uint8_t>(), mi->d_off.as<int64_t>(), (int)nseq, mi->d_pos.as<uint64_t>(), n, k,
                           d_keys.as<uint64_t>(), d_err.as<int32_t>());
        hipLaunchKernelGGL(k_idx_check_sorted, dim3(grid1d(n)), dim3(256), 0, st, d_h0.as<uint64_t>(), mi->d_pos.as<uint64_t>(), n, d_err.as<int32_t>() + 1);
        std::vector<uint64_t> a((size_t)n), bq((size_t)n);
        int32_t err[2] = {0, 0};
        VMX_TRY(download(err, d_err.p, 2, st)); VMX_TRY(download(a.data(), d_h0.p, (size_t)n, st)); VMX_TRY(download(bq.data(), d_keys.p, (size_t)n, st));
        VMX_HIP(hipStreamSynchronize(st));
        int64_t bad = 0; for (int64_t i = 0; i < n; ++i) bad += a[(size_t)i] != bq[(size_t)i];
        if (err[0] || err[1] || bad) {
            set_error(".mmi rejected: " + std::to_string(err[0]) + " positions are not canonical k-mer starts of the stored sequence, " + std::to_string(err[1]) +
                      " duplicates, " + std::to_string(bad) + " stored hashes differ from hash64 of the k-mer");
            return VM_ERR_IO;
        }
    }
    hs.clear(); hs.shrink_to_fit(); vs.clear(); vs.shrink_to_fit();
    VMX_TRY(vmx_index_finish_device(mi, d_keys.p ? d_keys : d_h1, n));
    *out = mi; g.m = nullptr;
    return VM_OK;
}

static int save_mmi_impl(const vm_index* mi, const char* path, int b) {
    vm_ctx* c = mi->ctx;
    const int k = mi->k;
    if (b < 0) b = 14;
    if (b > 2 * k) b = 2 * k;
    if (b > 28) { set_error("bucket bits > 28"); return VM_ERR_ARG; }
    for (int64_t ln : mi->lens) if (ln >= (1LL << 31)) { set_error(".mmi stores 32-bit contig positions: a contig of 2^31 bases or more cannot be written"); return VM_ERR_UNSUPPORTED; }
    for (auto& nm : mi->names) if (nm.size() > 255) { set_error(".mmi stores 8-bit name lengths"); return VM_ERR_UNSUPPORTED; }
    uint64_t *hh = nullptr, *pp = nullptr; int64_t n = 0;
    VMX_TRY(vm_index_minimizers(mi, &hh, &pp, &n));
    struct Free { uint64_t *a, *b; ~Free() { free(a); free(b); } } fr{hh, pp};
    const int64_t tot = mi->offsets.back();
    std::string dec;
    if (!mi->has_host_seq) {
        dec.resize((size_t)tot);
        for (size_t i = 0; i < mi->names.size(); ++i) if (mi->lens[i] > 0 && vm_index_seq(mi, (int)i, 0, mi->lens[i], &dec[(size_t)mi->offsets[i]]) < 0) return VM_ERR_HIP;
    }
    const std::string& bases = mi->has_host_seq ? mi->bases : dec;
    // bucket sort by the low b hash bits (stable: inside a bucket the pairs stay ordered by hash, then position)
    const uint64_t nb = 1ULL << b, bmask = nb - 1;
    std::vector<uint64_t> bstart(nb + 1, 0);
    for (int64_t i = 0; i < n; ++i) bstart[(hh[i] & bmask) + 1]++;
    for (uint64_t i = 0; i < nb; ++i) bstart[i + 1] += bstart[i];
    std::vector<uint64_t> bh((size_t)n), by((size_t)n);
    {
        std::vector<uint64_t> cur(bstart.begin(), bstart.end() - 1);
        for (int64_t i = 0; i < n; ++i) {
            const uint64_t g = pp[i] >> 1;                           // global start -> (rid, last base of the k-mer)
            const size_t rid = (size_t)(std::upper_bound(mi->offsets.begin(), mi->offsets.end(), (int64_t)g) - mi->offsets.begin()) - 1;
            const uint64_t last = g - (uint64_t)mi->offsets[rid] + (uint64_t)(k - 1);
            const uint64_t slot = cur[hh[i] & bmask]++;
            bh[slot] = hh[i] >> b; by[slot] = ((uint64_t)rid << 32) | (last << 1) | (pp[i] & 1);
        }
    }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return VM_ERR_IO; }
    uint32_t x[5] = {(uint32_t)mi->w, (uint32_t)k, (uint32_t)b, (uint32_t)mi->names.size(), 0};
    fwrite(MMI_MAGIC, 1, 4, f); fwrite(x, 4, 5, f);
    for (size_t i = 0; i < mi->names.size(); ++i) { const uint8_t l = (uint8_t)mi->names[i].size(); const uint32_t len = (uint32_t)mi->lens[i]; fwrite(&l, 1, 1, f); fwrite(mi->names[i].data(), 1, l, f); fwrite(&len, 4, 1, f); }
    std::vector<uint64_t> p, ent;
    for (uint64_t bi = 0; bi < nb; ++bi) {
        p.clear(); ent.clear();
        for (uint64_t i = bstart[bi]; i < bstart[bi + 1];) {
            uint64_t j = i; while (j < bstart[bi + 1] && bh[j] == bh[i]) ++j;
            if (j - i == 1) { ent.push_back((bh[i] << 1) | 1); ent.push_back(by[i]); }
            else { ent.push_back(bh[i] << 1); ent.push_back(((uint64_t)p.size() << 32) | (j - i)); for (uint64_t t = i; t < j; ++t) p.push_back(by[t]); }
            i = j;
        }
        const int32_t pn = (int32_t)p.size(); const uint32_t size = (uint32_t)(ent.size() / 2);
        fwrite(&pn, 4, 1, f); if (pn) fwrite(p.data(), 8, (size_t)pn, f);
        fwrite(&size, 4, 1, f); if (size) fwrite(ent.data(), 8, ent.size(), f);
    }
    std::vector<uint32_t> S((size_t)((tot + 7) / 8), 0);
    for (int64_t i = 0; i < tot; ++i) S[(size_t)(i >> 3)] |= (uint32_t)vmx_code((uint8_t)bases[(size_t)i]) << ((i & 7) << 2);
    if (!S.empty()) fwrite(S.data(), 4, S.size(), f);
    bool ok = !ferror(f); ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("write error"); return VM_ERR_IO; }
    (void)c;
    return VM_OK;
}

extern "C" {

int vm_index_load_mmi(vm_ctx* c, const char* path, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    try { return load_mmi_impl(c, path, out); }
    catch (const std::bad_alloc&) { set_error(".mmi load: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string(".mmi load: ") + e.what()); return VM_ERR_IO; }
}

int vm_index_save_mmi(const vm_index* mi, const char* path, int bucket_bits) {
    try { return save_mmi_impl(mi, path, bucket_bits); }
    catch (const std::bad_alloc&) { set_error(".mmi save: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string(".mmi save: ") + e.what()); return VM_ERR_IO; }
}

}  // extern "C"
