// vmx_index.hip — the minimizer index: built ON THE GPU (sketch of the whole reference, device radix sort, run-length grouping,
// open-addressing table), HBM-resident layout, save/load, and the seed-stage entry points vm_sketch_batch / vm_map_batch (kernels in
// k_seed.hip).
//
// Replaces `mp.Aligner(path, w=, k=)` and its accessors `.k`, `.seq_offset`, `.seq(name)` (/root/reference/src/vacmap/vacmap:344-367,
// mammap_clrnano.py:24024, :24098). Layout in HBM (replicated per GPU):
//   codes[total_len]          1 byte per base, A0 C1 G2 T3 other 4, contigs concatenated on one global axis
//   positions[n_minimizers]   uint64 (global pos << 1 | strand), grouped by hash, ascending inside a group
//   table[2^bits]             16-byte slots {hash, start, count}, open addressing (load factor <= 0.5)
//   offsets[nseq+1]           global start of every contig
// hg38-size (3.1 Gb, k 15, w 10): 3.1 GB + 4.5 GB + 16 GB; the build itself peaks at about 45 GB of HBM and takes seconds.
#include "vmx_host.h"
#include "vmx_index_prim.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>

using namespace vmx;

#include "vmx_index_priv.h"

__global__ void k_sketch32(const uint8_t* codes, const int64_t* roff, int n_reads, int k, int w, uint64_t* mz_hash, uint32_t* mz_ps, const int64_t* mz_off, int32_t* mz_cnt);
__global__ void k_sketch(const uint8_t* codes, const int64_t* roff, int n_reads, int k, int w, uint64_t* mz_hash, uint32_t* mz_ps,
                         const int64_t* mz_off, int32_t* mz_cnt);
__global__ void k_lookup(const uint64_t* mz_hash, const int64_t* mz_off, const int32_t* mz_cnt, int n_reads, const vmx_slot* tab, int bits,
                         int mid_occ, uint32_t* m_start, uint32_t* m_cnt, uint32_t* m_hoff, int64_t* nhits);
__global__ void k_fill_hits(const uint32_t* mz_ps, const int64_t* mz_off, const int32_t* mz_cnt, int n_reads, const uint32_t* m_start,
                            const uint32_t* m_cnt, const uint32_t* m_hoff, const uint64_t* idx_pos, uint64_t* keys, const int64_t* key_off,
                            const int64_t* nhits);
__global__ void k_cluster(uint64_t* keys, uint64_t* cl_keys, const int64_t* key_off, const int64_t* nhits, const int32_t* rlist, int nlist, int tile, int check_num, int kmer,
                          int64_t* rows, int32_t* n_anchors);
__global__ void k_cluster_gen(uint64_t* keys, uint64_t* cl_keys, const int64_t* key_off, const int64_t* nhits, const int32_t* rlist, int nlist, const int32_t* nlist_dev, int tile, int check_num,
                              int kmer, int64_t* rows, int32_t* n_anchors);
#ifndef VMX_CF_BLOCK
#define VMX_CF_BLOCK 1024
#endif
__global__ void k_cluster_big(uint64_t* keys, uint64_t* cl_keys, const int64_t* key_off, const int64_t* nhits, const int32_t* rlist, int nlist, int tile, int check_num, int kmer,
                              int64_t* rows, int32_t* n_anchors, int32_t* decl, int32_t* n_decl);
__global__ void k_cluster_long(uint64_t* keys, uint64_t* cl_keys, const int64_t* key_off, const int64_t* nhits, const int32_t* rlist, int nlist, const int32_t* nlist_dev, int tile, int check_num,
                               int kmer, int64_t* rows, int32_t* n_anchors, int32_t* decl, int32_t* n_decl);
__global__ void k_scan_i64(const int64_t* in, int64_t* out, int64_t n, int pow2_round);

// ------------------------------------------------------------------------------------------------ build kernels (spec VMX-S1)
__device__ __forceinline__ uint64_t vmx_idx_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask; key = key ^ key >> 24; key = ((key + (key << 3)) + (key << 8)) & mask; key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask; key = key ^ key >> 28; key = (key + (key << 31)) & mask;
    return key;
}

#define VMX_RS_TILE 2048          // k-mer starts per job
#define VMX_RS_HALO 256           // >= w - 1 on each side
struct vmx_refjob { int64_t goff, P, t0; };   // contig's global offset, its number of k-mer starts (len - k + 1), first start of the tile

// Minimizers of one tile of one contig: the k-mer starts [t0, t0 + TILE) of the contig, with w - 1 starts of context on both sides so that
// every window (windows lie inside the contig's starts, never across contigs) is seen whole. pass 0 counts, pass 1 writes
// (hash, gpos << 1 | strand) in ascending position order at off[job]. Same selection rule as k_sketch for reads.
__global__ void __launch_bounds__(256) k_ref_sketch(const uint8_t* __restrict__ codes, const vmx_refjob* __restrict__ jobs, int64_t njobs, int k, int w, int pass,
                                                    int64_t* __restrict__ cnt, const int64_t* __restrict__ off, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
    __shared__ uint8_t s_codes[VMX_RS_TILE + 2 * VMX_RS_HALO + 64];
    __shared__ uint64_t s_h[VMX_RS_TILE + 2 * VMX_RS_HALO];
    __shared__ uint64_t s_wmin[VMX_RS_TILE + 2 * VMX_RS_HALO];
    __shared__ uint8_t s_z[VMX_RS_TILE + 2 * VMX_RS_HALO];
    __shared__ int s_scan[20];
    const uint64_t mask = (1ULL << (2 * k)) - 1, INF = ~0ULL;
    const int shift = 2 * (k - 1);
    for (int64_t jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
        const vmx_refjob J = jobs[jb];
        const uint8_t* C = codes + J.goff;
        const int64_t P = J.P, t0 = J.t0;
        const int64_t nwin = P >= w ? P - w + 1 : 1;
        const int wl = P >= w ? w : (int)P;
        const int64_t lo = t0 - (w - 1) > 0 ? t0 - (w - 1) : 0;
        int64_t hi = t0 + VMX_RS_TILE + (w - 1); if (hi > P) hi = P;
        const int npos = (int)(hi - lo);
        for (int x = (int)threadIdx.x; x < npos + k - 1; x += 256) s_codes[x] = C[lo + x];
        __syncthreads();
        for (int x = (int)threadIdx.x; x < npos; x += 256) {
            uint64_t fwd = 0, rc = 0; bool ok = true;
            for (int i = 0; i < k; ++i) {
                const uint8_t c = s_codes[x + i];
                if (c > 3) ok = false;
                fwd = (fwd << 2) | (uint64_t)(c & 3);
                rc = (rc >> 2) | ((uint64_t)(3 - (c & 3)) << shift);
            }
            uint64_t h = INF; uint8_t z = 0;
            if (ok && fwd != rc) { z = rc < fwd ? 1 : 0; h = vmx_idx_hash64(fwd < rc ? fwd : rc, mask); }
            s_h[x] = h; s_z[x] = z;
        }
        __syncthreads();
        for (int x = (int)threadIdx.x; x < npos; x += 256) {
            const int64_t a = lo + x; uint64_t m = INF;
            if (a < nwin) { for (int j = 0; j < wl; ++j) { const int y = x + j; if (y < npos) { const uint64_t v = s_h[y]; m = v < m ? v : m; } } }
            s_wmin[x] = m;
        }
        __syncthreads();
        int64_t pend = t0 + VMX_RS_TILE; if (pend > P) pend = P;
        int written = 0;
        const int64_t obase = pass ? off[jb] : 0;
        for (int64_t pb = t0; pb < pend; pb += 256) {
            const int64_t p = pb + threadIdx.x;
            int sel = 0; uint64_t h = INF; uint8_t z = 0;
            if (p < pend) {
                const int x = (int)(p - lo); h = s_h[x]; z = s_z[x];
                if (h != INF) {
                    int64_t a0 = p - wl + 1; if (a0 < 0) a0 = 0;
                    int64_t a1 = p; if (a1 > nwin - 1) a1 = nwin - 1;
                    for (int64_t a = a0; a <= a1; ++a) if (s_wmin[a - lo] == h) { sel = 1; break; }
                }
            }
            int tot; const int ex = vmx_block_excl_scan(sel, s_scan, &tot);
            if (pass && sel) { keys[obase + written + ex] = h; vals[obase + written + ex] = ((uint64_t)(J.goff + p) << 1) | z; }
            written += tot;
            __syncthreads();
        }
        if (!pass && threadIdx.x == 0) cnt[jb] = written;
        __syncthreads();
    }
}

// histogram of the occurrence counts of the distinct minimizers (bins 0..65535, bin 65536 = anything larger): small counts are
// gathered per workgroup in LDS first (almost every count is 1 or 2; global atomics on two addresses would serialise)
__global__ void __launch_bounds__(256) k_idx_occ_hist(const uint32_t* __restrict__ counts, int64_t nd, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int s_h[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) s_h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nd; i += (int64_t)gridDim.x * 256) {
        const uint32_t c = counts[i];
        if (c < 1024) atomicAdd(&s_h[c], 1u); else atomicAdd(&hist[c < 65536 ? c : 65536], 1ULL);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

// open addressing, linear probing from the golden-ratio multiplicative hash (the probe of vmx_table_find, k_seed.hip). Which key ends
// up in which slot of a collision chain depends on the insertion order; look-ups do not.
__global__ void __launch_bounds__(256) k_idx_table_insert(const uint64_t* __restrict__ uniq, const uint32_t* __restrict__ starts, const uint32_t* __restrict__ counts,
                                                          int64_t nd, vmx_slot* __restrict__ tab, int bits) {
    const uint64_t m = (1ULL << bits) - 1;
    for (int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x; d < nd; d += (int64_t)gridDim.x * 256) {
        const uint64_t key = uniq[d];
        uint64_t i = (key * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
        while (true) {
            const unsigned long long old = atomicCAS((unsigned long long*)&tab[i].key, ~0ULL, (unsigned long long)key);
            if (old == ~0ULL) { tab[i].start = starts[d]; tab[i].count = counts[d]; break; }
            i = (i + 1) & m;
        }
    }
}

// load path: recompute the hash of every stored position from the codes and validate it (k-mer inside one contig, no ambiguous base,
// strand bit = which of fwd / rc is canonical). err[0] counts violations.
__global__ void __launch_bounds__(256) k_idx_pos_keys(const uint8_t* __restrict__ codes, const int64_t* __restrict__ coff, int nseq, const uint64_t* __restrict__ pos, int64_t n,
                                                      int k, uint64_t* __restrict__ keys, int32_t* __restrict__ err) {
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    const int shift = 2 * (k - 1);
    const int64_t total = coff[nseq];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t pv = pos[i]; const int64_t p = (int64_t)(pv >> 1);
        bool ok = p >= 0 && p + k <= total;
        uint64_t key = ~0ULL;
        if (ok) {
            int lo = 0, hi = nseq;                       // contig of p: last c with coff[c] <= p
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= p) lo = mid; else hi = mid; }
            ok = p + k <= coff[lo + 1];
        }
        if (ok) {
            uint64_t fwd = 0, rc = 0;
            for (int j = 0; j < k; ++j) { const uint8_t c = codes[p + j]; if (c > 3) ok = false; fwd = (fwd << 2) | (uint64_t)(c & 3); rc = (rc >> 2) | ((uint64_t)(3 - (c & 3)) << shift); }
            if (ok && fwd != rc && (uint64_t)(rc < fwd) == (pv & 1)) key = vmx_idx_hash64(fwd < rc ? fwd : rc, mask); else ok = false;
        }
        keys[i] = key;
        if (!ok) atomicAdd(err, 1);
    }
}
// (hash, position) must be strictly ascending
__global__ void __launch_bounds__(256) k_idx_check_sorted(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ pos, int64_t n, int32_t* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * 256)
        if (keys[i] < keys[i - 1] || (keys[i] == keys[i - 1] && pos[i] <= pos[i - 1])) atomicAdd(err, 1);
}
// tests: the sorted hash column back from the table (hashes[start .. start + count) = key of every occupied slot)
__global__ void __launch_bounds__(256) k_idx_fill_hashes(const vmx_slot* __restrict__ tab, int64_t nslots, uint64_t* __restrict__ hashes) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nslots; i += (int64_t)gridDim.x * 256) {
        const vmx_slot s = tab[i];
        if (s.key == ~0ULL) continue;
        for (uint32_t e = 0; e < s.count; ++e) hashes[(int64_t)s.start + e] = s.key;
    }
}
__global__ void __launch_bounds__(256) k_decode(const uint8_t* __restrict__ codes, char* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const uint8_t c = codes[i]; out[i] = c > 3 ? 'N' : "ACGT"[c]; }
}
__global__ void __launch_bounds__(256) k_fill_u8(uint8_t* p, int64_t n, uint8_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}


// sorted (hash, position) pairs on the device -> distinct keys, occurrence cap, hash table. d_keys is consumed (released).
int vmx_index_finish_device(vm_index* mi, DevBuf& d_keys, int64_t n) {
    vm_ctx* c = mi->ctx; hipStream_t st = c->stream;
    mi->n_min = n;
    if (n >= (1LL << 32)) { set_error("index: 2^32 or more minimizers (slot start is 32 bits)"); return VM_ERR_UNSUPPORTED; }
    DevBuf d_uniq, d_counts, d_starts, d_nruns, d_tmp, d_hist;
    struct Rel { DevBuf* b[6]; ~Rel() { for (auto* x : b) x->release(); } } rel{{&d_uniq, &d_counts, &d_starts, &d_nruns, &d_tmp, &d_hist}};
    uint64_t nd = 0;
    if (n > 0) {
        VMX_TRY(d_uniq.reserve(8 * (size_t)n)); VMX_TRY(d_counts.reserve(4 * (size_t)n)); VMX_TRY(d_nruns.reserve(64));
        size_t tb = 0;
        VMX_PRIM(vmx_prim_rle_u64(nullptr, &tb, d_keys.as<uint64_t>(), (size_t)n, d_uniq.as<uint64_t>(), d_counts.as<uint32_t>(), d_nruns.as<uint64_t>(), st));
        VMX_TRY(d_tmp.reserve(tb + 256));
        VMX_PRIM(vmx_prim_rle_u64(d_tmp.p, &tb, d_keys.as<uint64_t>(), (size_t)n, d_uniq.as<uint64_t>(), d_counts.as<uint32_t>(), d_nruns.as<uint64_t>(), st));
        VMX_TRY(download(&nd, d_nruns.p, 1, st));
        VMX_HIP(hipStreamSynchronize(st));
    }
    d_keys.release();
    mi->n_distinct = (int64_t)nd;
    // occurrence cap: max(10, count at the (1 - 2e-4) quantile of the distinct minimizers + 1)  [spec VMX-S1]
    int occ = 10;
    if (nd > 0) {
        VMX_TRY(d_hist.reserve(8 * 65537)); VMX_HIP(hipMemsetAsync(d_hist.p, 0, 8 * 65537, st));
        hipLaunchKernelGGL(k_idx_occ_hist, dim3(grid1d((int64_t)nd, 2048)), dim3(256), 0, st, d_counts.as<uint32_t>(), (int64_t)nd, d_hist.as<unsigned long long>());
        std::vector<unsigned long long> hist(65537);
        VMX_TRY(download(hist.data(), d_hist.p, 65537, st));
        VMX_HIP(hipStreamSynchronize(st));
        size_t kth = (size_t)((1.0 - 2e-4) * (double)nd); if (kth >= nd) kth = (size_t)nd - 1;
        unsigned long long cum = 0; int v = -1;
        for (int b = 0; b < 65536; ++b) { cum += hist[b]; if (cum > kth) { v = b; break; } }
        if (v < 0) {    // the quantile lies among counts >= 65536: take them to the host (never seen; kept exact)
            std::vector<uint32_t> cnt((size_t)nd);
            VMX_TRY(download(cnt.data(), d_counts.p, (size_t)nd, st)); VMX_HIP(hipStreamSynchronize(st));
            std::nth_element(cnt.begin(), cnt.begin() + kth, cnt.end()); v = (int)std::min<uint32_t>(cnt[kth], 0x7ffffffeu);
        }
        occ = std::max(occ, v + 1);
    }
    mi->mid_occ = occ;
    int bits = 4; while ((1ULL << bits) < 2 * nd + 1) ++bits;
    mi->table_bits = bits;
    const int64_t nslots = (int64_t)1 << bits;
    VMX_TRY(mi->d_table.reserve(sizeof(vmx_slot) * (size_t)nslots));
    VMX_HIP(hipMemsetAsync(mi->d_table.p, 0xff, sizeof(vmx_slot) * (size_t)nslots, st));
    if (nd > 0) {
        VMX_TRY(d_starts.reserve(4 * (size_t)nd));
        size_t tb = 0;
        VMX_PRIM(vmx_prim_excl_scan_u32(nullptr, &tb, d_counts.as<uint32_t>(), d_starts.as<uint32_t>(), (size_t)nd, st));
        VMX_TRY(d_tmp.reserve(tb + 256));
        VMX_PRIM(vmx_prim_excl_scan_u32(d_tmp.p, &tb, d_counts.as<uint32_t>(), d_starts.as<uint32_t>(), (size_t)nd, st));
        hipLaunchKernelGGL(k_idx_table_insert, dim3(grid1d((int64_t)nd)), dim3(256), 0, st, d_uniq.as<uint64_t>(), d_starts.as<uint32_t>(), d_counts.as<uint32_t>(), (int64_t)nd,
                           mi->d_table.as<vmx_slot>(), bits);
    }
    VMX_TRY(upload(mi->d_off, mi->offsets.data(), mi->offsets.size(), st));
    VMX_HIP(hipStreamSynchronize(st));
    VMX_HIP(hipGetLastError());
    return 0;
}

// codes of the whole reference already in mi->d_codes: sketch every contig, sort, finish
static int index_build_device(vm_index* mi) {
    vm_ctx* c = mi->ctx; hipStream_t st = c->stream;
    const int k = mi->k, w = mi->w;
    std::vector<vmx_refjob> jobs;
    for (size_t i = 0; i < mi->lens.size(); ++i) {
        const int64_t P = mi->lens[i] - k + 1;
        for (int64_t t0 = 0; t0 < P; t0 += VMX_RS_TILE) jobs.push_back(vmx_refjob{mi->offsets[i], P, t0});
    }
    const int64_t nj = (int64_t)jobs.size();
    DevBuf d_jobs, d_cnt, d_off, d_tmp, d_k0, d_v0, d_k1;
    struct Rel { DevBuf* b[7]; ~Rel() { for (auto* x : b) x->release(); } } rel{{&d_jobs, &d_cnt, &d_off, &d_tmp, &d_k0, &d_v0, &d_k1}};
    int64_t n = 0;
    if (nj > 0) {
        VMX_TRY(upload(d_jobs, jobs.data(), (size_t)nj, st));
        VMX_TRY(d_cnt.reserve(8 * (size_t)(nj + 1))); VMX_TRY(d_off.reserve(8 * (size_t)(nj + 1)));
        VMX_HIP(hipMemsetAsync((char*)d_cnt.p + 8 * (size_t)nj, 0, 8, st));                     // scan over nj + 1 entries: off[nj] = total
        const unsigned grid = (unsigned)std::min<int64_t>(nj, (int64_t)c->num_cu * 64);
        hipLaunchKernelGGL(k_ref_sketch, dim3(grid), dim3(256), 0, st, mi->d_codes.as<uint8_t>(), d_jobs.as<vmx_refjob>(), nj, k, w, 0, d_cnt.as<int64_t>(), (const int64_t*)nullptr,
                           (uint64_t*)nullptr, (uint64_t*)nullptr);
        size_t tb = 0;
        VMX_PRIM(vmx_prim_excl_scan_i64(nullptr, &tb, d_cnt.as<int64_t>(), d_off.as<int64_t>(), (size_t)nj + 1, st));
        VMX_TRY(d_tmp.reserve(tb + 256));
        VMX_PRIM(vmx_prim_excl_scan_i64(d_tmp.p, &tb, d_cnt.as<int64_t>(), d_off.as<int64_t>(), (size_t)nj + 1, st));
        VMX_TRY(download(&n, d_off.as<int64_t>() + nj, 1, st));
        VMX_HIP(hipStreamSynchronize(st));
        if (n > 0) {
            VMX_TRY(d_k0.reserve(8 * (size_t)n)); VMX_TRY(d_v0.reserve(8 * (size_t)n)); VMX_TRY(d_k1.reserve(8 * (size_t)n)); VMX_TRY(mi->d_pos.reserve(8 * (size_t)n + 8));
            hipLaunchKernelGGL(k_ref_sketch, dim3(grid), dim3(256), 0, st, mi->d_codes.as<uint8_t>(), d_jobs.as<vmx_refjob>(), nj, k, w, 1, d_cnt.as<int64_t>(), d_off.as<int64_t>(),
                               d_k0.as<uint64_t>(), d_v0.as<uint64_t>());
            // stable LSD radix sort on the 2k hash bits: positions were emitted in ascending order, so they stay ascending inside a hash group
            VMX_PRIM(vmx_prim_sort_pairs_u64(nullptr, &tb, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint64_t>(), mi->d_pos.as<uint64_t>(), (size_t)n, 2 * k, st));
            VMX_TRY(d_tmp.reserve(tb + 256));
            VMX_PRIM(vmx_prim_sort_pairs_u64(d_tmp.p, &tb, d_k0.as<uint64_t>(), d_k1.as<uint64_t>(), d_v0.as<uint64_t>(), mi->d_pos.as<uint64_t>(), (size_t)n, 2 * k, st));
            VMX_HIP(hipStreamSynchronize(st));
            d_k0.release(); d_v0.release(); d_tmp.release();
        }
    }
    if (n == 0) VMX_TRY(mi->d_pos.reserve(8));
    return vmx_index_finish_device(mi, d_k1, n);
}

// host bases (ASCII) -> device codes, staged through a bounded device buffer; pads the tail with 64 bytes of code 4
int vmx_index_upload_codes(vm_index* mi, const char* const* seqs) {
    vm_ctx* c = mi->ctx; hipStream_t st = c->stream;
    const int64_t tot = mi->offsets.back();
    VMX_TRY(mi->d_codes.reserve((size_t)tot + 64));
    DevBuf stage; struct Rel { DevBuf* b; ~Rel() { b->release(); } } rel{&stage};
    const int64_t CH = (int64_t)256 << 20;
    VMX_TRY(stage.reserve((size_t)std::min<int64_t>(std::max<int64_t>(tot, 1), CH)));
    for (size_t i = 0; i < mi->lens.size(); ++i)
        for (int64_t s = 0; s < mi->lens[i]; s += CH) {
            const int64_t m = std::min<int64_t>(CH, mi->lens[i] - s);
            VMX_HIP(hipMemcpyAsync(stage.p, seqs[i] + s, (size_t)m, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_encode, dim3(grid1d(m, 8192)), dim3(256), 0, st, stage.as<char>(), mi->d_codes.as<uint8_t>() + mi->offsets[i] + s, m);
            VMX_HIP(hipStreamSynchronize(st));           // the staging buffer is reused
        }
    hipLaunchKernelGGL(k_fill_u8, dim3(1), dim3(256), 0, st, mi->d_codes.as<uint8_t>() + tot, (int64_t)64, (uint8_t)4);
    return 0;
}

static void parallel_for_chunks(int64_t n, int64_t chunk, const std::function<void(int64_t, int64_t)>& fn) {
    const int64_t nch = (n + chunk - 1) / chunk;
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 16) nt = 16;
    if ((int64_t)nt > nch) nt = (unsigned)std::max<int64_t>(nch, 1);
    std::atomic<int64_t> next(0);
    auto work = [&]() { while (true) { const int64_t j = next.fetch_add(1); if (j >= nch) break; fn(j * chunk, std::min<int64_t>(n, (j + 1) * chunk)); } };
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> th; for (unsigned t = 0; t < nt; ++t) th.emplace_back(work); for (auto& t : th) t.join();
}

static int index_build_mem_impl(vm_ctx* c, int nseq, const char* const* names, const char* const* seqs, const int64_t* lens, int k, int w, vm_index** out) {
    if (k < 1 || k > 28 || w < 1 || w > 255) { set_error("k must be in [1,28], w in [1,255]"); return VM_ERR_ARG; }
    if (nseq < 0) { set_error("nseq < 0"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(c->device));
    vm_index* mi = new vm_index();
    struct Guard { vm_index* m; ~Guard() { if (m) vm_index_free(m); } } g{mi};
    mi->ctx = c; mi->k = k; mi->w = w;
    int64_t off = 0;
    for (int i = 0; i < nseq; ++i) {
        if (lens[i] < 0) { set_error("negative contig length"); return VM_ERR_ARG; }
        mi->names.emplace_back(names[i]); mi->lens.push_back(lens[i]); mi->offsets.push_back(off); off += lens[i];
    }
    mi->offsets.push_back(off);
    if (off >= (1LL << 35)) { set_error("reference longer than 2^35 bases"); return VM_ERR_UNSUPPORTED; }
    mi->bases.resize((size_t)off);
    for (int i = 0; i < nseq; ++i) {
        char* d = &mi->bases[(size_t)mi->offsets[i]]; const char* s = seqs[i];
        parallel_for_chunks(lens[i], (int64_t)16 << 20, [&](int64_t a, int64_t b) { for (int64_t x = a; x < b; ++x) { const char ch = s[x]; d[x] = (ch >= 'a' && ch <= 'z') ? ch - 32 : ch; } });
    }
    VMX_TRY(vmx_index_upload_codes(mi, seqs));
    VMX_TRY(index_build_device(mi));
    *out = mi; g.m = nullptr;
    return VM_OK;
}

extern "C" {

int vm_index_build_mem(vm_ctx* c, int nseq, const char* const* names, const char* const* seqs, const int64_t* lens, int k, int w, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    try { return index_build_mem_impl(c, nseq, names, seqs, lens, k, w, out); }
    catch (const std::bad_alloc&) { set_error("index build: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string("index build: ") + e.what()); return VM_ERR_ARG; }
}

int vm_index_build_fasta(vm_ctx* c, const char* path, int k, int w, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    try {
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
        return index_build_mem_impl(c, (int)names.size(), np.data(), sp.data(), ls.data(), k, w, out);
    }
    catch (const std::bad_alloc&) { set_error("index build: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string("index build: ") + e.what()); return VM_ERR_IO; }
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
int64_t vm_index_n_minimizers(const vm_index* mi) { return mi->n_min; }
int64_t vm_index_n_distinct(const vm_index* mi) { return mi->n_distinct; }
int vm_index_seq_info(const vm_index* mi, int i, const char** name, int64_t* len, int64_t* offset) {
    if (i < 0 || i >= (int)mi->names.size()) return VM_ERR_ARG;
    if (name) *name = mi->names[i].c_str(); if (len) *len = mi->lens[i]; if (offset) *offset = mi->offsets[i];
    return VM_OK;
}
// Aligner.seq(name)[start:end]: the host copy when there is one, else decoded from the HBM-resident codes (replicas made by
// vm_index_from_meta; ambiguous bases come back as N, as from minimap2's 4-bit store)
int64_t vm_index_seq(const vm_index* mi, int i, int64_t st, int64_t en, char* out) {
    if (i < 0 || i >= (int)mi->names.size()) return VM_ERR_ARG;
    if (st < 0) st = 0; if (en > mi->lens[i]) en = mi->lens[i];
    if (en <= st) return 0;
    if (mi->has_host_seq) { memcpy(out, mi->bases.data() + mi->offsets[i] + st, (size_t)(en - st)); return en - st; }
    vm_ctx* c = mi->ctx;
    if (hipSetDevice(c->device) != hipSuccess) return VM_ERR_HIP;
    DevBuf tmp; if (tmp.reserve((size_t)(en - st)) < 0) return VM_ERR_OOM;
    hipLaunchKernelGGL(k_decode, dim3(grid1d(en - st, 8192)), dim3(256), 0, c->stream, mi->d_codes.as<uint8_t>() + mi->offsets[i] + st, tmp.as<char>(), en - st);
    const hipError_t e = hipMemcpyAsync(out, tmp.p, (size_t)(en - st), hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    tmp.release();
    return (e == hipSuccess && e2 == hipSuccess) ? en - st : (int64_t)VM_ERR_HIP;
}
// tests: the sorted (hash, position) columns as the kernels see them, read back FROM THE DEVICE (hashes are recovered from the table)
int vm_index_minimizers(const vm_index* mi, uint64_t** hashes, uint64_t** positions, int64_t* n) {
    vm_ctx* c = mi->ctx;
    *n = mi->n_min;
    const size_t m = (size_t)std::max<int64_t>(*n, 1);
    *hashes = (uint64_t*)malloc(8 * m); *positions = (uint64_t*)malloc(8 * m);
    if (!*hashes || !*positions) { set_error("out of host memory"); return VM_ERR_OOM; }
    if (*n == 0) return VM_OK;
    VMX_HIP(hipSetDevice(c->device));
    DevBuf d_h; VMX_TRY(d_h.reserve(8 * m));
    hipLaunchKernelGGL(k_idx_fill_hashes, dim3(grid1d((int64_t)1 << mi->table_bits)), dim3(256), 0, c->stream, mi->d_table.as<vmx_slot>(), (int64_t)1 << mi->table_bits, d_h.as<uint64_t>());
    int rc = download(*hashes, d_h.p, (size_t)*n, c->stream);
    if (rc == 0) rc = download(*positions, mi->d_pos.p, (size_t)*n, c->stream);
    const hipError_t e = hipStreamSynchronize(c->stream);
    d_h.release();
    if (rc < 0) return rc;
    VMX_HIP(e);
    return VM_OK;
}

// ---- save / load: own format "<ref>.w<w>_k<k>.vmx" = header, contig table, upper-case bases, sorted positions. The hash column and
// the table are NOT stored: the loader recomputes every position's hash from the bases on the GPU, which also validates the file
// (position inside one contig, no ambiguous base, strand bit, strictly ascending (hash, position)).
static const char VMX_MAGIC[8] = {'V', 'M', 'X', 'I', 'D', 'X', '0', '2'};
int vm_index_save(const vm_index* mi, const char* path) {
    try {
        std::vector<uint64_t> pos((size_t)mi->n_min);
        vm_ctx* c = mi->ctx;
        VMX_HIP(hipSetDevice(c->device));
        VMX_TRY(download(pos.data(), mi->d_pos.p, pos.size(), c->stream));
        std::string dec;
        const int64_t tot = mi->offsets.back();
        if (!mi->has_host_seq) {
            dec.resize((size_t)tot);
            for (size_t i = 0; i < mi->names.size(); ++i) if (mi->lens[i] > 0 && vm_index_seq(mi, (int)i, 0, mi->lens[i], &dec[(size_t)mi->offsets[i]]) < 0) return VM_ERR_HIP;
        }
        VMX_HIP(hipStreamSynchronize(c->stream));
        FILE* f = fopen(path, "wb");
        if (!f) { set_error(std::string("cannot write ") + path); return VM_ERR_IO; }
        int64_t hdr[6] = {mi->k, mi->w, (int64_t)mi->names.size(), mi->n_min, tot, mi->mid_occ};
        fwrite(VMX_MAGIC, 1, 8, f); fwrite(hdr, 8, 6, f);
        for (size_t i = 0; i < mi->names.size(); ++i) { int64_t nl = (int64_t)mi->names[i].size(); fwrite(&nl, 8, 1, f); fwrite(mi->names[i].data(), 1, (size_t)nl, f); fwrite(&mi->lens[i], 8, 1, f); }
        const std::string& b = mi->has_host_seq ? mi->bases : dec;
        fwrite(b.data(), 1, b.size(), f);
        fwrite(pos.data(), 8, pos.size(), f);
        bool ok = !ferror(f); ok = (fclose(f) == 0) && ok;
        if (!ok) { set_error("write error"); return VM_ERR_IO; }
        return VM_OK;
    }
    catch (const std::exception& e) { set_error(std::string("index save: ") + e.what()); return VM_ERR_OOM; }
}

static int index_load_impl(vm_ctx* c, const char* path, vm_index** out) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
    struct Close { FILE* f; ~Close() { fclose(f); } } cl{f};
    struct stat sb; if (fstat(fileno(f), &sb) != 0) { set_error("cannot stat index file"); return VM_ERR_IO; }
    const int64_t fsize = (int64_t)sb.st_size;
    char mg[8]; int64_t hdr[6];
    if (fread(mg, 1, 8, f) != 8 || memcmp(mg, VMX_MAGIC, 8) || fread(hdr, 8, 6, f) != 6) { set_error("not a .vmx index (or an older format: rebuild it)"); return VM_ERR_IO; }
    const int64_t k = hdr[0], w = hdr[1], nseq = hdr[2], nmin = hdr[3], tot = hdr[4];
    // every header field is checked against its range and against the file size BEFORE anything is allocated
    if (k < 1 || k > 28 || w < 1 || w > 255 || nseq < 0 || nseq > (1 << 24) || nmin < 0 || nmin >= (1LL << 32) || tot < 0 || tot >= (1LL << 35) ||
        tot + 8 * nmin > fsize) { set_error("corrupt .vmx header"); return VM_ERR_IO; }
    vm_index* mi = new vm_index();
    struct Guard { vm_index* m; ~Guard() { if (m) vm_index_free(m); } } g{mi};
    mi->ctx = c; mi->k = (int)k; mi->w = (int)w;
    int64_t off = 0, meta = 8 + 48;
    for (int64_t i = 0; i < nseq; ++i) {
        int64_t nl = 0, ln = 0;
        if (fread(&nl, 8, 1, f) != 1 || nl < 0 || nl >= 65536) { set_error("corrupt .vmx contig table"); return VM_ERR_IO; }
        std::string nm((size_t)nl, ' ');
        if ((nl && fread(&nm[0], 1, (size_t)nl, f) != (size_t)nl) || fread(&ln, 8, 1, f) != 1 || ln < 0 || ln > tot - off) { set_error("corrupt .vmx contig table"); return VM_ERR_IO; }
        mi->names.push_back(nm); mi->lens.push_back(ln); mi->offsets.push_back(off); off += ln; meta += 16 + nl;
    }
    mi->offsets.push_back(off);
    if (off != tot || meta + tot + 8 * nmin != fsize) { set_error("corrupt .vmx index: sizes do not add up"); return VM_ERR_IO; }
    mi->bases.resize((size_t)tot);
    if (tot && fread(&mi->bases[0], 1, (size_t)tot, f) != (size_t)tot) { set_error("truncated .vmx index"); return VM_ERR_IO; }
    std::vector<uint64_t> pos((size_t)nmin);
    if (nmin && fread(pos.data(), 8, (size_t)nmin, f) != (size_t)nmin) { set_error("truncated .vmx index"); return VM_ERR_IO; }
    VMX_HIP(hipSetDevice(c->device));
    std::vector<const char*> sp; for (size_t i = 0; i < mi->names.size(); ++i) sp.push_back(mi->bases.data() + mi->offsets[i]);
    VMX_TRY(vmx_index_upload_codes(mi, sp.data()));
    VMX_TRY(upload(mi->d_pos, pos.data(), (size_t)nmin, c->stream));
    VMX_TRY(upload(mi->d_off, mi->offsets.data(), mi->offsets.size(), c->stream));
    DevBuf d_keys, d_err; struct Rel { DevBuf* b[2]; ~Rel() { for (auto* x : b) x->release(); } } rel{{&d_keys, &d_err}};
    VMX_TRY(d_keys.reserve(8 * (size_t)std::max<int64_t>(nmin, 1))); VMX_TRY(d_err.reserve(64)); VMX_HIP(hipMemsetAsync(d_err.p, 0, 8, c->stream));
    if (nmin) {
        hipLaunchKernelGGL(k_idx_pos_keys, dim3(grid1d(nmin)), dim3(256), 0, c->stream, mi->d_codes.as<uint8_t>(), mi->d_off.as<int64_t>(), (int)nseq, mi->d_pos.as<uint64_t>(), nmin, (int)k,
                           d_keys.as<uint64_t>(), d_err.as<int32_t>());
        hipLaunchKernelGGL(k_idx_check_sorted, dim3(grid1d(nmin)), dim3(256), 0, c->stream, d_keys.as<uint64_t>(), mi->d_pos.as<uint64_t>(), nmin, d_err.as<int32_t>() + 1);
    }
    int32_t err[2] = {0, 0};
    VMX_TRY(download(err, d_err.p, 2, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    if (err[0] || err[1]) { set_error("corrupt .vmx index: stored positions do not match the stored bases (" + std::to_string(err[0]) + " bad k-mers, " + std::to_string(err[1]) + " out of order)"); return VM_ERR_IO; }
    VMX_TRY(vmx_index_finish_device(mi, d_keys, nmin));
    *out = mi; g.m = nullptr;
    return VM_OK;
}
int vm_index_load(vm_ctx* c, const char* path, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    try { return index_load_impl(c, path, out); }
    catch (const std::bad_alloc&) { set_error("index load: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string("index load: ") + e.what()); return VM_ERR_IO; }
}

// ---- pieces of the HBM-resident index for the multi-GPU broadcast (RCCL over xGMI; vacmap_amd/dist.py)
int vm_index_blob_count(const vm_index*) { return 4; }
int vm_index_blob(const vm_index* mi, int i, void** dev_ptr, int64_t* bytes) {
    const DevBuf* b[4] = {&mi->d_codes, &mi->d_pos, &mi->d_table, &mi->d_off};
    int64_t sz[4] = {mi->offsets.back() + 64, mi->n_min * 8, (int64_t)sizeof(vmx_slot) << mi->table_bits, (int64_t)mi->offsets.size() * 8};
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
    int64_t hdr[8] = {mi->k, mi->w, (int64_t)mi->names.size(), mi->n_min, mi->offsets.back(), mi->mid_occ, mi->table_bits, mi->n_distinct};
    memcpy(p, hdr, 64); p += 64;
    for (size_t i = 0; i < mi->names.size(); ++i) { int64_t nl = (int64_t)mi->names[i].size(); memcpy(p, &nl, 8); p += 8; memcpy(p, &mi->lens[i], 8); p += 8; memcpy(p, mi->names[i].data(), (size_t)nl); p += nl; }
    return VM_OK;
}
int vm_index_from_meta(vm_ctx* c, const void* buf, int64_t bytes, vm_index** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (bytes < 64) { set_error("index metadata too short"); return VM_ERR_ARG; }
    try {
        const char* p = (const char*)buf; const char* end = p + bytes; int64_t hdr[8]; memcpy(hdr, p, 64); p += 64;
        if (hdr[0] < 1 || hdr[0] > 28 || hdr[1] < 1 || hdr[1] > 255 || hdr[2] < 0 || hdr[2] > (1 << 24) || hdr[3] < 0 || hdr[3] >= (1LL << 32) || hdr[4] < 0 || hdr[4] >= (1LL << 35) ||
            hdr[6] < 4 || hdr[6] > 40) { set_error("corrupt index metadata"); return VM_ERR_ARG; }
        vm_index* mi = new vm_index();
        struct Guard { vm_index* m; ~Guard() { if (m) vm_index_free(m); } } g{mi};
        mi->ctx = c; mi->k = (int)hdr[0]; mi->w = (int)hdr[1]; mi->n_min = hdr[3]; mi->mid_occ = (int)hdr[5]; mi->table_bits = (int)hdr[6]; mi->n_distinct = hdr[7];
        mi->has_host_seq = false;
        int64_t off = 0;
        for (int64_t i = 0; i < hdr[2]; ++i) {
            int64_t nl, ln;
            if (end - p < 16) { set_error("corrupt index metadata"); return VM_ERR_ARG; }
            memcpy(&nl, p, 8); p += 8; memcpy(&ln, p, 8); p += 8;
            if (nl < 0 || nl > end - p || ln < 0 || ln > hdr[4] - off) { set_error("corrupt index metadata"); return VM_ERR_ARG; }
            mi->names.emplace_back(p, (size_t)nl); p += nl; mi->lens.push_back(ln); mi->offsets.push_back(off); off += ln;
        }
        mi->offsets.push_back(off);
        if (off != hdr[4]) { set_error("corrupt index metadata"); return VM_ERR_ARG; }
        VMX_HIP(hipSetDevice(c->device));
        VMX_TRY(mi->d_codes.reserve((size_t)off + 64)); VMX_TRY(mi->d_pos.reserve((size_t)hdr[3] * 8 + 8));
        VMX_TRY(mi->d_table.reserve(sizeof(vmx_slot) << mi->table_bits)); VMX_TRY(mi->d_off.reserve(mi->offsets.size() * 8));
        *out = mi; g.m = nullptr;
        return VM_OK;
    }
    catch (const std::exception& e) { set_error(std::string("index replica: ") + e.what()); return VM_ERR_OOM; }
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
                   DevBuf* B /* >= 13 buffers */, std::vector<int64_t>& h_koff, std::vector<int64_t>& h_nhits, DevBuf* arena, int64_t** rows_out) {
    if (mid_occ <= 0) mid_occ = mi->mid_occ;
    // arena: a pool of the caller that is idle during this stage (the align path hands in its traceback pool, which only the gap fill uses, long
    // after the rows were compacted): the hit keys, their scratch copy and the row slots (48 B per hit: 8.4 GB for a batch of 35 kb reads) are views
    // into it instead of three pools of their own. *rows_out = where the rows are.
    DevBuf vk, vc, vr;
    DevBuf &mzh = B[0], &mzp = B[1], &mzc = B[2], &mst = B[3], &mcn = B[4], &mho = B[5], &nh = B[6], &koff = B[7], &keys = arena ? vk : B[8], &ckeys = arena ? vc : B[9],
           &rows = arena ? vr : B[10], &nanc = B[11];
    VMX_TRY(mzh.reserve(8 * (size_t)(total_bases + 1))); VMX_TRY(mzp.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mzc.reserve(4 * (size_t)(n + 1)));
    VMX_TRY(mst.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mcn.reserve(4 * (size_t)(total_bases + 1))); VMX_TRY(mho.reserve(4 * (size_t)(total_bases + 1)));
    VMX_TRY(nh.reserve(8 * (size_t)(n + 2))); VMX_TRY(koff.reserve(8 * (size_t)(n + 2))); VMX_TRY(nanc.reserve(4 * (size_t)(n + 1)));
    const unsigned grid = (unsigned)std::max<int64_t>(std::min<int64_t>(n, (int64_t)c->num_cu * 4), 1);
    if (2 * mi->k < 32) hipLaunchKernelGGL(k_sketch32, dim3(grid), dim3(256), 0, c->stream, d_codes, d_roff, (int)n, mi->k, mi->w, mzh.as<uint64_t>(), mzp.as<uint32_t>(), d_roff, mzc.as<int32_t>());      // hashes of fewer than 32 bits (k <= 15): the 32-bit form — at k = 16 a valid hash could equal its all-ones sentinel
    else hipLaunchKernelGGL(k_sketch, dim3(grid), dim3(256), 0, c->stream, d_codes, d_roff, (int)n, mi->k, mi->w, mzh.as<uint64_t>(), mzp.as<uint32_t>(), d_roff, mzc.as<int32_t>());
    hipLaunchKernelGGL(k_lookup, dim3(grid), dim3(256), 0, c->stream, mzh.as<uint64_t>(), d_roff, mzc.as<int32_t>(), (int)n, mi->d_table.as<vmx_slot>(), mi->table_bits, mid_occ,
                       mst.as<uint32_t>(), mcn.as<uint32_t>(), mho.as<uint32_t>(), nh.as<int64_t>());
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, nh.as<int64_t>(), koff.as<int64_t>(), n, 1);
    h_koff.resize((size_t)n + 1); h_nhits.resize((size_t)n);
    VMX_TRY(vmx_fetch(c, h_koff.data(), koff.p, (size_t)n + 1)); VMX_TRY(vmx_fetch(c, h_nhits.data(), nh.p, (size_t)n));
    std::vector<int32_t> h_mzc((size_t)n);
    VMX_TRY(vmx_fetch(c, h_mzc.data(), mzc.p, (size_t)n));
    VMX_HIP(vmx_stream_sync(c));   // sizing sync #1: total (power-of-two padded) hits of the batch
    c->last_n_minimizers = 0; for (int64_t r = 0; r < n; ++r) c->last_n_minimizers += h_mzc[r];
    const int64_t ktot = h_koff[n];
    if (arena) {
        const size_t kb = (8 * (size_t)(ktot + 1) + 255) & ~(size_t)255;
        VMX_TRY(arena->reserve(2 * kb + 32 * (size_t)(ktot + 1)));
        vk.p = arena->p; vk.cap = kb; vc.p = (char*)arena->p + kb; vc.cap = kb; vr.p = (char*)arena->p + 2 * kb; vr.cap = 32 * (size_t)(ktot + 1);
    } else {
        VMX_TRY(keys.reserve(8 * (size_t)(ktot + 1))); VMX_TRY(ckeys.reserve(8 * (size_t)(ktot + 1))); VMX_TRY(rows.reserve(32 * (size_t)(ktot + 1)));
    }
    if (rows_out) *rows_out = rows.as<int64_t>();
    hipLaunchKernelGGL(k_fill_hits, dim3(grid), dim3(256), 0, c->stream, mzp.as<uint32_t>(), d_roff, mzc.as<int32_t>(), (int)n, mst.as<uint32_t>(), mcn.as<uint32_t>(),
                       mho.as<uint32_t>(), mi->d_pos.as<uint64_t>(), keys.as<uint64_t>(), koff.as<int64_t>(), nh.as<int64_t>());
    // clustering, one launch per size class (the sort's LDS tile): reads with <= 4096 hits (32 KB of LDS, several workgroups per CU) and
    // the others (16384-key tile = 128 KB, one 1024-thread workgroup per CU), longest first
    {
        std::vector<int32_t> small, big;
        static const int big_env = [] { const char* e = getenv("VMX_CLUSTER_BIG"); return e ? atoi(e) : -1; }();     // tuning knob: 0 = every read through the 32 KB kernel
        const bool use_big = big_env >= 0 ? big_env != 0 : true;
        int64_t small_max = VMX_SORT_LDS;                            // test knob: reads with more hits than this take the 1024-thread kernel
        if (const char* e = getenv("VMX_CLUSTER_SMALL_MAX")) { const long long v = atoll(e); if (v >= 0 && v <= VMX_SORT_LDS) small_max = v; }
        for (int64_t r = 0; r < n; ++r) ((h_nhits[r] <= small_max || !use_big) ? small : big).push_back((int32_t)r);
        if (!use_big) std::stable_sort(small.begin(), small.end(), [&](int32_t a, int32_t b) { return h_nhits[a] > h_nhits[b]; });
        std::stable_sort(big.begin(), big.end(), [&](int32_t a, int32_t b) { return h_nhits[a] > h_nhits[b]; });
        // the read list, then two lists filled on the device, each followed by its length (zero): the reads k_cluster_big declines (they go to
        // k_cluster_long) and the reads k_cluster_long declines (they take the general path)
        std::vector<int32_t> rl(small); rl.insert(rl.end(), big.begin(), big.end());
        const size_t decl_at = rl.size();
        rl.resize(decl_at + 2 * (big.size() + 1), 0);
        VMX_TRY(vmx_push(c, B[12], rl.data(), rl.size()));
        const int32_t* d_rl = B[12].as<int32_t>();
        int32_t* d_decl = B[12].as<int32_t>() + decl_at; int32_t* d_ndecl = d_decl + big.size();
        int32_t* d_decl2 = d_ndecl + 1; int32_t* d_ndecl2 = d_decl2 + big.size();
        (void)hipEventRecord(c->kev[2], c->stream);
        if (!big.empty()) {
#ifndef VMX_EMU
            VMX_HIP(hipFuncSetAttribute((const void*)k_cluster_gen, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * VMX_SORT_LDS_BIG));
            VMX_HIP(hipFuncSetAttribute((const void*)k_cluster_long, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * VMX_SORT_LDS_BIG));
            VMX_HIP(hipFuncSetAttribute((const void*)k_cluster_big, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * VMX_SORT_LDS_BIG));
#endif
            // `big` is ordered by hit count, largest first: reads with more than 16383 hits go straight to the LONG filtered form (k_cluster_long: 128 KB
            // tile); the others run the filtered form in a 64 KB tile, two workgroups per CU (k_cluster_big; VMX_CLUSTER_TILE: keys of that tile, tuning
            // knob), those it declines follow in a second launch of k_cluster_long, and what that declines takes the general path (k_cluster_gen)
            static const int mid_tile = [] { const char* e = getenv("VMX_CLUSTER_TILE"); const int v = e ? atoi(e) : 0; return v >= 8192 && v <= VMX_SORT_LDS_BIG ? v : (VMX_SORT_LDS_BIG >= 16384 ? 8192 : VMX_SORT_LDS_BIG); }();
            static const int filt_wgs = [] { const char* e = getenv("VMX_CLUSTER_WGS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 2; }();
            int64_t huge_min = 0x3fff;                                 // test knob: reads with more hits than this go straight to k_cluster_long
            if (const char* e = getenv("VMX_CLUSTER_HUGE_MIN")) { const long long v = atoll(e); if (v >= 0 && v <= 0x3fff) huge_min = v; }
            size_t n_huge = 0; while (n_huge < big.size() && h_nhits[big[n_huge]] > huge_min) ++n_huge;
            const size_t n_mid = big.size() - n_huge;
            if (n_huge)
                hipLaunchKernelGGL(k_cluster_long, dim3((unsigned)std::min<int64_t>((int64_t)n_huge, c->num_cu)), dim3(1024), 8 * VMX_SORT_LDS_BIG, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(),
                                   koff.as<int64_t>(), nh.as<int64_t>(), d_rl + small.size(), (int)n_huge, (const int32_t*)nullptr, VMX_SORT_LDS_BIG, check_num, mi->k, rows.as<int64_t>(), nanc.as<int32_t>(),
                                   d_decl2, d_ndecl2);
            if (n_mid) {
                hipLaunchKernelGGL(k_cluster_big, dim3((unsigned)std::min<int64_t>((int64_t)n_mid, (int64_t)c->num_cu * filt_wgs)), dim3(VMX_CF_BLOCK), (size_t)8 * mid_tile, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(),
                                   koff.as<int64_t>(), nh.as<int64_t>(), d_rl + small.size() + n_huge, (int)n_mid, mid_tile, check_num, mi->k, rows.as<int64_t>(), nanc.as<int32_t>(), d_decl, d_ndecl);
                hipLaunchKernelGGL(k_cluster_long, dim3((unsigned)std::min<int64_t>((int64_t)n_mid, c->num_cu)), dim3(1024), 8 * VMX_SORT_LDS_BIG, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(),
                                   koff.as<int64_t>(), nh.as<int64_t>(), (const int32_t*)d_decl, 0, (const int32_t*)d_ndecl, VMX_SORT_LDS_BIG, check_num, mi->k, rows.as<int64_t>(), nanc.as<int32_t>(),
                                   d_decl2, d_ndecl2);
            }
            hipLaunchKernelGGL(k_cluster_gen, dim3((unsigned)std::min<int64_t>((int64_t)big.size(), c->num_cu)), dim3(1024), 8 * VMX_SORT_LDS_BIG, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(),
                               koff.as<int64_t>(), nh.as<int64_t>(), (const int32_t*)d_decl2, 0, (const int32_t*)d_ndecl2, VMX_SORT_LDS_BIG, check_num, mi->k, rows.as<int64_t>(), nanc.as<int32_t>());
        }
        if (!small.empty())
            hipLaunchKernelGGL(k_cluster, dim3((unsigned)std::min<int64_t>((int64_t)small.size(), (int64_t)c->num_cu * 4)), dim3(256), 8 * VMX_SORT_LDS, c->stream, keys.as<uint64_t>(), ckeys.as<uint64_t>(),
                               koff.as<int64_t>(), nh.as<int64_t>(), d_rl, (int)small.size(), VMX_SORT_LDS, check_num, mi->k, rows.as<int64_t>(), nanc.as<int32_t>());
        (void)hipEventRecord(c->kev[3], c->stream); c->kev_set |= 2;
    }
    return 0;
}

int vm_map_batch(vm_ctx* c, const vm_index* mi, int check_num, int mid_occ, int64_t n, const char* seqs, const int64_t* off, int64_t** anchors, int64_t** anchor_off) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_reads(c, n, seqs, off, c->b[0], c->b[1], c->b[2]));
    std::vector<int64_t> koff, nhits;
    if (n) VMX_TRY(vmx_seed_stage(c, mi, check_num, mid_occ, n, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(), off[n], &c->b[3], koff, nhits, nullptr, nullptr));
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
