// k_seed.hip — minimizer sketch, hashed index lookup and hit clustering on gfx950.
//
// Replaces `Aligner.map(seq, check_num=, mid_occ=)` of the un-vendored vacmap_index (call site
// /root/reference/src/vacmap/mammap_clrnano.py:23985). Implements the build's spec VMX-S1 (oracle/vmo_seed.cc):
//   k_sketch    one workgroup per read: read codes are staged in LDS, every lane hashes one k-mer start per step
//               (2-bit canonical k-mer, invertible 64-bit mix), window minima over w starts, ordered compaction.
//   k_lookup    one workgroup per read: open-addressing probe of the HBM-resident table (16-B slots), occurrence cap,
//               per-minimizer hit offsets by a block scan.
//   k_fill_hits one workgroup per read: packs every (ref pos, read pos, strand) hit into one 64-bit sort key.
//   k_cluster   one workgroup per read: bitonic sort of the keys (LDS when they fit), clusters cut at ref gaps > 5000,
//               clusters ranked by (size desc, first ref pos asc), the first check_num emitted as anchor rows.
// HBM-bound integer/byte work: coalesced loads, LDS staging, no MFMA.
#include "vmx_device.h"
#include "vmx_kernels.h"

__device__ __forceinline__ uint64_t vmx_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

#define VMX_SK_TILE 2048          // k-mer starts per tile
#define VMX_SK_HALO 256           // >= w-1 on each side
#define VMX_INF64 (~0ULL)

// codes: 1 byte per base (0..4). Outputs per read at mz_off[r] (capacity = number of k-mer starts): hash, pos<<1|strand.
__global__ void __launch_bounds__(256) k_sketch(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, int n_reads, int k, int w,
                                                uint64_t* __restrict__ mz_hash, uint32_t* __restrict__ mz_ps, const int64_t* __restrict__ mz_off,
                                                int32_t* __restrict__ mz_cnt) {
    __shared__ uint8_t s_codes[VMX_SK_TILE + 2 * VMX_SK_HALO + 64];
    __shared__ uint64_t s_h[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ uint64_t s_wmin[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ uint8_t s_z[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ int s_scan[20];
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    const int shift = 2 * (k - 1);
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint8_t* C = codes + roff[r];
        const int L = (int)(roff[r + 1] - roff[r]);
        const int P = L - k + 1;
        uint64_t* oh = mz_hash + mz_off[r];
        uint32_t* op = mz_ps + mz_off[r];
        int written = 0;
        if (P <= 0) { if (threadIdx.x == 0) mz_cnt[r] = 0; continue; }
        const int nwin = P >= w ? P - w + 1 : 1;
        const int wl = P >= w ? w : P;
        if (w == 10 && P >= 10 && blockDim.x == 256) {
            // The common geometry (w = 10, every mode's default): each thread owns EIGHT consecutive positions of the tile.
            //  * k-mers are rolled (two shifts per base instead of a k-base loop per position);
            //  * the window minima of its 8 starts come from 17 hashes held in registers, by doubling (pairs, fours, eights; 10 = 8 + 2);
            //  * a position is a minimizer iff its hash equals the minimum of SOME window that holds it; every such window's minimum is <= the
            //    hash, so that is the same as: the MAXIMUM of the minima of the windows holding it equals the hash — again a width-10
            //    sliding reduction by doubling (windows outside [0, nwin) count as 0, below every hash that could match);
            //  * one block scan per tile places the tile's minimizers in order.
            for (int t0 = 0; t0 < P; t0 += VMX_SK_TILE) {
                const int lo = t0 - 9 > 0 ? t0 - 9 : 0;
                int hi = t0 + VMX_SK_TILE + 9; if (hi > P) hi = P;
                const int npos = hi - lo;
                for (int x = (int)threadIdx.x; x < npos + k - 1; x += 256) s_codes[x] = C[lo + x];
                __syncthreads();
                for (int x0 = 8 * (int)threadIdx.x; x0 < npos; x0 += 8 * 256) {
                    uint64_t fwd = 0, rc = 0; int nval = 0;
                    for (int i = 0; i < k - 1; ++i) { const uint8_t c = s_codes[x0 + i]; nval = c > 3 ? 0 : nval + 1; fwd = (fwd << 2) | (uint64_t)(c & 3); rc = (rc >> 2) | ((uint64_t)(3 - (c & 3)) << shift); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (x0 + j < npos) {
                            const uint8_t c = s_codes[x0 + j + k - 1]; nval = c > 3 ? 0 : nval + 1;
                            fwd = ((fwd << 2) | (uint64_t)(c & 3)) & mask; rc = (rc >> 2) | ((uint64_t)(3 - (c & 3)) << shift);
                            uint64_t h = VMX_INF64; uint8_t z = 0;
                            if (nval >= k && fwd != rc) { z = rc < fwd ? 1 : 0; h = vmx_hash64(fwd < rc ? fwd : rc, mask); }
                            s_h[x0 + j] = h; s_z[x0 + j] = z;
                        }
                    }
                }
                __syncthreads();
                for (int x0 = 8 * (int)threadIdx.x; x0 < npos; x0 += 8 * 256) {
                    uint64_t v[17], m2[16], m4[14], m8[8];
#pragma unroll
                    for (int i = 0; i < 17; ++i) v[i] = x0 + i < npos ? s_h[x0 + i] : VMX_INF64;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m2[i] = v[i] < v[i + 1] ? v[i] : v[i + 1];
#pragma unroll
                    for (int i = 0; i < 14; ++i) m4[i] = m2[i] < m2[i + 2] ? m2[i] : m2[i + 2];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { m8[i] = m4[i] < m4[i + 4] ? m4[i] : m4[i + 4]; const uint64_t m = m8[i] < m2[i + 8] ? m8[i] : m2[i + 8]; if (x0 + i < npos) s_wmin[x0 + i] = (lo + x0 + i < nwin) ? m : 0ULL; }
                }
                __syncthreads();
                int pend = t0 + VMX_SK_TILE; if (pend > P) pend = P;
                const int p0 = t0 + 8 * (int)threadIdx.x;               // this thread's positions p0 .. p0 + 7
                unsigned selmask = 0; uint64_t hs[8];
                if (p0 < pend) {
                    const int xa = p0 - 9 - lo;                          // LDS index of window start p0 - 9 (negative: before the sequence)
                    uint64_t v[17], m2[16], m4[14];
#pragma unroll
                    for (int i = 0; i < 17; ++i) v[i] = (xa + i >= 0 && xa + i < npos) ? s_wmin[xa + i] : 0ULL;
#pragma unroll
                    for (int i = 0; i < 16; ++i) m2[i] = v[i] > v[i + 1] ? v[i] : v[i + 1];
#pragma unroll
                    for (int i = 0; i < 14; ++i) m4[i] = m2[i] > m2[i + 2] ? m2[i] : m2[i + 2];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint64_t m8 = m4[j] > m4[j + 4] ? m4[j] : m4[j + 4];
                        const uint64_t mx = m8 > m2[j + 8] ? m8 : m2[j + 8];          // windows p - 9 .. p
                        const uint64_t h = p0 + j < pend ? s_h[p0 + j - lo] : VMX_INF64;
                        hs[j] = h;
                        if (h != VMX_INF64 && mx == h) selmask |= 1u << j;
                    }
                }
                int tot; const int ex = vmx_block_excl_scan(__popc(selmask), s_scan, &tot);
                if (selmask) {
                    int o = written + ex;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if ((selmask >> j) & 1u) { oh[o] = hs[j]; op[o] = ((uint32_t)(p0 + j) << 1) | s_z[p0 + j - lo]; ++o; }
                }
                written += tot;
                __syncthreads();
            }
            if (threadIdx.x == 0) mz_cnt[r] = written;
            continue;
        }
        for (int t0 = 0; t0 < P; t0 += VMX_SK_TILE) {
            // positions [t0 - (w-1), t0 + TILE + (w-1)) clipped to [0, P) are needed for the window minima
            const int lo = t0 - (w - 1) > 0 ? t0 - (w - 1) : 0;
            int hi = t0 + VMX_SK_TILE + (w - 1); if (hi > P) hi = P;
            const int npos = hi - lo;                       // <= TILE + 2(w-1)
            for (int x = (int)threadIdx.x; x < npos + k - 1; x += (int)blockDim.x) s_codes[x] = C[lo + x];
            __syncthreads();
            for (int x = (int)threadIdx.x; x < npos; x += (int)blockDim.x) {
                uint64_t fwd = 0, rc = 0; bool ok = true;
                for (int i = 0; i < k; ++i) {
                    uint8_t c = s_codes[x + i];
                    if (c > 3) ok = false;
                    fwd = (fwd << 2) | (uint64_t)(c & 3);
                    rc = (rc >> 2) | ((uint64_t)(3 - (c & 3)) << shift);
                }
                uint64_t h = VMX_INF64; uint8_t z = 0;
                if (ok && fwd != rc) { z = rc < fwd ? 1 : 0; h = vmx_hash64(fwd < rc ? fwd : rc, mask); }
                s_h[x] = h; s_z[x] = z;
            }
            __syncthreads();
            // window minima for window starts a in [lo, min(hi, nwin)): min over [a, a+wl)
            for (int x = (int)threadIdx.x; x < npos; x += (int)blockDim.x) {
                int a = lo + x; uint64_t m = VMX_INF64;
                if (a < nwin) { for (int j = 0; j < wl; ++j) { int y = x + j; if (y < npos) { uint64_t v = s_h[y]; m = v < m ? v : m; } } }
                s_wmin[x] = m;   // windows whose span leaves [lo,hi) are never consulted for this tile's positions
            }
            __syncthreads();
            // selection + ordered compaction for p in [t0, min(t0+TILE, P))
            int pend = t0 + VMX_SK_TILE; if (pend > P) pend = P;
            for (int pb = t0; pb < pend; pb += (int)blockDim.x) {
                int p = pb + (int)threadIdx.x;
                int sel = 0; uint64_t h = VMX_INF64; uint8_t z = 0;
                if (p < pend) {
                    int x = p - lo; h = s_h[x]; z = s_z[x];
                    if (h != VMX_INF64) {
                        int a0 = p - wl + 1; if (a0 < 0) a0 = 0;
                        int a1 = p; if (a1 > nwin - 1) a1 = nwin - 1;
                        for (int a = a0; a <= a1; ++a) if (s_wmin[a - lo] == h) { sel = 1; break; }
                    }
                }
                int tot; int ex = vmx_block_excl_scan(sel, s_scan, &tot);
                if (sel) { oh[written + ex] = h; op[written + ex] = ((uint32_t)p << 1) | z; }
                written += tot;
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) mz_cnt[r] = written;
    }
}

// open-addressing table: slot = {key, start, count}; empty key = ~0. probe = golden-ratio multiplicative hash, linear.
__device__ __forceinline__ bool vmx_table_find(const vmx_slot* __restrict__ tab, int bits, uint64_t key, uint32_t& start, uint32_t& cnt) {
    uint64_t m = (1ULL << bits) - 1;
    uint64_t i = (key * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
    while (true) {
        vmx_slot s = tab[i];
        if (s.key == key) { start = s.start; cnt = s.count; return true; }
        if (s.key == VMX_INF64) return false;
        i = (i + 1) & m;
    }
}

__global__ void __launch_bounds__(256) k_lookup(const uint64_t* __restrict__ mz_hash, const int64_t* __restrict__ mz_off,
                                                const int32_t* __restrict__ mz_cnt, int n_reads, const vmx_slot* __restrict__ tab, int bits,
                                                int mid_occ, uint32_t* __restrict__ m_start, uint32_t* __restrict__ m_cnt,
                                                uint32_t* __restrict__ m_hoff, int64_t* __restrict__ nhits) {
    __shared__ int s_scan[20];
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const int64_t base = mz_off[r];
        const int M = mz_cnt[r];
        int run = 0;
        for (int m0 = 0; m0 < M; m0 += (int)blockDim.x) {
            int m = m0 + (int)threadIdx.x;
            uint32_t st = 0, cn = 0;
            if (m < M) { if (!vmx_table_find(tab, bits, mz_hash[base + m], st, cn) || (int)cn > mid_occ) cn = 0; }
            int tot; int ex = vmx_block_excl_scan((int)cn, s_scan, &tot);
            if (m < M) { m_start[base + m] = st; m_cnt[base + m] = cn; m_hoff[base + m] = (uint32_t)(run + ex); }
            run += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) nhits[r] = run;
    }
}

// key = ref_pos << 28 | read_pos << 1 | (strand == +1)
__global__ void __launch_bounds__(256) k_fill_hits(const uint32_t* __restrict__ mz_ps, const int64_t* __restrict__ mz_off, const int32_t* __restrict__ mz_cnt,
                                                   int n_reads, const uint32_t* __restrict__ m_start, const uint32_t* __restrict__ m_cnt,
                                                   const uint32_t* __restrict__ m_hoff, const uint64_t* __restrict__ idx_pos,
                                                   uint64_t* __restrict__ keys, const int64_t* __restrict__ key_off, const int64_t* __restrict__ nhits) {
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const int64_t base = mz_off[r];
        const int M = mz_cnt[r];
        uint64_t* K = keys + key_off[r];
        for (int m = (int)threadIdx.x; m < M; m += (int)blockDim.x) {
            uint32_t cn = m_cnt[base + m];
            if (!cn) continue;
            uint32_t ps = mz_ps[base + m];
            uint64_t q = ps >> 1; uint32_t zq = ps & 1;
            uint64_t* out = K + m_hoff[base + m];
            const uint64_t* src = idx_pos + m_start[base + m];
            for (uint32_t e = 0; e < cn; ++e) {
                uint64_t pv = src[e];
                uint64_t sb = ((uint32_t)(pv & 1) == zq) ? 1ULL : 0ULL;
                out[e] = ((pv >> 1) << 28) | (q << 1) | sb;
            }
        }
        // pad to the next power of two for the bitonic sort
        int n = (int)nhits[r]; int N = 1; while (N < n) N <<= 1;
        if (n == 0) N = 0;   // a read without hits owns no key slots
        for (int i = n + (int)threadIdx.x; i < N; i += (int)blockDim.x) K[i] = VMX_INF64;
    }
}

// sort hits, cut clusters (ref gap > 5000), rank by (size desc, first ref asc), emit the first check_num clusters.
// cl_keys: scratch with the same geometry as keys. rows out at key_off[r] (capacity nhits[r]).
// The reads of a launch come from `rlist` (one size class per launch): `tile` keys of dynamic LDS hold the sort — 4096 (32 KB, several
// workgroups per CU) for reads with few hits, 16384 (128 KB, one workgroup of BLOCK threads per CU) for the others; a read with more
// hits than a tile (a 15 kb read meets ~11 k hits in an hg38-size index, a 60 kb read 45 k) is sorted tile-wise with the few long-distance
// steps through HBM (vmx_block_sort_u64_tiled).
template <int BLOCK>
__device__ __forceinline__ void vmx_cluster_body(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                 const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, int tile, int check_num, int kmer,
                                                 int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors) {
    VMX_DYN_SHARED(uint64_t, s_sort);
    __shared__ int s_scan[20];
    __shared__ int s_ncl;
    for (int x = blockIdx.x; x < nlist; x += gridDim.x) {
        const int r = rlist[x];
        const int n = (int)nhits[r];
        if (n == 0) { if (threadIdx.x == 0) n_anchors[r] = 0; continue; }
        int N = 1; while (N < n) N <<= 1;
        uint64_t* K = keys + key_off[r];
        uint64_t* CK = cl_keys + key_off[r];
        if (N > 1) vmx_block_sort_u64_tiled(K, N, s_sort, tile);
        __syncthreads();
        // cluster starts, compacted in order: CK[c] = start index of cluster c (temporarily)
        int run = 0;
        for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
            int i = i0 + (int)threadIdx.x;
            int f = 0;
            if (i < n) f = (i == 0) || ((long long)(K[i] >> 28) - (long long)(K[i - 1] >> 28) > 5000);
            int tot; int ex = vmx_block_excl_scan(f, s_scan, &tot);
            if (f) CK[run + ex] = (uint64_t)i;
            run += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) s_ncl = run;
        __syncthreads();
        const int ncl = s_ncl;
        // rank key = (0xffffffff - size) << 32 | start  (ascending = size desc, start asc = first ref pos asc)
        int NC = 1; while (NC < ncl) NC <<= 1;
        const bool topk = check_num > 0 && check_num < ncl && check_num <= 1024;
        if (topk) {
            // Only the check_num best-ranked clusters are emitted, and a read's ~10^4 clusters in an hg38-size index are almost all
            // single random hits: instead of sorting every rank key, find the size s* of the check_num-th cluster from a size histogram,
            // take every cluster larger than s* and the first of size s* in reference order (CK is in that order), and sort those few keys.
            int* hist = (int*)s_sort;                                 // sizes 1..1022, bin 1023 = larger (all of them are taken)
            for (int i = (int)threadIdx.x; i < 1024; i += (int)blockDim.x) hist[i] = 0;
            __syncthreads();
            for (int c = (int)threadIdx.x; c < ncl; c += (int)blockDim.x) {
                const uint64_t st = CK[c], en = (c + 1 < ncl) ? CK[c + 1] : (uint64_t)n;
                const int sz = (int)(en - st);
                atomicAdd(&hist[sz < 1023 ? sz : 1023], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int acc = 0, sstar = 0, above = 0;
                for (int b = 1023; b >= 1; --b) { if (acc + hist[b] >= check_num) { sstar = b; above = acc; break; } acc += hist[b]; }
                s_ncl = sstar; s_scan[18] = above;                    // clusters strictly larger than s*, and s* (1023: the overflow bin holds the cut)
            }
            __syncthreads();
            const int sstar = s_ncl, above = s_scan[18];
            __syncthreads();
            if (sstar >= 1023 || sstar == 0) {
                // the cut falls among clusters of 1023 hits or more (or nothing was found): keep the general path
            } else {
                const int need = check_num - above;                   // clusters of size exactly s* to take, first in reference order
                uint64_t* sel = s_sort + 512;                         // selected rank keys (<= check_num <= 1024), behind the histogram
                int nsel = 0, neq = 0;
                for (int c0 = 0; c0 < ncl; c0 += (int)blockDim.x) {
                    const int c = c0 + (int)threadIdx.x;
                    int sz = 0; uint64_t st = 0;
                    if (c < ncl) { st = CK[c]; const uint64_t en = (c + 1 < ncl) ? CK[c + 1] : (uint64_t)n; sz = (int)(en - st); }
                    const int eq = sz == sstar;
                    int toteq; const int exeq = vmx_block_excl_scan(eq, s_scan, &toteq);
                    const int take = sz > sstar || (eq && neq + exeq < need);
                    __syncthreads();
                    int tott; const int ext = vmx_block_excl_scan(take, s_scan, &tott);
                    if (take) sel[nsel + ext] = ((uint64_t)(0xffffffffu - (uint32_t)sz) << 32) | st;
                    nsel += tott; neq += toteq;
                    __syncthreads();
                }
                int NS = 1; while (NS < nsel) NS <<= 1;
                for (int i = nsel + (int)threadIdx.x; i < NS; i += (int)blockDim.x) sel[i] = VMX_INF64;
                __syncthreads();
                if (NS > 1) vmx_block_bitonic_passes(sel, NS);
                __syncthreads();
                for (int i = (int)threadIdx.x; i < nsel; i += (int)blockDim.x) CK[i] = sel[i];
                __syncthreads();
                NC = 0;                                               // CK[0 .. check_num) is ranked: skip the full sort
            }
        }
        if (NC > 0) {
        // sizes need the next start: read all first, then overwrite
        uint64_t mykey[1];
        for (int c0 = 0; c0 < NC; c0 += (int)blockDim.x) {
            int c = c0 + (int)threadIdx.x;
            uint64_t kk = VMX_INF64;
            if (c < ncl) {
                uint64_t st = CK[c]; uint64_t en = (c + 1 < ncl) ? CK[c + 1] : (uint64_t)n;
                kk = ((uint64_t)(0xffffffffu - (uint32_t)(en - st)) << 32) | st;
            }
            mykey[0] = kk;
            __syncthreads();
            if (c < NC) CK[c] = mykey[0];
            __syncthreads();
        }
        if (NC > 1) vmx_block_sort_u64_tiled(CK, NC, s_sort, tile);
        __syncthreads();
        }
        int keep = ncl; if (check_num > 0 && check_num < keep) keep = check_num;
        // emit: exclusive scan of kept cluster sizes gives the output offset of each cluster
        int64_t* out = rows + 4 * key_off[r];
        int outbase = 0;
        if (keep <= tile) {
            // (offset, start) of every kept cluster parked in LDS (the sort tile is idle now), then ALL threads copy hits: element e belongs
            // to the last cluster whose offset is <= e. One thread per cluster would leave the true locus (hundreds of hits) to a single lane.
            for (int c0 = 0; c0 < keep; c0 += (int)blockDim.x) {
                int c = c0 + (int)threadIdx.x;
                int sz = 0, st = 0;
                if (c < keep) { uint64_t kk = CK[c]; sz = (int)(0xffffffffu - (uint32_t)(kk >> 32)); st = (int)(kk & 0xffffffffu); }
                int tot; int ex = vmx_block_excl_scan(sz, s_scan, &tot);
                if (c < keep) s_sort[c] = ((uint64_t)(uint32_t)(outbase + ex) << 32) | (uint32_t)st;
                outbase += tot;
                __syncthreads();
            }
            for (int e = (int)threadIdx.x; e < outbase; e += (int)blockDim.x) {
                int lo = 0, hi = keep;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)(s_sort[mid] >> 32) <= e) lo = mid; else hi = mid; }
                const uint64_t ce = s_sort[lo];
                const uint64_t hk = K[(int)(ce & 0xffffffffu) + (e - (int)(ce >> 32))];
                int64_t* o = out + 4 * (int64_t)e;
                o[0] = (int64_t)((hk >> 1) & 0x7ffffffULL); o[1] = (int64_t)(hk >> 28); o[2] = (hk & 1) ? 1 : -1; o[3] = kmer;
            }
            __syncthreads();
        } else
        for (int c0 = 0; c0 < keep; c0 += (int)blockDim.x) {
            int c = c0 + (int)threadIdx.x;
            int sz = 0, st = 0;
            if (c < keep) { uint64_t kk = CK[c]; sz = (int)(0xffffffffu - (uint32_t)(kk >> 32)); st = (int)(kk & 0xffffffffu); }
            int tot; int ex = vmx_block_excl_scan(sz, s_scan, &tot);
            // each thread copies its own cluster
            for (int e = 0; e < sz; ++e) {
                uint64_t hk = K[st + e];
                int64_t* o = out + 4 * (int64_t)(outbase + ex + e);
                o[0] = (int64_t)((hk >> 1) & 0x7ffffffULL); o[1] = (int64_t)(hk >> 28); o[2] = (hk & 1) ? 1 : -1; o[3] = kmer;
            }
            outbase += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) n_anchors[r] = outbase;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_cluster(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                 const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, int tile, int check_num, int kmer,
                                                 int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors) {
    vmx_cluster_body<256>(keys, cl_keys, key_off, nhits, rlist, nlist, tile, check_num, kmer, rows, n_anchors);
}
__global__ void __launch_bounds__(1024) k_cluster_big(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                      const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, int tile, int check_num, int kmer,
                                                      int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors) {
    vmx_cluster_body<1024>(keys, cl_keys, key_off, nhits, rlist, nlist, tile, check_num, kmer, rows, n_anchors);
}

// three-phase exclusive scan for large n: per-chunk sums -> k_scan_i64 over the sums -> per-chunk scan with its base
__global__ void __launch_bounds__(256) k_scan_part(const int64_t* __restrict__ in, int64_t* __restrict__ part, int64_t n, int64_t chunk) {
    __shared__ long long s_w[4];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    long long s = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += in[i];
    s = vmx_wave_sum_i64(s);
    if (vmx_lane() == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(256) k_scan_apply(const int64_t* __restrict__ in, int64_t* __restrict__ out, const int64_t* __restrict__ part_off, int64_t n, int64_t chunk) {
    __shared__ long long s_part[256];
    __shared__ long long s_base;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x == 0) s_base = part_off[blockIdx.x];
    __syncthreads();
    for (int64_t i0 = lo; i0 < hi; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        long long v = i < hi ? in[i] : 0;
        // wave-level inclusive scan, then combine the 4 waves
        long long inc = v;
        for (int o = 1; o < 64; o <<= 1) { long long x = __shfl_up(inc, o); if (vmx_lane() >= o) inc += x; }
        if (vmx_lane() == 63) s_part[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long wb = 0; for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wb += s_part[w];
        const long long tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (i < hi) out[i] = s_base + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = s_base;
}

// the same two phases with the element count read from the device (no host round trip to learn it): gridDim.x chunks of ceil(n / gridDim.x)
// elements; the middle phase is k_scan_i64 over the gridDim.x partial sums, whose total (part_off[gridDim.x]) becomes out[n]
__global__ void __launch_bounds__(256) k_scan_part_dev(const int64_t* __restrict__ in, int64_t* __restrict__ part, const int32_t* __restrict__ n_ptr) {
    __shared__ long long s_w[4];
    const int64_t n = *n_ptr, chunk = (n + gridDim.x - 1) / gridDim.x;
    int64_t lo = (int64_t)blockIdx.x * chunk; if (lo > n) lo = n;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    long long s = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += in[i];
    s = vmx_wave_sum_i64(s);
    if (vmx_lane() == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(256) k_scan_apply_dev(const int64_t* __restrict__ in, int64_t* __restrict__ out, const int64_t* __restrict__ part_off, const int32_t* __restrict__ n_ptr) {
    __shared__ long long s_part[256];
    __shared__ long long s_base;
    const int64_t n = *n_ptr, chunk = (n + gridDim.x - 1) / gridDim.x;
    int64_t lo = (int64_t)blockIdx.x * chunk; if (lo > n) lo = n;
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x == 0) s_base = part_off[blockIdx.x];
    __syncthreads();
    for (int64_t i0 = lo; i0 < hi; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        long long v = i < hi ? in[i] : 0;
        long long inc = v;
        for (int o = 1; o < 64; o <<= 1) { long long x = __shfl_up(inc, o); if (vmx_lane() >= o) inc += x; }
        if (vmx_lane() == 63) s_part[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long wb = 0; for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wb += s_part[w];
        const long long tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (i < hi) out[i] = s_base + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = part_off[gridDim.x];
}

// exclusive scan of n int64 values (single workgroup, n up to a few million): out[n] = total
__global__ void __launch_bounds__(256) k_scan_i64(const int64_t* __restrict__ in, int64_t* __restrict__ out, int64_t n, int pow2_round) {
    __shared__ long long s_part[256];
    __shared__ long long s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += 256) {
        int64_t i = i0 + threadIdx.x;
        long long v = 0;
        if (i < n) { v = in[i]; if (pow2_round) { long long N = 1; while (N < v) N <<= 1; v = v ? N : 0; } }
        s_part[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) { long long acc = s_base; for (int t = 0; t < 256; ++t) { long long x = s_part[t]; s_part[t] = acc; acc += x; } s_base = acc; }
        __syncthreads();
        if (i < n) out[i] = s_part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = s_base;
}
