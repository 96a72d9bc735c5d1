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

// the w = 10 form of k_sketch (below), with the hashes in H = uint32_t when 2 k <= 32 (k = 15: 30-bit hashes — half the LDS, half the
// instructions of the 64-bit form, the same values: every step of the hash is taken modulo 2^(2k)) or uint64_t (k = 19)
template <typename H> __device__ __forceinline__ H vmx_hash_t(H key, H mask) {
    key = (H)((~key + (key << 21)) & mask);
    key = (H)(key ^ key >> 24);
    key = (H)(((key + (key << 3)) + (key << 8)) & mask);
    key = (H)(key ^ key >> 14);
    key = (H)(((key + (key << 2)) + (key << 4)) & mask);
    key = (H)(key ^ key >> 28);
    key = (H)((key + (key << 31)) & mask);
    return key;
}
template <typename H>
__device__ __forceinline__ int vmx_sketch_w10(const uint8_t* __restrict__ C, int P, int k, int nwin, uint8_t* s_codes, H* s_h, H* s_wmin, uint8_t* s_z, int* s_scan,
                                              uint64_t* __restrict__ oh, uint32_t* __restrict__ op) {
    const H mask = (H)((2 * k >= (int)(8 * sizeof(H))) ? ~(H)0 : (((H)1 << (2 * k)) - 1));
    const int shift = 2 * (k - 1);
    const H HINF = ~(H)0;
    int written = 0;
            for (int t0 = 0; t0 < P; t0 += VMX_SK_TILE) {
    const int lo = t0 - 9 > 0 ? t0 - 9 : 0;
    int hi = t0 + VMX_SK_TILE + 9; if (hi > P) hi = P;
    const int npos = hi - lo;
    for (int x = (int)threadIdx.x; x < npos + k - 1; x += 256) s_codes[x] = C[lo + x];
    __syncthreads();
    for (int x0 = 8 * (int)threadIdx.x; x0 < npos; x0 += 8 * 256) {
        H fwd = 0, rc = 0; int nval = 0;
        for (int i = 0; i < k - 1; ++i) { const uint8_t c = s_codes[x0 + i]; nval = c > 3 ? 0 : nval + 1; fwd = (H)((fwd << 2) | (H)(c & 3)); rc = (H)((rc >> 2) | ((H)(3 - (c & 3)) << shift)); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (x0 + j < npos) {
                const uint8_t c = s_codes[x0 + j + k - 1]; nval = c > 3 ? 0 : nval + 1;
                fwd = (H)(((fwd << 2) | (H)(c & 3)) & mask); rc = (H)((rc >> 2) | ((H)(3 - (c & 3)) << shift));
                H h = HINF; uint8_t z = 0;
                if (nval >= k && fwd != rc) { z = rc < fwd ? 1 : 0; h = vmx_hash_t<H>(fwd < rc ? fwd : rc, mask); }
                s_h[x0 + j] = h; s_z[x0 + j] = z;
            }
        }
    }
    __syncthreads();
    for (int x0 = 8 * (int)threadIdx.x; x0 < npos; x0 += 8 * 256) {
        H v[17], m2[16], m4[14], m8[8];
#pragma unroll
        for (int i = 0; i < 17; ++i) v[i] = x0 + i < npos ? s_h[x0 + i] : HINF;
#pragma unroll
        for (int i = 0; i < 16; ++i) m2[i] = v[i] < v[i + 1] ? v[i] : v[i + 1];
#pragma unroll
        for (int i = 0; i < 14; ++i) m4[i] = m2[i] < m2[i + 2] ? m2[i] : m2[i + 2];
#pragma unroll
        for (int i = 0; i < 8; ++i) { m8[i] = m4[i] < m4[i + 4] ? m4[i] : m4[i + 4]; const H m = m8[i] < m2[i + 8] ? m8[i] : m2[i + 8]; if (x0 + i < npos) s_wmin[x0 + i] = (lo + x0 + i < nwin) ? m : (H)0; }
    }
    __syncthreads();
    int pend = t0 + VMX_SK_TILE; if (pend > P) pend = P;
    const int p0 = t0 + 8 * (int)threadIdx.x;               // this thread's positions p0 .. p0 + 7
    unsigned selmask = 0; H hs[8];
    if (p0 < pend) {
        const int xa = p0 - 9 - lo;                          // LDS index of window start p0 - 9 (negative: before the sequence)
        H v[17], m2[16], m4[14];
#pragma unroll
        for (int i = 0; i < 17; ++i) v[i] = (xa + i >= 0 && xa + i < npos) ? s_wmin[xa + i] : (H)0;
#pragma unroll
        for (int i = 0; i < 16; ++i) m2[i] = v[i] > v[i + 1] ? v[i] : v[i + 1];
#pragma unroll
        for (int i = 0; i < 14; ++i) m4[i] = m2[i] > m2[i + 2] ? m2[i] : m2[i + 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const H m8 = m4[j] > m4[j + 4] ? m4[j] : m4[j + 4];
            const H mx = m8 > m2[j + 8] ? m8 : m2[j + 8];          // windows p - 9 .. p
            const H h = p0 + j < pend ? s_h[p0 + j - lo] : HINF;
            hs[j] = h;
            if (h != HINF && mx == h) selmask |= 1u << j;
        }
    }
    int tot; const int ex = vmx_block_excl_scan(__popc(selmask), s_scan, &tot);
    if (selmask) {
        int o = written + ex;
#pragma unroll
        for (int j = 0; j < 8; ++j) if ((selmask >> j) & 1u) { oh[o] = (uint64_t)hs[j]; op[o] = ((uint32_t)(p0 + j) << 1) | s_z[p0 + j - lo]; ++o; }
    }
    written += tot;
    __syncthreads();
}
    return written;
}

// codes: 1 byte per base (0..4). Outputs per read at mz_off[r] (capacity = number of k-mer starts): hash, pos<<1|strand.
// H: uint32_t for 2 k <= 32 (k_sketch32: 25 KB of LDS per workgroup instead of 46), uint64_t otherwise (k_sketch).
template <typename H>
__device__ __forceinline__ void vmx_sketch_body(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, int n_reads, int k, int w,
                                                uint64_t* __restrict__ mz_hash, uint32_t* __restrict__ mz_ps, const int64_t* __restrict__ mz_off,
                                                int32_t* __restrict__ mz_cnt) {
    __shared__ uint8_t s_codes[VMX_SK_TILE + 2 * VMX_SK_HALO + 64];
    __shared__ H s_h[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ H s_wmin[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ uint8_t s_z[VMX_SK_TILE + 2 * VMX_SK_HALO];
    __shared__ int s_scan[20];
    const H mask = (H)((2 * k >= (int)(8 * sizeof(H))) ? ~(H)0 : (((H)1 << (2 * k)) - 1));
    const H HINF = ~(H)0;
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
            const int written = vmx_sketch_w10<H>(C, P, k, nwin, s_codes, s_h, s_wmin, s_z, s_scan, oh, op);
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
                H fwd = 0, rc = 0; bool ok = true;
                for (int i = 0; i < k; ++i) {
                    uint8_t c = s_codes[x + i];
                    if (c > 3) ok = false;
                    fwd = (H)((fwd << 2) | (H)(c & 3));
                    rc = (H)((rc >> 2) | ((H)(3 - (c & 3)) << shift));
                }
                H h = HINF; uint8_t z = 0;
                if (ok && fwd != rc) { z = rc < fwd ? 1 : 0; h = vmx_hash_t<H>((H)(fwd & mask) < rc ? (H)(fwd & mask) : rc, mask); }
                s_h[x] = h; s_z[x] = z;
            }
            __syncthreads();
            // window minima for window starts a in [lo, min(hi, nwin)): min over [a, a+wl)
            for (int x = (int)threadIdx.x; x < npos; x += (int)blockDim.x) {
                int a = lo + x; H m = HINF;
                if (a < nwin) { for (int j = 0; j < wl; ++j) { int y = x + j; if (y < npos) { H v = s_h[y]; m = v < m ? v : m; } } }
                s_wmin[x] = m;   // windows whose span leaves [lo,hi) are never consulted for this tile's positions
            }
            __syncthreads();
            // selection + ordered compaction for p in [t0, min(t0+TILE, P))
            int pend = t0 + VMX_SK_TILE; if (pend > P) pend = P;
            for (int pb = t0; pb < pend; pb += (int)blockDim.x) {
                int p = pb + (int)threadIdx.x;
                int sel = 0; H h = HINF; uint8_t z = 0;
                if (p < pend) {
                    int x = p - lo; h = s_h[x]; z = s_z[x];
                    if (h != HINF) {
                        int a0 = p - wl + 1; if (a0 < 0) a0 = 0;
                        int a1 = p; if (a1 > nwin - 1) a1 = nwin - 1;
                        for (int a = a0; a <= a1; ++a) if (s_wmin[a - lo] == h) { sel = 1; break; }
                    }
                }
                int tot; int ex = vmx_block_excl_scan(sel, s_scan, &tot);
                if (sel) { oh[written + ex] = (uint64_t)h; op[written + ex] = ((uint32_t)p << 1) | z; }
                written += tot;
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) mz_cnt[r] = written;
    }
}

__global__ void __launch_bounds__(256) k_sketch(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, int n_reads, int k, int w,
                                                uint64_t* __restrict__ mz_hash, uint32_t* __restrict__ mz_ps, const int64_t* __restrict__ mz_off,
                                                int32_t* __restrict__ mz_cnt) {
    vmx_sketch_body<uint64_t>(codes, roff, n_reads, k, w, mz_hash, mz_ps, mz_off, mz_cnt);
}
__global__ void __launch_bounds__(256) k_sketch32(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, int n_reads, int k, int w,
                                                  uint64_t* __restrict__ mz_hash, uint32_t* __restrict__ mz_ps, const int64_t* __restrict__ mz_off,
                                                  int32_t* __restrict__ mz_cnt) {
    vmx_sketch_body<uint32_t>(codes, roff, n_reads, k, w, mz_hash, mz_ps, mz_off, mz_cnt);
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
        // a wavefront takes 64 minimizers at a time and copies their hits 64 at a time: hit j of the group belongs to the first lane whose
        // running count passes j (a six-step search over the lanes' counts), so every lane has a hit to copy and the 64 keys go out as one
        // contiguous store. (One thread per minimizer copied its own run: a wave took as long as its most frequent minimizer — tens of
        // positions against an average of four.)
        const int lane = vmx_lane(), wv = (int)(threadIdx.x >> 6), nwv = (int)(blockDim.x >> 6);
        for (int m0 = wv * 64; m0 < M; m0 += nwv * 64) {
            const int m = m0 + lane;
            int cn = 0, hoff = 0, start = 0, ps = 0;
            if (m < M) { cn = (int)m_cnt[base + m]; hoff = (int)m_hoff[base + m]; start = (int)m_start[base + m]; ps = (int)mz_ps[base + m]; }
            const int inc = vmx_wave_incl_scan_i32(cn);
            const int tot = __shfl(inc, 63);
            const int ex = inc - cn;
            const int hoff0 = __shfl(hoff, 0);                       // the group's hits are contiguous in the read's key array
            for (int j0 = 0; j0 < tot; j0 += 64) {
                const int j = j0 + lane;
                int lo = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) { const int v = __shfl(inc, lo + step - 1); if (v <= j) lo += step; }
                const int ow = lo < 64 ? lo : 63;
                const int o_ex = __shfl(ex, ow), o_start = __shfl(start, ow), o_ps = __shfl(ps, ow);
                if (j < tot) {
                    const uint64_t pv = idx_pos[(uint32_t)o_start + (uint32_t)(j - o_ex)];
                    const uint64_t q = (uint32_t)o_ps >> 1; const uint32_t zq = (uint32_t)o_ps & 1u;
                    const uint64_t sb = ((uint32_t)(pv & 1) == zq) ? 1ULL : 0ULL;
                    K[hoff0 + j] = ((pv >> 1) << 28) | (q << 1) | sb;
                }
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
// ---- k_cluster_big, filtered form ------------------------------------------------------------------------------------------------
// A 15 kb read meets ~11 k hits in an hg38-size index, of which ~10 k are isolated random hits: no other hit within 5000 bp, so each can
// only ever be a cluster of one, and only the first few by reference position can reach the output (they rank behind every larger
// cluster). Sorting them is wasted work. Here a 2-bit-per-slot filter over 8192-bp reference bins (2^18 slots in LDS: "one hit" / "two or
// more") finds the hits that have a neighbour in their own or an adjacent bin — a superset of every hit with another hit within 5000 bp, so
// every cluster of two or more lies wholly inside it — and only those CANDIDATES (~1.5-2 k) are sorted and cut into clusters, in LDS. If fewer
// than check_num clusters of two or more exist, the missing ones are the singletons with the smallest reference positions (candidate
// clusters of one and isolated hits alike): their cut-off comes from a three-pass radix select, no sort. The result is VMX-S1's, hit for hit.
// Returns false (nothing written) when the read does not fit the scheme; the caller then takes the general path.
#ifdef VMX_EMU
#define VMX_CF_SLOT_BITS 15                 // emulator build: a layout that fits its 8192-key tile, so that the CPU tests run this path
#define VMX_CF_CAND 2048
#define VMX_CFB_SLOT_BITS 15
#else
#define VMX_CF_SLOT_BITS 18
#define VMX_CF_CAND 4096
#define VMX_CFB_SLOT_BITS 19
#endif
#define VMX_CF_F_U64 (1 << (VMX_CF_SLOT_BITS - 5))                       /* filter: 2 bits per slot */
#define VMX_CF_CS_U64 ((VMX_CF_CAND + 512) / 4)                          /* cluster starts, uint16 each */
/* the filter and the arrays of the later phases share the tile: the filter's verdict on every hit goes to a byte array in HBM, the candidate keys
   pass through HBM once, and everything after the filter phase (candidates, cluster starts, histogram, selection) overlays the filter's 64 KB.
   64 KB instead of 121 KB per workgroup: the kernel no longer needs a CU's LDS to itself (VMX_CLUSTER_BIG=0, the 32 KB kernel without this
   form, timed the same in the pipeline as the 121 KB form — what the filtered form saved alone it lost by keeping k_local_seed off its CUs) */
#define VMX_CF_REST_U64 (VMX_CF_CAND + VMX_CF_CS_U64 + 1024 + 1024)
#define VMX_CF_TOTAL_U64 (VMX_CF_F_U64 > VMX_CF_REST_U64 ? VMX_CF_F_U64 : VMX_CF_REST_U64)
/* the LONG form (reads of more than 16383 hits, inside k_cluster_gen, a CU's LDS to itself): a filter of 2^19 slots in the whole 128 KB tile,
   up to tile - 1024 candidates (15360) sorted through the tile like any one-tile sort of the general path and then read from HBM (their copy lives in
   the cl_keys scratch), the small arrays of the later phases at the start of the tile. A 40 kb read: 30 k hits, ~8 k candidates, one 8192-key tile
   sort instead of the general path's two 16384-key tiles and their HBM steps; reads of more than 65535 hits or 15360 candidates keep the general path. */
#define VMX_CFB_CAP (VMX_SORT_LDS_BIG - 1024)
template <int SB>
__device__ __forceinline__ bool vmx_cf_is_cand(const uint32_t* F, uint64_t key) {
    const unsigned slot = (unsigned)((key >> 28) >> 13) & ((1u << SB) - 1u);
    const unsigned lo = (slot - 1u) & ((1u << SB) - 1u), hi = (slot + 1u) & ((1u << SB) - 1u);
    return ((F[slot >> 4] >> (2 * (slot & 15))) & 2u) || ((F[lo >> 4] >> (2 * (lo & 15))) & 1u) || ((F[hi >> 4] >> (2 * (hi & 15))) & 1u);
}
#ifdef VMX_EMU
// emulator-only test hook: how many reads the filtered form answered / handed back to the general path
static int g_vmx_cf_taken = 0, g_vmx_cf_declined = 0;
extern "C" int vmx_emu_cf_count(int which) { return which ? g_vmx_cf_declined : g_vmx_cf_taken; }
#endif
#ifdef VMX_CF_TICKS
__device__ unsigned long long g_vmx_cf_ticks[24];
extern "C" void vmx_cf_ticks_dump() { unsigned long long h[24]; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_vmx_cf_ticks), sizeof h);
    fprintf(stderr, "cf ticks:"); for (int i = 0; i < 24; ++i) fprintf(stderr, " %llu", h[i]); fprintf(stderr, "\n"); }
#define VMX_CFT(ph) do { __syncthreads(); if (threadIdx.x == 0) { long long t1_ = VMX_CLOCK(); atomicAdd(&g_vmx_cf_ticks[ph], (unsigned long long)(t1_ - cft0)); cft0 = t1_; } } while (0)
#else
#define VMX_CFT(ph) do { } while (0)
#endif
#ifndef VMX_CF_WAVES
#define VMX_CF_WAVES 8                      // waves per SIMD the filtered kernel is built for: 8 = two 1024-thread workgroups per CU
#endif
template <int BLOCK, bool LONG>
__device__ __forceinline__ bool vmx_cluster_filtered(const uint64_t* __restrict__ K, int n, int check_num, int kmer, uint64_t* s_sort, int tile, int* s_scan,
                                                     int64_t* __restrict__ out, uint64_t* __restrict__ CK, int32_t* __restrict__ n_anchors_r) {
    __shared__ int s_cf[8];
    constexpr int SB = LONG ? VMX_CFB_SLOT_BITS : VMX_CF_SLOT_BITS;
    constexpr int CAP = LONG ? VMX_CFB_CAP : VMX_CF_CAND;        // candidates the form can take
    constexpr int HB = LONG ? VMX_CFB_CAP : 0x2000;              // handles of the rank keys: below HB a start among the candidates, from HB on a slot of ISO
    constexpr int CS_U64 = (CAP + 512) / 4;
    uint32_t* F = (uint32_t*)s_sort;                              // 64 KB (LONG: 128 KB): 2 bits per slot (filter phase only)
    uint64_t* CAND = s_sort;                                      // 32 KB: candidate keys, sorted in place (over the filter, once it is done); LONG: not used
    uint64_t* GC = LONG ? CK : (uint64_t*)out;                    // HBM: candidate keys — in the read's (still unused) output rows, LONG: in its cl_keys scratch, where they stay
    uint8_t* FLAG = (uint8_t*)out + 8 * (size_t)n;                // ... and the filter's verdict per hit (1 = candidate); 9 n of the 32 n bytes
    uint64_t* REST = LONG ? s_sort : CAND + CAP;                  // the later phases' arrays
    uint16_t* CS = (uint16_t*)REST;                               // 9 KB: first hit of every candidate cluster (+ end); later the output offsets
    uint32_t* HIST = (uint32_t*)(REST + CS_U64);                  // 8 KB: radix-select / size histogram; later the isolated keys that were selected
    uint64_t* ISO = (uint64_t*)HIST;
    uint64_t* SEL = REST + CS_U64 + 1024;                         // 8 KB: rank keys of the selected clusters
    (void)tile; (void)CAND;
    const int tid = (int)threadIdx.x;
#ifdef VMX_CF_TICKS
    long long cft0 = VMX_CLOCK();
#endif
    if (n > (LONG ? 0xffff : 0x3fff)) return false;               // (the radix select counts in 16 bits; the host sends reads of more than 16383 hits to the LONG form)
    { struct alignas(16) q128 { uint64_t a, b; }; for (int i = tid; i < (1 << SB) / 64; i += BLOCK) ((q128*)F)[i] = q128{0ULL, 0ULL}; }   // 16-byte stores
    if (tid == 0) { s_cf[0] = 0; s_cf[3] = 0; }
    __syncthreads();
    for (int i0 = tid; i0 < n; i0 += 4 * BLOCK) {                // four keys on their way per thread before the first atomic
        uint64_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * BLOCK; kk[u] = K[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * BLOCK < n) {
            const unsigned slot = (unsigned)((kk[u] >> 28) >> 13) & ((1u << SB) - 1u);
            const unsigned sh = 2 * (slot & 15);
            const unsigned old = atomicOr(&F[slot >> 4], 1u << sh);
            if ((old >> sh) & 1u) atomicOr(&F[slot >> 4], 2u << sh);
        }
    }
    __syncthreads();
    VMX_CFT(0);
    for (int i = tid; i < n; i += BLOCK) {
        const uint64_t key = K[i];
        const bool cnd = vmx_cf_is_cand<SB>(F, key);
        FLAG[i] = cnd ? 1 : 0;
        if (cnd) { const int p = atomicAdd(&s_cf[0], 1); if (p < CAP) GC[p] = key; }
    }
    __syncthreads();
    const int ncand = s_cf[0];
    __syncthreads();
    VMX_CFT(1);
    if (ncand > CAP) return false;
    int NC = 1; while (NC < ncand) NC <<= 1;
    if constexpr (LONG) {
        // the filter is dead from here on: the tile sorts the candidates (where they are, in HBM), then holds the later phases' arrays
        for (int i = ncand + tid; i < NC; i += BLOCK) GC[i] = VMX_INF64;
        __syncthreads();
        VMX_CFT(2);
        // (eight keys per thread: the sixteen-key form of the general path needs its 128 registers)
        const bool rb = NC >= 64;
        for (int i = tid; i < NC; i += BLOCK) s_sort[rb ? vmx_sw(i) : i] = GC[i];
        __syncthreads();
        if (rb) vmx_bitonic_tile_sw_t<3>(s_sort, NC, 0, NC); else if (NC > 1) vmx_block_bitonic_passes(s_sort, NC);
        __syncthreads();
        for (int i = tid; i < NC; i += BLOCK) GC[i] = s_sort[rb ? vmx_sw(i) : i];
        __syncthreads();
    } else {
        for (int i = tid; i < ncand; i += BLOCK) CAND[i] = GC[i]; // the filter is dead from here on: its LDS holds the later phases' arrays
        __syncthreads();
        for (int i = ncand + tid; i < NC; i += BLOCK) CAND[i] = VMX_INF64;
        __syncthreads();
        VMX_CFT(2);
        if (NC >= 4) vmx_rank_merge_sort_lds(CAND, CAND + CAP, NC);   // (the second buffer lies over CS / HIST / SEL, which are not in use yet)
        else if (NC > 1) vmx_block_bitonic_passes(CAND, NC);
    }
#define VMX_CF_C(i) (LONG ? GC[i] : CAND[i])
    VMX_CFT(3);
    // candidate clusters: starts in order (CS), then sizes by difference
    int ncl = 0;
    for (int i0 = 0; i0 < ncand; i0 += BLOCK) {
        const int i = i0 + tid;
        int f = 0;
        if (i < ncand) f = (i == 0) || ((long long)(VMX_CF_C(i) >> 28) - (long long)(VMX_CF_C(i - 1) >> 28) > 5000);
        int tot; const int ex = vmx_block_excl_scan(f, s_scan, &tot);
        if (f) CS[ncl + ex] = (uint16_t)i;
        ncl += tot;
        __syncthreads();
    }
    if (tid == 0) CS[ncl] = (uint16_t)ncand;
    for (int i = tid; i < 2048; i += BLOCK) HIST[i] = 0u;
    __syncthreads();
    VMX_CFT(4);
    // clusters of two or more: how many, and the histogram of their sizes (bin 1023 = larger)
    int m2 = 0;
    for (int c = tid; c < ncl; c += BLOCK) { const int sz = (int)CS[c + 1] - (int)CS[c]; if (sz >= 2) { ++m2; atomicAdd(&HIST[sz < 1023 ? sz : 1023], 1u); } }
    { int tot; (void)vmx_block_excl_scan(m2, s_scan, &tot); m2 = tot; }
    __syncthreads();
    const int niso = n - ncand;                                   // isolated hits: singletons by construction
    const int nsingle = (ncl - m2) + niso;
    int nsel = 0;
    VMX_CFT(5);
#ifdef VMX_CF_TICKS
    if (tid == 0) { atomicAdd(&g_vmx_cf_ticks[m2 >= check_num ? 20 : 21], 1ULL); atomicAdd(&g_vmx_cf_ticks[22], (unsigned long long)n); atomicAdd(&g_vmx_cf_ticks[23], (unsigned long long)ncand); }
#endif
    if (m2 >= check_num) {
        // the cut falls among the clusters of two or more: size s* of the check_num-th, everything larger, and the first of size s* in reference order
        vmx_hist_cut(HIST, 2, check_num, s_scan, &s_cf[1]);
        const int sstar = s_cf[1], need = check_num - s_cf[2];
        __syncthreads();
        VMX_CFT(6);
        if (sstar >= 1023 || sstar < 2) return false;
        int neq = 0;
        for (int c0 = 0; c0 < ncl; c0 += BLOCK) {
            const int c = c0 + tid;
            int sz = 0, st = 0;
            if (c < ncl) { st = CS[c]; sz = (int)CS[c + 1] - st; }
            const int eq = sz == sstar;
            int toteq; const int exeq = vmx_block_excl_scan(eq, s_scan, &toteq);
            const int take = sz > sstar || (eq && neq + exeq < need);
            __syncthreads();
            int tott; const int ext = vmx_block_excl_scan(take, s_scan, &tott);
            if (take) SEL[nsel + ext] = ((uint64_t)(0x3fffu - (unsigned)sz) << 50) | ((VMX_CF_C(st) >> 28) << 14) | (uint64_t)st;
            nsel += tott; neq += toteq;
            __syncthreads();
        }
        VMX_CFT(7);
    } else {
        // every cluster of two or more is taken; the rest are the R singletons with the smallest reference positions
        for (int c0 = 0; c0 < ncl; c0 += BLOCK) {
            const int c = c0 + tid;
            int sz = 0, st = 0;
            if (c < ncl) { st = CS[c]; sz = (int)CS[c + 1] - st; }
            if (sz > 0x3fff) s_cf[3] = 1;
            const int take = sz >= 2;
            int tott; const int ext = vmx_block_excl_scan(take, s_scan, &tott);
            if (take) SEL[nsel + ext] = ((uint64_t)(0x3fffu - (unsigned)sz) << 50) | ((VMX_CF_C(st) >> 28) << 14) | (uint64_t)st;
            nsel += tott;
            __syncthreads();
        }
        VMX_CFT(8);
        int R = check_num - m2; if (R > nsingle) R = nsingle;
        uint64_t T = ~0ULL;                                       // singletons with reference position <= T are taken
        if (R < nsingle) {
            // radix select of the R-th smallest position: 12 bits per pass, most significant first; positions of singletons are distinct
            uint64_t prefix = 0; int rem = R;
            for (int shift = 24; shift >= 0; shift -= 12) {
                for (int i = tid; i < 2048; i += BLOCK) HIST[i] = 0u;         // 4096 16-bit counters (n <= 16384)
                __syncthreads();
                const uint64_t himask = shift == 24 ? 0ULL : (~0ULL << (shift + 12));
                for (int c = tid; c < ncl; c += BLOCK) if ((int)CS[c + 1] - (int)CS[c] == 1) {
                    const uint64_t v = VMX_CF_C(CS[c]) >> 28;
                    if ((v & himask) == prefix) { const unsigned b = (unsigned)(v >> shift) & 0xfffu; atomicAdd(&HIST[b >> 1], 1u << (16 * (b & 1))); }
                }
                for (int i = tid; i < n; i += BLOCK) {
                    const uint64_t key = K[i];
                    if (!FLAG[i]) { const uint64_t v = key >> 28; if ((v & himask) == prefix) { const unsigned b = (unsigned)(v >> shift) & 0xfffu; atomicAdd(&HIST[b >> 1], 1u << (16 * (b & 1))); } }
                }
                __syncthreads();
                // bins 4 tid .. 4 tid + 3 per thread; the bin where the running count reaches `rem`
                constexpr int PER = 4096 / BLOCK > 0 ? 4096 / BLOCK : 1;          // bins per thread (BLOCK * PER >= 4096)
                int cnt[PER]; int sum = 0;
                for (int j = 0; j < PER; ++j) { const unsigned b = (unsigned)PER * (unsigned)tid + (unsigned)j; cnt[j] = b < 4096u ? (int)((HIST[b >> 1] >> (16 * (b & 1))) & 0xffffu) : 0; sum += cnt[j]; }
                int tot; int ex = vmx_block_excl_scan(sum, s_scan, &tot);
                for (int j = 0; j < PER; ++j) { if (ex < rem && rem <= ex + cnt[j]) { s_cf[4] = PER * tid + j; s_cf[5] = ex; } ex += cnt[j]; }
                __syncthreads();
                prefix |= (uint64_t)(unsigned)s_cf[4] << shift; rem -= s_cf[5];
                __syncthreads();
            }
            T = prefix;
        }
        VMX_CFT(9);
        // collect the singletons at or below T: candidate clusters of one (handle = start in CAND) and isolated hits (handle = slot in ISO)
        if (tid == 0) s_cf[6] = 0;
        __syncthreads();
        for (int c0 = 0; c0 < ncl; c0 += BLOCK) {
            const int c = c0 + tid;
            int take = 0, st = 0;
            if (c < ncl && (int)CS[c + 1] - (int)CS[c] == 1) { st = CS[c]; take = (VMX_CF_C(st) >> 28) <= T; }
            int tott; const int ext = vmx_block_excl_scan(take, s_scan, &tott);
            if (take) SEL[nsel + ext] = ((uint64_t)(0x3fffu - 1u) << 50) | ((VMX_CF_C(st) >> 28) << 14) | (uint64_t)st;
            nsel += tott;
            __syncthreads();
        }
        for (int i = tid; i < n; i += BLOCK) {
            const uint64_t key = K[i];
            if (!FLAG[i] && (key >> 28) <= T) {
                const int q = atomicAdd(&s_cf[6], 1);
                if (nsel + q < 1024) { ISO[q] = key; SEL[nsel + q] = ((uint64_t)(0x3fffu - 1u) << 50) | ((key >> 28) << 14) | (uint64_t)(HB + q); }
            }
        }
        __syncthreads();
        nsel += s_cf[6];
        __syncthreads();
        if (nsel > 1024 || s_cf[3]) return false;
        VMX_CFT(10);
    }
    if (nsel > check_num) return false;                           // (cannot happen; the general path is the safe answer to a broken invariant)
    int NS = 1; while (NS < nsel) NS <<= 1;
    for (int i = nsel + tid; i < NS; i += BLOCK) SEL[i] = VMX_INF64;
    __syncthreads();
    if (NS >= 4 && NS <= CS_U64) vmx_rank_merge_sort_lds(SEL, (uint64_t*)CS, NS);     // (CS is done with: its slots are the second buffer)
    else if (NS > 1) vmx_block_bitonic_passes(SEL, NS);
    __syncthreads();
    VMX_CFT(11);
    // emit cluster by cluster in rank order: output offsets (over CS, which is done with), then every thread copies hits
    uint32_t* OFF = (uint32_t*)CS;
    int total = 0;
    for (int c0 = 0; c0 < nsel; c0 += BLOCK) {
        const int c = c0 + tid;
        int sz = 0;
        if (c < nsel) sz = 0x3fff - (int)(SEL[c] >> 50);
        int tot; const int ex = vmx_block_excl_scan(sz, s_scan, &tot);
        if (c < nsel) OFF[c] = (uint32_t)(total + ex);
        total += tot;
        __syncthreads();
    }
    VMX_CFT(12);
    for (int e = tid; e < total; e += BLOCK) {
        int lo = 0, hi = nsel;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)OFF[mid] <= e) lo = mid; else hi = mid; }
        const uint64_t sk = SEL[lo]; const int h = (int)(sk & 0x3fffu);
        const uint64_t hk = h >= HB ? ISO[h - HB] : VMX_CF_C(h + (e - (int)OFF[lo]));
        int64_t* o = out + 4 * (int64_t)e;
        o[0] = (int64_t)((hk >> 1) & 0x7ffffffULL); o[1] = (int64_t)(hk >> 28); o[2] = (hk & 1) ? 1 : -1; o[3] = kmer;
    }
    if (tid == 0) *n_anchors_r = total;
    __syncthreads();
    VMX_CFT(13);
    return true;
#undef VMX_CF_C
}

// FORM 0: the general path. FORM 1: the filtered form alone; a read it declines goes on the `decl` list (count in *n_decl) for a later launch
// of the general path — the two forms in one kernel cost the filtered one the general path's 128 registers, i.e. one workgroup per CU.
template <int BLOCK, int FORM>
__device__ __forceinline__ void vmx_cluster_body(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                 const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, int tile, int check_num, int kmer,
                                                 int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors, int32_t* __restrict__ decl = nullptr, int32_t* __restrict__ n_decl = nullptr) {
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
        if constexpr (FORM == 1) {
            // (uniform: every thread sees the same n and gets the same answer)
            const bool cf_done = tile >= VMX_CF_TOTAL_U64 && check_num > 0 && check_num <= 1024 &&
                                 vmx_cluster_filtered<BLOCK, false>(K, n, check_num, kmer, s_sort, tile, s_scan, rows + 4 * key_off[r], CK, &n_anchors[r]);
#ifdef VMX_EMU
            if (threadIdx.x == 0) ++(cf_done ? g_vmx_cf_taken : g_vmx_cf_declined);
#endif
            if (!cf_done && threadIdx.x == 0) decl[atomicAdd(n_decl, 1)] = r;
            __syncthreads();
            continue;
        }
        if constexpr (FORM == 3) {
            // the LONG filtered form alone (reads of up to 65535 hits, up to 15360 candidates); what it declines goes on the list for the general path
            const bool cf_done = tile >= VMX_SORT_LDS_BIG && check_num > 0 && check_num <= 1024 &&
                                 vmx_cluster_filtered<BLOCK, true>(K, n, check_num, kmer, s_sort, tile, s_scan, rows + 4 * key_off[r], CK, &n_anchors[r]);
#ifdef VMX_EMU
            if (threadIdx.x == 0) ++(cf_done ? g_vmx_cf_taken : g_vmx_cf_declined);
#endif
            if (!cf_done && threadIdx.x == 0) decl[atomicAdd(n_decl, 1)] = r;
            __syncthreads();
            continue;
        }
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
            __shared__ int s_cut[2];                                  // s*, and the clusters strictly larger than s* (1023: the overflow bin holds the cut)
            vmx_hist_cut((const uint32_t*)hist, 1, check_num, s_scan, s_cut);
            const int sstar = s_cut[0], above = s_cut[1];
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
    vmx_cluster_body<256, 0>(keys, cl_keys, key_off, nhits, rlist, nlist, tile, check_num, kmer, rows, n_anchors);
}
// reads with more hits than k_cluster's tile: the filtered form, two 1024-thread workgroups per CU (64 KB of LDS, 64 registers) ...
#ifndef VMX_CF_BLOCK
#define VMX_CF_BLOCK 1024                   // threads of the filtered kernel's workgroup (tuning knob, with the launch in vmx_index.hip)
#endif
__global__ void __launch_bounds__(VMX_CF_BLOCK, VMX_CF_WAVES) k_cluster_big(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                      const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, int tile, int check_num, int kmer,
                                                      int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors, int32_t* __restrict__ decl, int32_t* __restrict__ n_decl) {
    vmx_cluster_body<VMX_CF_BLOCK, 1>(keys, cl_keys, key_off, nhits, rlist, nlist, tile, check_num, kmer, rows, n_anchors, decl, n_decl);
}
// ... and the general path for the reads that form declines or cannot take (more than 16383 hits); nlist_dev: the list's length, on the device
__global__ void __launch_bounds__(1024) k_cluster_gen(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                      const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, const int32_t* __restrict__ nlist_dev,
                                                      int tile, int check_num, int kmer, int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors) {
    vmx_cluster_body<1024, 0>(keys, cl_keys, key_off, nhits, rlist, nlist_dev ? *nlist_dev : nlist, tile, check_num, kmer, rows, n_anchors);
}
// the LONG filtered form (reads of more than 16383 hits, and the reads k_cluster_big declines): one workgroup per CU (128 KB of LDS) at 64
// registers, so that the CU's other half of the register file stays free for LDS-less kernels of the other batches (the gap fill)
__global__ void __launch_bounds__(1024, VMX_CF_WAVES) k_cluster_long(uint64_t* __restrict__ keys, uint64_t* __restrict__ cl_keys, const int64_t* __restrict__ key_off,
                                                      const int64_t* __restrict__ nhits, const int32_t* __restrict__ rlist, int nlist, const int32_t* __restrict__ nlist_dev,
                                                      int tile, int check_num, int kmer, int64_t* __restrict__ rows, int32_t* __restrict__ n_anchors,
                                                      int32_t* __restrict__ decl, int32_t* __restrict__ n_decl) {
    vmx_cluster_body<1024, 3>(keys, cl_keys, key_off, nhits, rlist, nlist_dev ? *nlist_dev : nlist, tile, check_num, kmer, rows, n_anchors, decl, n_decl);
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
    // tiles of 2048 values, eight consecutive ones per thread; a wavefront scan of the threads' sums, the four wave totals through LDS
    // (this kernel walked every tile of 256 values in ONE thread: 0.4 ms per batch on the critical path of every batch)
    __shared__ long long s_w[4];
    __shared__ long long s_base;
    const int tid = (int)threadIdx.x, lane = vmx_lane(), w = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += 2048) {
        const int64_t b = i0 + 8 * tid;
        long long v[8]; long long sum = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long long x = b + u < n ? in[b + u] : 0;
            if (pow2_round) { long long N = 1; while (N < x) N <<= 1; x = x ? N : 0; }
            v[u] = x; sum += x;
        }
        long long inc = sum;
        for (int o = 1; o < 64; o <<= 1) { const long long x = __shfl_up(inc, o); if (lane >= o) inc += x; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        long long wb = 0; for (int ww = 0; ww < w; ++ww) wb += s_w[ww];
        const long long tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        long long ex = s_base + wb + inc - sum;
#pragma unroll
        for (int u = 0; u < 8; ++u) { if (b + u < n) out[b + u] = ex; ex += v[u]; }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) out[n] = s_base;
}
