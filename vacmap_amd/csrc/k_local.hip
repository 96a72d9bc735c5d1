// k_local.hip — local 9-mer re-seeding around the guide chains and the local chain DP.
//
//   k_orient        reverse-complements the reads whose global chain is on the minus strand (get_readmap_DP_test,
//                   /root/reference/src/vacmap/mammap_clrnano.py:24060-24065)
//   k_local_prep    L1 merge_chain / drop_somechains / sort by 1/len (:28482-28574): one thread per read (vmx_local.h)
//   k_local_seed    L2 get_localmap_multi_all_forDP_inv_guide_1 (:23069-23345): one workgroup per read.
//                   Python's hash(str) tables are exact 9-mer identity, so the table is an 18-bit direct-address
//                   counting sort (4^9 = 262144 buckets in HBM scratch, positions ascending inside a bucket);
//                   per read position forward + reverse-complement lookups, proximity filter against the two closest
//                   guide anchors (findClosest_1 :17560), then the sequential "flush once a run reaches 20" merge
//                   (:23232-23344) is executed per diagonal (hits sorted by diagonal key; one lane per diagonal),
//                   and the reference's append order + the stable argsort by q+l (:28585) are restored by two key sorts.
//   k_chain_local   L3/L4 LC-exact (:27305-27528) and LC-mm (:28250-28476): one wavefront per read, same 64-wide
//                   descending-S candidate scan as k_chain_global, traceback with overlap trimming (:27508-27526).
// Deviation D1: a 9-mer holding a non-ACGT base never matches.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"

#define vmx_block_sort_u64(g, N, lds) vmx_block_sort_u64_impl((g), (N), (lds), VMX_SORT_LDS)

// ------------------------------------------------------------------------------------------------ orient
__global__ void k_orient(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, const double* __restrict__ gscore, int n_reads,
                         uint8_t* __restrict__ ocodes) {
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const int64_t o = roff[r]; const int L = (int)(roff[r + 1] - o);
        const bool rev = gscore[r] < 0.0;
        for (int i = (int)threadIdx.x; i < L; i += (int)blockDim.x) {
            uint8_t c = rev ? codes[o + L - 1 - i] : codes[o + i];
            if (rev && c < 4) c = 3 - c;
            ocodes[o + i] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------ L1 prep
__global__ void k_local_prep(const vmx_anchor* __restrict__ path_rows, const int32_t* __restrict__ path_len, const int32_t* __restrict__ n_paths,
                             const int64_t* __restrict__ aoff, const double* __restrict__ gscore, int n_reads, int mode,
                             vmx_anchor* __restrict__ guide_rows, int32_t* __restrict__ guide_len, int32_t* __restrict__ n_guides_used,
                             int32_t* __restrict__ n_guides_total) {
    int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n_reads) return;
    const int64_t a0 = aoff[r];
    n_guides_used[r] = 0; n_guides_total[r] = 0;
    if (gscore[r] == 0.0 || n_paths[r] <= 0) return;
    vmx_local_prep(path_rows + a0, path_len + a0, n_paths[r], mode, guide_rows + a0, guide_len + a0, &n_guides_used[r], &n_guides_total[r]);
}

// ------------------------------------------------------------------------------------------------ L2 seeding
__device__ __forceinline__ int vmx_pos2contig(const int64_t* __restrict__ coff, int nseq, long long pos) {   // :51-59
    int pre = 0;
    for (int c = 0; c < nseq; ++c) { if (pos < coff[c]) break; pre = c; }
    return pre;
}

// findClosest_1 :17560-17582 on the guide sorted by read position (gq ascending)
__device__ __forceinline__ void vmx_find_closest(const int* gq, int n, int target, int& b0, int& b1, int& i0, int& i1) {
    if (target <= gq[0]) { b0 = b1 = gq[0] - target; i0 = i1 = 0; return; }
    if (target >= gq[n - 1]) { b0 = b1 = target - gq[n - 1]; i0 = i1 = n - 1; return; }
    int i = 0, j = n, mid = 0;
    while (i < j) {
        mid = (i + j) >> 1;
        if (gq[mid] == target) { b0 = b1 = 0; i0 = i1 = mid; return; }
        if (target < gq[mid]) j = mid; else i = mid + 1;
    }
    b0 = gq[j - 1] - target; if (b0 < 0) b0 = -b0;
    b1 = gq[j] - target; if (b1 < 0) b1 = -b1;
    i0 = j - 1; i1 = j;
}

__device__ __forceinline__ uint32_t vmx_kmer_at(const uint8_t* s, long long x, int k, bool& ok) {
    uint32_t v = 0; ok = true;
    for (int i = 0; i < k; ++i) { uint8_t c = s[x + i]; if (c > 3) ok = false; v = (v << 2) | (uint32_t)(c & 3); }
    return v;
}
__device__ __forceinline__ uint32_t vmx_kmer_rc(uint32_t fw, int k) {
    uint32_t rv = 0;
    for (int i = 0; i < k; ++i) { rv = (rv << 2) | (3 - (fw & 3)); fw >>= 2; }
    return rv;
}

__global__ void __launch_bounds__(256) k_local_seed(vmx_lseed_args A) {
    __shared__ uint64_t s_sort[VMX_SORT_LDS];
    __shared__ int s_scan[20];
    __shared__ long long s_iv[64][2];     // disjoint intervals of k-mer starts [lo, hi) in global ref coordinates
    __shared__ int s_niv;
    __shared__ int s_flag;
    __shared__ long long s_tot;
    __shared__ unsigned long long s_min, s_max;
    const int k = A.k;
    const int nkey = 1 << (2 * k);
    uint64_t* HKEY2 = A.hkey2_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int32_t* CNT = A.cnt_pool + (size_t)blockIdx.x * (size_t)(nkey + 1);
    int32_t* CUR = A.cur_pool + (size_t)blockIdx.x * (size_t)nkey;
    int64_t* TPOS = A.tpos_pool + (size_t)blockIdx.x * (size_t)A.tpos_cap;
    uint64_t* HKEY = A.hkey_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int64_t* HVAL = A.hval_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int32_t* HQ = A.hq_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int32_t* GOFF = A.goff_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int32_t* PCNT = A.pcnt_pool + (size_t)blockIdx.x * (size_t)A.pcnt_cap;
    uint64_t* GKEY = A.gkey_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    int32_t* GQ = A.gq_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    int64_t* GR = A.gr_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    for (int r = blockIdx.x; r < A.n_reads; r += gridDim.x) {
        const int ng = A.n_guides_used[r];
        const int64_t a0 = A.aoff[r];
        const uint8_t* RD = A.ocodes + A.roff[r];
        const int L = (int)(A.roff[r + 1] - A.roff[r]);
        vmx_anchor* OUT = A.la_rows + A.la_off[r];
        uint64_t* OKEY = A.la_ekey + A.la_off[r];
        vmx_anchor* SORTED = A.la_sorted + A.la_off[r];
        const int out_cap = (int)(A.la_off[r + 1] - A.la_off[r]);
        int n_out = 0;
        int status = 0;
        int gbase = 0;
        for (int g = 0; g < ng && status == 0; ++g) {
            const vmx_anchor* G = A.guide_rows + a0 + gbase;     // descending read order
            const int m = A.guide_len[a0 + g];
            gbase += m;
            int N = 1; while (N < m) N <<= 1;
            if (N > A.gkey_cap) { status = VM_READ_CAPACITY_DEV; break; }
            // --- :23095-23102 readgap
            int rg = 0;
            for (int i = 1 + (int)threadIdx.x; i < m; i += (int)blockDim.x) { int d = G[i].q - G[i - 1].q; if (d < 0) d = -d; rg = d > rg ? d : rg; }
            rg = vmx_wave_max_i32(rg);
            if (threadIdx.x == 0) s_flag = 0;
            __syncthreads();
            if (vmx_lane() == 0) atomicMax(&s_flag, rg);
            __syncthreads();
            long long readgap = (long long)s_flag + 1000; if (readgap < 5000) readgap = 5000;
            __syncthreads();
            // --- guide sorted by ref position (stable, :23103): key = r << 24 | original index
            for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) GKEY[i] = i < m ? (((uint64_t)G[i].r << 24) | (uint64_t)i) : ~0ULL;
            __syncthreads();
            if (N > 1) vmx_block_sort_u64(GKEY, N, s_sort);
            // guide in ascending read order (:23183) = reverse of the stored descending order; read positions are distinct
            for (int i = (int)threadIdx.x; i < m; i += (int)blockDim.x) { GQ[i] = G[m - 1 - i].q; GR[i] = G[m - 1 - i].r; }
            __syncthreads();
            // --- windows (serial, thread 0): :23105-23180
            if (threadIdx.x == 0) {
                int niv = 0; bool overflow = false;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    const bool split = attempt == 1;
                    niv = 0; bool retry = false;
                    long long ws = (long long)(GKEY[0] >> 24), we = ws;
                    int cur = vmx_pos2contig(A.coff, A.nseq, ws);
                    for (int i = 1; i <= m; ++i) {
                        bool close_it = true; long long rr = 0;
                        if (i < m) {
                            rr = (long long)(GKEY[i] >> 24);
                            bool same = (rr - we) < readgap;
                            if (split) same = same && (cur == vmx_pos2contig(A.coff, A.nseq, rr));
                            if (same) { we = rr; close_it = false; }
                        }
                        if (close_it) {
                            if (ws != we) {   // single-point windows are dropped (:23110, :23113)
                                int c = vmx_pos2contig(A.coff, A.nseq, ws);
                                if (c != vmx_pos2contig(A.coff, A.nseq, we)) { retry = true; break; }
                                long long cst = A.coff[c], clen = A.coff[c + 1] - cst;
                                long long lf = ws - cst < A.look_span ? ws - cst : A.look_span;
                                long long lo = ws - lf - cst, hi = we + A.look_span - cst; if (hi > clen) hi = clen;
                                long long nk = (hi - lo) - k + 1;
                                if (nk > 0) {
                                    long long a = cst + lo, b = a + nk;
                                    if (niv > 0 && a < s_iv[niv - 1][1]) { if (b > s_iv[niv - 1][1]) s_iv[niv - 1][1] = b; }
                                    else if (niv < 64) { s_iv[niv][0] = a; s_iv[niv][1] = b; ++niv; }
                                    else overflow = true;
                                }
                            }
                            if (i < m) { ws = we = rr; if (split) cur = vmx_pos2contig(A.coff, A.nseq, rr); }
                        }
                    }
                    if (!retry) break;
                }
                s_niv = niv; s_flag = overflow ? 1 : 0;
            }
            __syncthreads();
            if (s_flag) { status = VM_READ_CAPACITY_DEV; break; }
            const int niv = s_niv;
            long long nk_total = 0;
            for (int v = 0; v < niv; ++v) nk_total += s_iv[v][1] - s_iv[v][0];
            if (nk_total > A.tpos_cap) { status = VM_READ_CAPACITY_DEV; break; }
            // --- counting sort of the window k-mers by their 18-bit key
            for (int i = (int)threadIdx.x; i <= nkey; i += (int)blockDim.x) CNT[i] = 0;
            __syncthreads();
            for (int v = 0; v < niv; ++v) {
                const long long lo = s_iv[v][0], hi = s_iv[v][1];
                for (long long x = lo + threadIdx.x; x < hi; x += blockDim.x) {
                    bool ok; uint32_t km = vmx_kmer_at(A.ref, x, k, ok);
                    if (ok) atomicAdd(&CNT[km + 1], 1);
                }
            }
            __syncthreads();
            {   // CNT[i] held the count of bucket i-1; an inclusive scan turns it into the start of bucket i
                const int per = (nkey + (int)blockDim.x) / (int)blockDim.x;
                int lo = (int)threadIdx.x * per, hi = lo + per;
                if (lo > nkey + 1) lo = nkey + 1;
                if (hi > nkey + 1) hi = nkey + 1;
                int sum = 0; for (int i = lo; i < hi; ++i) sum += CNT[i];
                int tot; int ex = vmx_block_excl_scan(sum, s_scan, &tot);
                int acc = ex; for (int i = lo; i < hi; ++i) { acc += CNT[i]; CNT[i] = acc; }
                __syncthreads();
            }
            for (int i = (int)threadIdx.x; i < nkey; i += (int)blockDim.x) CUR[i] = CNT[i];
            __syncthreads();
            for (int v = 0; v < niv; ++v) {
                const long long lo = s_iv[v][0], hi = s_iv[v][1];
                for (long long x = lo + threadIdx.x; x < hi; x += blockDim.x) {
                    bool ok; uint32_t km = vmx_kmer_at(A.ref, x, k, ok);
                    if (ok) { int slot = atomicAdd(&CUR[km], 1); TPOS[slot] = x; }
                }
            }
            __syncthreads();
            for (int b = (int)threadIdx.x; b < nkey; b += (int)blockDim.x) {   // buckets were filled in arbitrary order: sort ascending
                int s = CNT[b], e = CNT[b + 1];
                for (int i = s + 1; i < e; ++i) { int64_t v = TPOS[i]; int j = i - 1; while (j >= s && TPOS[j] > v) { TPOS[j + 1] = TPOS[j]; --j; } TPOS[j + 1] = v; }
            }
            __syncthreads();
            // --- read window :23183-23191
            int readstart = GQ[0] - A.read_span; if (readstart < 0) readstart = 0;
            int readend = GQ[m - 1] + A.read_span; if (readend > L - k + 1) readend = L - k + 1;
            const int npos = readend > readstart ? readend - readstart : 0;
            if (npos > A.pcnt_cap) { status = VM_READ_CAPACITY_DEV; break; }
            // pass A: accepted hits per read position -> exclusive offsets
            long long run = 0;
            for (int p0 = 0; p0 < npos; p0 += (int)blockDim.x) {
                const int pi = p0 + (int)threadIdx.x;
                int cnt = 0;
                if (pi < npos) {
                    const int iloc = readstart + pi;
                    bool ok; uint32_t fw = vmx_kmer_at(RD, iloc, k, ok);
                    uint32_t rv = vmx_kmer_rc(fw, k);
                    if (ok && fw != rv) {
                        int b0, b1, c0, c1; vmx_find_closest(GQ, m, iloc, b0, b1, c0, c1);
                        long long interval = (long long)b0 + b1 + 500; if (interval > 2000) interval = 2000;
                        const long long ref1 = GR[c0], ref2 = GR[c1];
                        long long rgap = (long long)iloc - GQ[c0]; if (rgap < 0) rgap = -rgap;
                        for (int t = CNT[fw]; t < CNT[fw + 1]; ++t) if (vmx_local_accept(TPOS[t], ref1, ref2, interval, rgap)) ++cnt;
                        if (iloc > 0) for (int t = CNT[rv]; t < CNT[rv + 1]; ++t) if (vmx_local_accept(TPOS[t], ref1, ref2, interval, rgap)) ++cnt;
                    }
                }
                int tot; int ex = vmx_block_excl_scan(cnt, s_scan, &tot);
                if (pi < npos) PCNT[pi] = (int)(run + ex);
                run += tot;
                __syncthreads();
            }
            const long long H = run;
            long long NH = 1; while (NH < H) NH <<= 1;
            if (NH > A.hit_cap) { status = VM_READ_CAPACITY_DEV; break; }
            // pass B: hits in stream order (read pos asc; forward before reverse; ref pos asc).
            // key = (point + 2^36) << 26 | stream index, point = r - q (forward) or -(r + q) (reverse): ONE key space, like pointdict (Q1)
            for (int pi = (int)threadIdx.x; pi < npos; pi += (int)blockDim.x) {
                const int iloc = readstart + pi;
                bool ok; uint32_t fw = vmx_kmer_at(RD, iloc, k, ok);
                uint32_t rv = vmx_kmer_rc(fw, k);
                if (!ok || fw == rv) continue;
                int b0, b1, c0, c1; vmx_find_closest(GQ, m, iloc, b0, b1, c0, c1);
                long long interval = (long long)b0 + b1 + 500; if (interval > 2000) interval = 2000;
                const long long ref1 = GR[c0], ref2 = GR[c1];
                long long rgap = (long long)iloc - GQ[c0]; if (rgap < 0) rgap = -rgap;
                long long w = PCNT[pi];
                for (int t = CNT[fw]; t < CNT[fw + 1]; ++t) {
                    long long rl = TPOS[t];
                    if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) { long long point = rl - iloc; HKEY[w] = ((uint64_t)(point + (1LL << 36)) << 26) | (uint64_t)w; HVAL[w] = (rl << 1) | 1; HQ[w] = iloc; ++w; }
                }
                if (iloc > 0) for (int t = CNT[rv]; t < CNT[rv + 1]; ++t) {
                    long long rl = TPOS[t];
                    if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) { long long point = -(rl + iloc); HKEY[w] = ((uint64_t)(point + (1LL << 36)) << 26) | (uint64_t)w; HVAL[w] = (rl << 1); HQ[w] = iloc; ++w; }
                }
            }
            __syncthreads();
            // group the hits by diagonal, keeping stream order inside a diagonal: stable LSD radix sort on (point - min point)
            {
                unsigned long long mn = ~0ULL, mx = 0ULL;
                for (long long i = threadIdx.x; i < H; i += blockDim.x) { const unsigned long long pk = HKEY[i] >> 26; mn = pk < mn ? pk : mn; mx = pk > mx ? pk : mx; }
                for (int o = 32; o > 0; o >>= 1) { unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
                if (threadIdx.x == 0) { s_min = ~0ULL; s_max = 0ULL; }
                __syncthreads();
                if (vmx_lane() == 0) { atomicMin((unsigned long long*)&s_min, mn); atomicMax((unsigned long long*)&s_max, mx); }
                __syncthreads();
                int nbits = 0; { unsigned long long range = H > 0 ? s_max - s_min : 0; while (nbits < 40 && (range >> nbits)) ++nbits; }
                uint64_t* res = vmx_block_radix_sort_u64(HKEY, HKEY2, (int)H, 26, (uint64_t)s_min, nbits, (int*)s_sort, s_scan);
                if (res != HKEY) { for (long long i = threadIdx.x; i < H; i += blockDim.x) HKEY[i] = HKEY2[i]; }
                __syncthreads();
            }
            // --- run-merge per diagonal (:23232-23344): the lane whose index starts a key group walks that group.
            // pass 0 counts the anchors each group emits (-> offsets), pass 1 writes them with their emission key.
            for (int pass = 0; pass < 2 && status == 0; ++pass) {
                long long base_run = 0;
                for (long long i0 = 0; i0 < H; i0 += blockDim.x) {
                    const long long i = i0 + threadIdx.x;
                    const bool starts = i < H && (i == 0 || (HKEY[i] >> 26) != (HKEY[i - 1] >> 26));
                    int wr = 0;
                    if (starts) {
                        long long cq = 0, cr = 0, cl = 0; int cs = 0; bool have = false;
                        const uint64_t first_stream = HKEY[i] & ((1ULL << 26) - 1);
                        const uint64_t pk = HKEY[i] >> 26;
                        const int myoff = pass == 1 ? GOFF[i] : 0;
                        for (long long j = i; j < H && (HKEY[j] >> 26) == pk; ++j) {
                            const uint64_t sidx = HKEY[j] & ((1ULL << 26) - 1);
                            const long long hv = HVAL[sidx];
                            const long long refloc = hv >> 1; const int strand = (hv & 1) ? 1 : -1;
                            const long long iloc = HQ[sidx];
                            bool flush = false;
                            long long nq = 0, nr = 0, nl = 0; int ns = 0;
                            if (!have) { cq = iloc; cr = refloc; cs = strand; cl = k; have = true; }
                            else if (cq + cl >= iloc) {
                                long long bouns = iloc - (cq + cl) + k;
                                if (bouns > 0) {
                                    if (cl + bouns < 20) { if (strand == 1) { cs = 1; cl += bouns; } else { cr = refloc; cs = -1; cl += bouns; } }
                                    else { flush = true; nq = cq + cl; nl = bouns; if (strand == 1) { nr = cr + cl; ns = 1; } else { nr = refloc; ns = -1; } }
                                }
                            } else { flush = true; nq = iloc; nr = refloc; ns = strand; nl = k; }
                            if (flush) {
                                if (pass == 1) { vmx_anchor a; a.q = (int)cq; a.r = cr; a.s = (int16_t)cs; a.l = (int16_t)cl; OUT[n_out + myoff + wr] = a; OKEY[n_out + myoff + wr] = ((uint64_t)g << 28) | sidx; }
                                ++wr;
                                cq = nq; cr = nr; cs = ns; cl = nl;
                            }
                        }
                        // leftover entry: appended after every flush, in first-appearance order of the diagonal (:23343)
                        if (pass == 1) { vmx_anchor a; a.q = (int)cq; a.r = cr; a.s = (int16_t)cs; a.l = (int16_t)cl; OUT[n_out + myoff + wr] = a; OKEY[n_out + myoff + wr] = ((uint64_t)g << 28) | (1ULL << 26) | first_stream; }
                        ++wr;
                    }
                    if (pass == 0) {
                        int tot; int ex = vmx_block_excl_scan(wr, s_scan, &tot);
                        if (starts) GOFF[i] = (int)(base_run + ex);
                        base_run += tot;
                        __syncthreads();
                    }
                }
                if (pass == 0) {
                    if (threadIdx.x == 0) s_tot = base_run;
                    __syncthreads();
                    if (n_out + s_tot > out_cap) status = VM_READ_CAPACITY_DEV;
                    __syncthreads();
                }
            }
            if (status) break;
            n_out += (int)s_tot;
            __syncthreads();
        }
        // --- restore the reference's append order (emission key), then the stable argsort by q+l (:28585)
        if (status == 0 && n_out > 0) {
            long long NO = 1; while (NO < n_out) NO <<= 1;
            if (NO > A.hit_cap) status = VM_READ_CAPACITY_DEV;
            else {
                for (long long i = threadIdx.x; i < NO; i += blockDim.x) HKEY[i] = i < n_out ? ((OKEY[i] << 32) | (uint64_t)i) : ~0ULL;
                __syncthreads();
                if (NO > 1) vmx_block_sort_u64(HKEY, (int)NO, s_sort);
                __syncthreads();
                // rank e -> anchor index; second key = (q+l) << 32 | e
                for (long long e = threadIdx.x; e < n_out; e += blockDim.x) { int idx = (int)(HKEY[e] & 0xffffffffu); GOFF[e] = idx; }
                __syncthreads();
                for (long long e = threadIdx.x; e < NO; e += blockDim.x) {
                    uint64_t kk = ~0ULL;
                    if (e < n_out) { const vmx_anchor a = OUT[GOFF[e]]; kk = ((uint64_t)(uint32_t)(a.q + a.l) << 32) | (uint64_t)e; }
                    HKEY[e] = kk;
                }
                __syncthreads();
                if (NO > 1) vmx_block_sort_u64(HKEY, (int)NO, s_sort);
                __syncthreads();
                for (long long x = threadIdx.x; x < n_out; x += blockDim.x) SORTED[x] = OUT[GOFF[(int)(HKEY[x] & 0xffffffffu)]];
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) { A.la_cnt[r] = status ? 0 : n_out; A.status[r] = status; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ L3 / L4 local chain DP
// :13229-13265 literal
__device__ __forceinline__ int vmx_smallorequal(const double* arr, double target, int n, const int* point) {
    if (target < arr[point[0]]) return -1;
    if (target >= arr[point[n - 1]]) return n - 1;
    int i = 0, j = n, mid = 0;
    while (i < j) {
        mid = (i + j) >> 1;
        double am = arr[point[mid]];
        if (target == am) {
            if (mid < n - 1) { if (arr[point[mid + 1]] > target) return mid; else i = mid + 1; }
            else return mid;
        } else if (target < am) {
            if (mid > 0 && target >= arr[point[mid - 1]]) return mid - 1;
            j = mid;
        } else {
            if (mid < n - 1 && target < arr[point[mid + 1]]) return mid;
            i = mid + 1;
        }
    }
    return mid;
}

__device__ __forceinline__ void vmx_sarg_insert_l(int* SA, int loc, int k, int lane) {
    for (int hi = k; hi > loc; hi -= 64) {
        int x = hi - lane; int v = 0;
        if (x > loc) v = SA[x - 1];
        __syncthreads();
        if (x > loc) SA[x] = v;
        __syncthreads();
    }
    if (lane == 0) SA[loc] = k;
    __syncthreads();
}


__global__ void __launch_bounds__(64) k_chain_local(const vmx_anchor* __restrict__ anchors, const int64_t* __restrict__ la_off,
                                                    const int32_t* __restrict__ la_cnt, const int32_t* __restrict__ n_guides_total,
                                                    const int32_t* __restrict__ rlist, int nlist, int lds_cap, vmx_tables tab,
                                                    const double* __restrict__ gapcost_list, double skip_exact, double skip_mm, int maxdiff,
                                                    int maxgap, int mode, double* __restrict__ S_pool, int32_t* __restrict__ P_pool,
                                                    int32_t* __restrict__ SA_pool, double* __restrict__ out_score,
                                                    vmx_anchor* __restrict__ out_chain, int32_t* __restrict__ out_len, int32_t* __restrict__ out_variant,
                                                    int32_t* __restrict__ status) {
    VMX_DYN_SHARED(char, smem);
    __shared__ double s_gapcost[64];
    const int lane = vmx_lane();
    for (int x = lane; x <= maxdiff && x < 64; x += 64) s_gapcost[x] = gapcost_list[x];
    __syncthreads();
    const long long extra_size = (long long)tab.extra_n - 1;
    const long long l2c_size = (long long)tab.log2cache_n - 1;
    for (int li_ = blockIdx.x; li_ < nlist; li_ += gridDim.x) {
        const int rd = rlist[li_];
        const int64_t a0 = la_off[rd];
        const int n = la_cnt[rd];
        if (n <= 0) { if (lane == 0) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_RAISED_DEV; } continue; }   // np.array([]) indexing raises
        const bool mm = n_guides_total[rd] > 1;
        const double skipcost = mm ? skip_mm : skip_exact;
        const float* rgc = mm ? tab.large_readgap : (mode == 3 ? tab.readgap_r : tab.readgap_h);
        const vmx_anchor* A = anchors + a0;
        double* S; int* P; int* SA; int* Q; long long* R; int* LS;
        const bool in_lds = n <= lds_cap;
        if (in_lds) { S = (double*)smem; R = (long long*)(S + lds_cap); Q = (int*)(R + lds_cap); LS = Q + lds_cap; P = LS + lds_cap; SA = P + lds_cap; }
        else { S = S_pool + a0; P = P_pool + a0; SA = SA_pool + a0; Q = nullptr; R = nullptr; LS = nullptr; }
        if (in_lds) for (int i = lane; i < n; i += 64) { vmx_anchor a = A[i]; Q[i] = a.q; R[i] = a.r; LS[i] = ((int)a.l & 0xffff) | ((int)a.s << 16); }
        __syncthreads();
#define AQ(i) (in_lds ? Q[i] : A[i].q)
#define AR(i) (in_lds ? R[i] : (long long)A[i].r)
#define AL(i) (in_lds ? (LS[i] & 0xffff) : (int)A[i].l)
#define AS(i) (in_lds ? (LS[i] >> 16) : (int)A[i].s)
        long long prereadloc = (long long)AQ(0) + AL(0);
        int testspace_en = 1;
        if (lane == 0) { SA[0] = 0; S[0] = (double)AL(0); P[0] = VMX_NOPRE; }
        __syncthreads();
        double g_max_scores = (double)AL(0); int g_max_index = 0;
        long long opcount = 0;
        bool need_fast = false;
        for (int i = 1; i < n; ++i) {
            const int qi = AQ(i); const long long ri = AR(i); const int li = AL(i); const int si = AS(i);
            if (prereadloc < (long long)qi + li) {
                if (opcount > 100000 && ((double)opcount / (double)prereadloc) > 1000.0) { need_fast = true; break; }   // :27380 -> *_fast
                for (int k = testspace_en; k < i; ++k) {
                    int loc = vmx_smallorequal(S, S[k], k, SA) + 1;
                    vmx_sarg_insert_l(SA, loc, k, lane);
                }
                testspace_en = i;
                prereadloc = (long long)qi + li;
            }
            const double dli = (double)li;
            double max_scores = dli; int pre_index = VMX_NOPRE;
            for (int base = testspace_en - 1; base >= 0; base -= 64) {
                const int x = base - lane;
                const bool valid = x >= 0;
                int j = 0; double Sj = 0.0; double test = -1e300;
                if (valid) {
                    j = SA[x]; Sj = S[j];
                    const int qj = AQ(j), lj = AL(j), sj = AS(j); const long long rj = AR(j);
                    long long readgap = (long long)qi - qj - lj, refgap, bonus;
                    bool skip = false;
                    if (readgap < 0) {
                        bonus = (long long)qi + li - qj - lj;
                        if (bonus <= 0) skip = true;
                        readgap = 0;
                        long long overlap = (long long)qj + lj - qi;
                        if (si == sj) { if (si == 1) refgap = ri + overlap - (rj + lj); else refgap = rj - (ri + bonus); }
                        else { if (sj == -1) refgap = ri + overlap - rj + 1; else refgap = ri + bonus - 1 - (rj + lj); }
                    } else {
                        bonus = li;
                        if (si == sj) { if (si == 1) refgap = ri - rj - lj; else refgap = rj - ri - li; }
                        else { if (sj == -1) refgap = ri - rj + 1; else refgap = ri + li - 1 - rj - lj; }
                    }
                    if (!skip) {
                        long long gapcost = readgap - refgap; if (gapcost < 0) gapcost = -gapcost;
                        if (si == sj && refgap >= 0 && readgap <= maxgap && gapcost <= maxdiff) {
                            test = Sj + (double)bonus - s_gapcost[gapcost] - (double)rgc[readgap];
                        } else if (!mm) {
                            if (gapcost > extra_size) gapcost = extra_size;
                            double pen;
                            if (si != sj) pen = (skipcost < 50.0 ? skipcost : 50.0) + (double)tab.extra[gapcost];
                            else pen = skipcost + (double)tab.extra[gapcost];
                            test = Sj + (double)bonus - pen;
                        } else {
                            double pen = skipcost + tab.log2cache[gapcost < l2c_size ? gapcost : l2c_size];
                            test = Sj + (double)bonus - pen;
                        }
                    }
                }
                const double m_before = vmx_wave_excl_max_f64(test, max_scores);
                const bool brk = valid && (Sj < (m_before - dli));       // strict; opcount is bumped BEFORE this test (:27410-27415)
                const unsigned long long bmask = __ballot(brk);
                const unsigned long long vmask = __ballot(valid);
                const int first = bmask ? (__ffsll((unsigned long long)bmask) - 1) : 64;
                opcount += bmask ? (first + 1) : __popcll(vmask);
                double best = (lane < first && valid) ? test : -1e300; int bl = lane;
                for (int off = 32; off > 0; off >>= 1) {
                    double ob = __shfl_xor(best, off); int ol = __shfl_xor(bl, off);
                    if (ob > best || (ob == best && ol < bl)) { best = ob; bl = ol; }
                }
                const int jb = __shfl(j, bl);
                if (best > max_scores) { max_scores = best; pre_index = jb; }
                if (first < 64) break;
            }
            if (lane == 0) { S[i] = max_scores; P[i] = pre_index; }
            if (max_scores > g_max_scores) { g_max_scores = max_scores; g_max_index = i; }
            __syncthreads();
        }
        // traceback with overlap trimming :27508-27526 (serial, lane 0)
        if (lane == 0) {
            if (need_fast) { out_len[rd] = 0; out_score[rd] = 0; status[rd] = VM_READ_FASTPATH_DEV; }
            else {
                vmx_anchor* O = out_chain + a0;
                int w = 0; int take = g_max_index;
                vmx_anchor pre; pre.q = AQ(take); pre.r = AR(take); pre.l = (int16_t)AL(take); pre.s = (int16_t)AS(take);
                O[w++] = pre;
                while (P[take] != VMX_NOPRE) {
                    take = P[take];
                    vmx_anchor now; now.q = AQ(take); now.r = AR(take); now.l = (int16_t)AL(take); now.s = (int16_t)AS(take);
                    if (pre.q < now.q + now.l) {
                        int ov = now.q + now.l - pre.q;
                        vmx_anchor t = pre; t.q = pre.q + ov; t.l = (int16_t)(pre.l - ov); if (pre.s == 1) t.r = pre.r + ov;
                        O[w - 1] = t;
                    }
                    O[w++] = now;
                    pre = now;
                }
                out_len[rd] = w; out_score[rd] = g_max_scores; status[rd] = 0;
            }
            out_variant[rd] = mm ? 1 : 0;
        }
        __syncthreads();
#undef AQ
#undef AR
#undef AL
#undef AS
    }
}
