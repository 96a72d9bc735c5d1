// k_local.hip — local 9-mer re-seeding around the guide chains. The local chain DP is in k_chain_local.hip.
//
//   k_orient        reverse-complements the reads whose global chain is on the minus strand (get_readmap_DP_test,
//                   /root/reference/src/vacmap/mammap_clrnano.py:24060-24065)
//   k_local_prep    L1 merge_chain / drop_somechains / sort by 1/len (:28482-28574): one thread per read (vmx_local.h)
//   k_local_seed    L2 get_localmap_multi_all_forDP_inv_guide_1 (:23069-23345): one workgroup per read.
//                   Python's hash(str) tables are exact 9-mer identity, so the table is an 18-bit direct-address structure:
//                   HEAD[4^9] + NEXT[] linked lists over the window positions (built with one atomic exchange per position;
//                   head entries carry the slot's epoch, so the table is never torn down; an occupancy bitmap of it in LDS
//                   answers the look-ups of empty lists; a position's reference coordinate is implied by the window
//                   interval list). Per read position: forward + reverse-complement
//                   lookups, proximity filter against the two closest guide anchors (findClosest_1 :17560; the guide lives
//                   in LDS), accepted hits written in the reference's stream order (read pos asc, forward before reverse,
//                   ref pos asc). The sequential "flush once a run reaches 20" merge (:23232-23344) only couples hits of
//                   one diagonal that are <= k read positions apart, so after a stable radix sort by diagonal every such
//                   RUN is walked by its own lane; the reference's append order and its stable argsort by q+l (:28585)
//                   are restored by two key sorts on the emission key.
// Deviation D1: a 9-mer holding a non-ACGT base never matches.
#define VMX_SORT_LOGR 3          // eight keys per thread in the block sorts: this kernel is held to 80 VGPRs (three 512-thread workgroups per CU)
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"
#include "vmx_local_dev.h"

#define vmx_block_sort_u64(g, N, lds) vmx_block_sort_u64_impl((g), (N), (lds), VMX_SORT_LDS)
#define VMX_GUIDE_LDS 1024      // guide anchors kept in LDS (q 4 B + r 8 B); longer guides are read from HBM

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
__global__ void __launch_bounds__(64) k_local_prep(const vmx_anchor* __restrict__ path_rows, const int32_t* __restrict__ path_len, const int32_t* __restrict__ n_paths,
                             const int64_t* __restrict__ aoff, const double* __restrict__ gscore, int n_reads, int mode,
                             vmx_anchor* __restrict__ guide_rows, int32_t* __restrict__ guide_len, int32_t* __restrict__ n_guides_used,
                             int32_t* __restrict__ n_guides_total, int32_t* __restrict__ ws_pool) {
    // one wavefront per read: lane 0 takes the decisions (a handful of chains), all lanes copy the guide rows (a HiFi guide holds ~3000 anchors:
    // copied by the deciding thread alone they were most of the kernel's 1.5 ms)
    __shared__ int s_n;
    for (int r = (int)blockIdx.x; r < n_reads; r += (int)gridDim.x) {
        const int64_t a0 = aoff[r];
        int32_t* ws = ws_pool + VMX_PREP_WS * a0;                    // (a read has at most as many paths as anchors)
        const int np = n_paths[r];
        if (threadIdx.x == 0) {
            int32_t segs = 0;
            n_guides_used[r] = 0; n_guides_total[r] = 0;
            if (!(gscore[r] == 0.0 || np <= 0))
                vmx_local_prep(path_rows + a0, path_len + a0, np, mode, guide_rows + a0, guide_len + a0, &n_guides_used[r], &n_guides_total[r], ws, &segs);
            s_n = segs;
        }
        __syncthreads();
        const int nseg = s_n;
        int w = 0;
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const int src = ws[7 * np + sgi], ln = ws[8 * np + sgi];
            for (int t = (int)threadIdx.x; t < ln; t += (int)blockDim.x) guide_rows[a0 + w + t] = path_rows[a0 + src + t];
            w += ln;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ L2 seeding
// ascending insertion sort of a short run of int64 values in HBM (accepted hits of one read position and strand)
__device__ __forceinline__ void vmx_isort_i64(int64_t* a, int n) {
    for (int i = 1; i < n; ++i) { int64_t v = a[i]; int j = i - 1; while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; } a[j + 1] = v; }
}

__global__ void __launch_bounds__(512, VMX_LSEED_WAVES) k_local_seed(vmx_lseed_args A) {
    __shared__ uint64_t s_sort[VMX_SORT_LDS];
    __shared__ int s_gq[VMX_GUIDE_LDS];
    __shared__ long long s_gr[VMX_GUIDE_LDS];
    __shared__ int s_scan[20];
    __shared__ long long s_iv[64][2];     // disjoint intervals of k-mer starts [lo, hi) in global ref coordinates
    __shared__ int s_ivbase[65];          // cumulative number of k-mer starts before interval v
    __shared__ int s_niv;
    __shared__ int s_flag;
    __shared__ long long s_tot;
    __shared__ int s_emit;
    __shared__ unsigned long long s_min, s_max, s_min2, s_max2;
    __shared__ int s_wmax[16];
    __shared__ int s_next;
    const int nw = (int)(blockDim.x >> 6);
    const int k = A.k;
    const int nkey = 1 << (2 * k);
    // head entries are (epoch << 23 | window index): an entry of another epoch is an empty list, so the table is never reset between
    // uses (that was one random HBM write per window position); the slot's epoch lives in A.epoch_pool across launches, 511 = never used
    // k = 9 (the only local k-mer size the reference uses, vacmap:255): the 4^9 heads are folded into 2^14 BUCKETS (the k-mer's low 14
    // bits); a list entry carries the k-mer's remaining 4 bits next to its 23-bit link, and a walk skips entries of another k-mer. The
    // slot's table shrinks from 1 MB to 64 KB — 768 resident slots fit the L2 / Infinity Cache instead of spraying an 800 MB region with
    // random atomics — while the exact per-k-mer occupancy bitmap in LDS still answers nine in ten look-ups without touching it.
    const bool bucketed = 2 * k > 14 && ((size_t)1 << (2 * k)) <= (size_t)VMX_SORT_LDS * 64;
    const int nhead = bucketed ? (1 << 14) : nkey;
    int32_t* HEAD = A.head_pool + (size_t)blockIdx.x * (size_t)A.head_stride;
    unsigned ep = (unsigned)A.epoch_pool[blockIdx.x];
    auto head_idx = [&](int h) { return (((unsigned)h) >> 23) == ep ? (int)(((unsigned)h) & 0x7fffffu) : -1; };
#define VMX_HEAD_IDX(h) head_idx(h)
#define VMX_HB(km) (bucketed ? ((km) & 0x3fffu) : (km))                                     /* head slot of a k-mer */
#define VMX_ENT_OK(e, km) (!bucketed || ((unsigned)(e) >> 23) == ((km) >> 14))              /* list entry belongs to this k-mer */
#define VMX_ENT_NEXT(e) (bucketed ? (((e) & 0x7fffff) == 0x7fffff ? -1 : ((e) & 0x7fffff)) : (e))
    int32_t* NEXT = A.next_pool + (size_t)blockIdx.x * (size_t)A.tpos_cap;
    uint64_t* HKEY = A.hkey_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    uint64_t* HKEY2 = A.hkey2_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int64_t* HVAL = A.hval_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;     // by stream index: read position << 24 | window index << 1 | (strand == +1)
#define VMX_HV(q, t, f) ((int64_t)(((uint64_t)(q) << 24) | ((uint64_t)(t) << 1) | (uint64_t)(f)))          /* window index < 2^23 (checked with the windows): no limit on the reference's size */
    int32_t* SQ = A.sq_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;         // by sorted index: read position
    int32_t* DST = A.dst_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;       // by sorted index: start of the diagonal group
    int32_t* GOFF = A.goff_pool + (size_t)blockIdx.x * (size_t)A.hit_cap;
    int32_t* PCNT = A.pcnt_pool + (size_t)blockIdx.x * (size_t)A.pcnt_cap;
    int32_t* PC2 = A.pc2_pool + (size_t)blockIdx.x * (size_t)A.pcnt_cap;             // per read position: accepted forward | reverse << 16
    int64_t* STG = A.stg_pool + (size_t)blockIdx.x * 2 * (size_t)A.pcnt_cap;         // per read position: window index of the first accepted forward / reverse hit
    uint64_t* GKEY = A.gkey_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    int32_t* GQg = A.gq_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    int64_t* GRg = A.gr_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    long long vt0 = VMX_CLOCK();
#define VMX_T(ph) do { if (A.dbg && threadIdx.x == 0) { long long t1_ = VMX_CLOCK(); atomicAdd(&A.dbg[ph], (unsigned long long)(t1_ - vt0)); vt0 = t1_; } } while (0)
    while (true) {
        // reads are taken longest first from a device-side queue; every thread sees the same queue index, so the exit is uniform
        if (threadIdx.x == 0) s_next = atomicAdd(A.queue, 1);
        __syncthreads();
        const int qi = vmx_uniform_i32(s_next);
        __syncthreads();
        if (qi >= A.n_reads) break;
        const int r = A.order[qi];
        const int ng = A.n_guides_used[r];
        const int64_t a0 = A.aoff[r];
        const uint8_t* RD = A.ocodes + (A.rd_off ? A.rd_off[r] : A.roff[r]);
        const int L = A.rd_off ? (int)A.rd_len[r] : (int)(A.roff[r + 1] - A.roff[r]);
        vmx_anchor* OUT = A.la_rows + A.la_off[r];
        uint64_t* OKEY = A.la_ekey + A.la_off[r];
        vmx_anchor* SORTED = A.la_sorted + A.la_off[r];
        const int out_cap = A.rd_off ? (int)(A.la_off[r + 1] - A.la_off[r]) : (int)(A.la_slot_len * VMX_LA_SLOT((int64_t)L));     // slot of this read in the local-anchor pools
        int n_out = 0;
        int status = 0;
        int gbase = 0;
        // emission keys (the reference's append order) of guide g live in [ebase, ebase + 2 H_g): flushes inside runs by the stream index of the hit that
        // triggers them, then the leftovers of the last runs (+ H_g). A running base instead of the guide's number in fixed bits: any number of guides
        // (round 4 stopped at 511); 37 bits, i.e. 6.9e10 hits over all guides of a read, before the anchor's own 27-bit index
        uint64_t ebase = 0;
        for (int g = 0; g < ng; ++g) {
            // NOTE: no barrier-skipping break/continue below: a failed guide sets `status` and every later phase of this guide runs on
            // empty ranges, so all waves of the workgroup meet every __syncthreads() (see the ED kernel's note on hardware hangs)
            const vmx_anchor* G = A.guide_rows + a0 + gbase;     // descending read order
            const int m = A.guide_len[a0 + g];
            gbase += m;
            int N = 1; while (N < m) N <<= 1;
            if (N > A.gkey_cap) status = VM_READ_CAPACITY_DEV;
            const int mm = status ? 0 : m;
            // --- :23095-23102 readgap
            int rg = 0;
            for (int i = 1 + (int)threadIdx.x; i < mm; i += (int)blockDim.x) { int d = G[i].q - G[i - 1].q; if (d < 0) d = -d; rg = d > rg ? d : rg; }
            rg = vmx_wave_max_i32(rg);
            if (threadIdx.x == 0) s_flag = 0;
            __syncthreads();
            if (vmx_lane() == 0) atomicMax(&s_flag, rg);
            __syncthreads();
            long long readgap = (long long)s_flag + 1000; if (readgap < 5000) readgap = 5000;
            __syncthreads();
            // --- guide sorted by ref position (stable, :23103): key = r << 24 | original index
            const int NN = status ? 0 : N;
            for (int i = (int)threadIdx.x; i < NN; i += (int)blockDim.x) GKEY[i] = i < m ? (((uint64_t)G[i].r << 24) | (uint64_t)i) : ~0ULL;
            __syncthreads();
            int gk_lds = 0;                                       // does the sort leave the sorted guide keys in its LDS tile, and in which order
            if (NN > 1) gk_lds = vmx_block_sort_u64_tiled(GKEY, NN, s_sort, VMX_SORT_LDS);
            // guide in ascending read order (:23183) = reverse of the stored descending order; read positions are distinct
            const bool g_lds = mm <= VMX_GUIDE_LDS;
            for (int i = (int)threadIdx.x; i < mm; i += (int)blockDim.x) {
                const vmx_anchor a = G[mm - 1 - i];
                if (g_lds) { s_gq[i] = a.q; s_gr[i] = a.r; } else { GQg[i] = a.q; GRg[i] = a.r; }
            }
            __syncthreads();
            const int* GQ = g_lds ? s_gq : GQg;
            const long long* GR = g_lds ? s_gr : (const long long*)GRg;
            // --- windows (serial, thread 0): :23105-23180
            if (threadIdx.x == 0) {
                // the block sort leaves the sorted keys in its LDS staging buffer when they fit: walk them there, not in HBM
                auto GK = [&](int i) -> uint64_t { return gk_lds == 2 ? s_sort[vmx_sw(i)] : (gk_lds == 1 ? s_sort[i] : GKEY[i]); };
                int niv = 0; bool overflow = false;
                for (int attempt = 0; attempt < 2 && mm > 0; ++attempt) {
                    const bool split = attempt == 1;
                    niv = 0; bool retry = false;
                    long long ws = (long long)(GK(0) >> 24), we = ws;
                    int cur = vmx_pos2contig(A.coff, A.nseq, ws);
                    for (int i = 1; i <= mm; ++i) {
                        bool close_it = true; long long rr = 0;
                        if (i < mm) {
                            rr = (long long)(GK(i) >> 24);
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
                            if (i < mm) { ws = we = rr; if (split) cur = vmx_pos2contig(A.coff, A.nseq, rr); }
                        }
                    }
                    if (!retry) break;
                }
                long long tot = 0;
                for (int v = 0; v < niv; ++v) { s_ivbase[v] = (int)tot; tot += s_iv[v][1] - s_iv[v][0]; }
                s_ivbase[niv] = (int)(tot < 0x7fffffff ? tot : 0x7fffffff);
                if (tot > A.tpos_cap || tot >= (1LL << 23) - 1) overflow = true;     // 0x7fffff is the end-of-list mark of bucketed entries
                s_niv = niv; s_flag = overflow ? 1 : 0;
            }
            __syncthreads();
            if (s_flag) status = VM_READ_CAPACITY_DEV;
            const int niv = status ? 0 : s_niv;
            // occupancy bitmap of the head table (one bit per k-mer, 32 KB for k = 9) in the sort buffer, which is idle until the radix sort:
            // the windows fill about a tenth of the 4^k heads, so nine in ten look-ups of passes A and B are answered from LDS
            ep = ep + 1;
            if (ep >= 511u) { for (int i = (int)threadIdx.x; i < nhead; i += (int)blockDim.x) HEAD[i] = -1; ep = 0; __syncthreads(); }   // epochs used up: one real reset
            unsigned* BM = (unsigned*)s_sort;
            const bool use_bm = ((size_t)1 << (2 * k)) <= (size_t)VMX_SORT_LDS * 64;
            if (use_bm) { for (int i = (int)threadIdx.x; i < (1 << (2 * k)) / 32; i += (int)blockDim.x) BM[i] = 0u; __syncthreads(); }
            VMX_T(0);
            // the reference position of window index t is implied by the interval list (LDS): no position array in HBM
            auto tpos_of = [&](int t) -> long long { int v = 0; while (v + 1 < niv && t >= s_ivbase[v + 1]) ++v; return s_iv[v][0] + (long long)(t - s_ivbase[v]); };
            // --- read window :23183-23191
            int readstart = 0, readend = 0;
            if (mm > 0) {
                readstart = GQ[0] - A.read_span; if (readstart < 0) readstart = 0;
                readend = GQ[mm - 1] + A.read_span; if (readend > L - k + 1) readend = L - k + 1;
                if (A.r_st) { readstart = A.r_st[r]; readend = A.r_en[r] - k; }          // :22580, :22589
            }
            int npos = (status == 0 && readend > readstart) ? readend - readstart : 0;
            if (npos > A.pcnt_cap) { status = VM_READ_CAPACITY_DEV; npos = 0; }
            // --- table: one atomic exchange per window position links it in front of its 9-mer's list
            if (use_bm) {
                // only the window positions whose 9-mer the read can ask for are linked (about one in nine: the read window holds ~30 k of the
                // 4^9 k-mers): the bitmap first holds the READ's k-mers (forward and reverse complement), a coalesced sweep lists the matching
                // window positions, then the bitmap is rebuilt as the table's occupancy map while the marked positions are linked.
                // Nine in ten of the random atomic exchanges on the 1 MB head table in HBM disappear; a list a look-up can reach is complete.
                // (both sweeps roll the k-mer over runs of 8 consecutive positions per thread: 2 byte loads per position instead of k)
                const uint32_t KMASK = (1u << (2 * k)) - 1u;
                for (int p0 = 8 * (int)threadIdx.x; p0 < npos; p0 += 8 * (int)blockDim.x) {
                    uint32_t fw = 0; int nval = 0;
                    for (int i = 0; i < k - 1; ++i) { const uint8_t c = RD[readstart + p0 + i]; nval = c > 3 ? 0 : nval + 1; fw = (fw << 2) | (uint32_t)(c & 3); }
                    for (int j = 0; j < 8 && p0 + j < npos; ++j) {
                        const uint8_t c = RD[readstart + p0 + j + k - 1]; nval = c > 3 ? 0 : nval + 1; fw = ((fw << 2) | (uint32_t)(c & 3)) & KMASK;
                        if (nval >= k) { const uint32_t rv = vmx_kmer_rc(fw, k); if (fw != rv) { atomicOr(&BM[fw >> 5], 1u << (fw & 31)); atomicOr(&BM[rv >> 5], 1u << (rv & 31)); } }
                    }
                }
                if (threadIdx.x == 0) s_next = 0;
                __syncthreads();
                // the window positions whose k-mer the read holds are appended to a compact list (the hit pool SQ, idle until the sort):
                // ~3 k indices instead of a mark per window position written to and read back from HBM
                for (int v = 0; v < niv; ++v) {
                    const long long lo = s_iv[v][0], hi = s_iv[v][1]; const int base = s_ivbase[v];
                    for (long long x0 = lo + 8 * (long long)threadIdx.x; x0 < hi; x0 += 8 * (long long)blockDim.x) {
                        uint32_t km = 0; int nval = 0;
                        for (int i = 0; i < k - 1; ++i) { const uint8_t c = A.ref[x0 + i]; nval = c > 3 ? 0 : nval + 1; km = (km << 2) | (uint32_t)(c & 3); }
                        for (int j = 0; j < 8 && x0 + j < hi; ++j) {
                            const uint8_t c = A.ref[x0 + j + k - 1]; nval = c > 3 ? 0 : nval + 1; km = ((km << 2) | (uint32_t)(c & 3)) & KMASK;
                            if (nval >= k && ((BM[km >> 5] >> (km & 31)) & 1u)) { const int p = atomicAdd(&s_next, 1); if (p < A.hit_cap) SQ[p] = base + (int)(x0 + j - lo); }
                        }
                    }
                }
                __syncthreads();
                int nmark = s_next;
                __syncthreads();
                if (nmark > A.hit_cap) { status = VM_READ_CAPACITY_DEV; nmark = 0; npos = 0; }      // re-run with larger pools (vmx_stage_local.hip)
                for (int i = (int)threadIdx.x; i < (1 << (2 * k)) / 32; i += (int)blockDim.x) BM[i] = 0u;
                __syncthreads();
                for (int e = (int)threadIdx.x; e < nmark; e += (int)blockDim.x) {
                    const int idx = SQ[e];
                    bool ok; const uint32_t km = vmx_kmer_at(A.ref, tpos_of(idx), k, ok);
                    const int old = VMX_HEAD_IDX(atomicExch(&HEAD[VMX_HB(km)], (int)((ep << 23) | (unsigned)idx)));
                    atomicOr(&BM[km >> 5], 1u << (km & 31));
                    NEXT[idx] = bucketed ? (int)(((km >> 14) << 23) | (unsigned)(old < 0 ? 0x7fffff : old)) : old;
                }
            } else
            for (int v = 0; v < niv; ++v) {
                const long long lo = s_iv[v][0], hi = s_iv[v][1]; const int base = s_ivbase[v];
                for (long long x = lo + threadIdx.x; x < hi; x += 2 * blockDim.x) {          // two positions per round: both exchanges in flight together
                    const long long x2 = x + blockDim.x;
                    bool ok, ok2 = false; const uint32_t km = vmx_kmer_at(A.ref, x, k, ok);
                    uint32_t km2 = 0; if (x2 < hi) km2 = vmx_kmer_at(A.ref, x2, k, ok2);
                    const int idx = base + (int)(x - lo), idx2 = base + (int)(x2 - lo);
                    int old = -1, old2 = -1;
                    if (ok) old = VMX_HEAD_IDX(atomicExch(&HEAD[km], (int)((ep << 23) | (unsigned)idx)));
                    if (ok2) old2 = VMX_HEAD_IDX(atomicExch(&HEAD[km2], (int)((ep << 23) | (unsigned)idx2)));
                    if (ok) NEXT[idx] = old;
                    if (ok2) NEXT[idx2] = old2;
                }
            }
            __syncthreads();
            VMX_T(1);
            // pass A: accepted hits per read position and strand (any order), no barrier inside the loop so that the waves overlap their
            // table walks. The first accepted hit of either strand is parked in STG: a position with at most one hit per strand (nearly all)
            // needs no second walk in pass B.
            // (instantiated for the LDS copy of the guide and for the HBM one: a pointer chosen at run time would turn the bisection's
            // dependent loads into flat accesses)
            auto pass_a = [&](const int* GQ, const long long* GR) __attribute__((always_inline)) {
            for (int pi = (int)threadIdx.x; pi < npos; pi += (int)blockDim.x) {
                const int iloc = readstart + pi;
                int cf = 0, cr = 0; long long ff = 0, fr = 0;
                bool ok; const uint32_t fw = vmx_kmer_at(RD, iloc, k, ok);
                const uint32_t rv = vmx_kmer_rc(fw, k);
                const bool pf = ok && fw != rv && (!use_bm || ((BM[fw >> 5] >> (fw & 31)) & 1u));
                const bool pr = ok && fw != rv && iloc > 0 && (!use_bm || ((BM[rv >> 5] >> (rv & 31)) & 1u));
                if (pf || pr) {                                  // (the guide bisection only for the one position in five that can hit at all)
                    const int hf = pf ? VMX_HEAD_IDX(HEAD[VMX_HB(fw)]) : -1, hr = pr ? VMX_HEAD_IDX(HEAD[VMX_HB(rv)]) : -1;   // both list heads in flight together
                    int b0, b1, c0, c1; vmx_find_closest(GQ, mm, iloc, b0, b1, c0, c1);
                    long long interval = (long long)b0 + b1 + 500; if (interval > 2000) interval = 2000;
                    const long long ref1 = GR[c0], ref2 = GR[c1];
                    long long rgap = (long long)iloc - GQ[c0]; if (rgap < 0) rgap = -rgap;
                    for (int t = hf; t >= 0;) { const int e = NEXT[t]; if (VMX_ENT_OK(e, fw)) { const long long rl = tpos_of(t); if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) { if (cf == 0) ff = t; ++cf; } } t = VMX_ENT_NEXT(e); }
                    for (int t = hr; t >= 0;) { const int e = NEXT[t]; if (VMX_ENT_OK(e, rv)) { const long long rl = tpos_of(t); if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) { if (cr == 0) fr = t; ++cr; } } t = VMX_ENT_NEXT(e); }
                }
                PCNT[pi] = cf + cr;
                PC2[pi] = (cf > 0xffff ? 0xffff : cf) | ((cr > 0x7fff ? 0x7fff : cr) << 16);
                if (cf + cr) { STG[2 * pi] = ff; STG[2 * pi + 1] = fr; }      // (read back only where there are hits)
            }
            };
            if (g_lds) pass_a(s_gq, s_gr); else pass_a(GQg, (const long long*)GRg);
            __syncthreads();
            // exclusive offsets: every thread scans a contiguous slice of the per-position counts
            long long H;
            {
                const int chunk = (npos + (int)blockDim.x - 1) / (int)blockDim.x;
                const int lo = (int)threadIdx.x * chunk < npos ? (int)threadIdx.x * chunk : npos;
                const int hi = lo + chunk < npos ? lo + chunk : npos;
                int sum = 0;
                for (int i = lo; i < hi; ++i) sum += PCNT[i];
                int tot; int runx = vmx_block_excl_scan(sum, s_scan, &tot);
                for (int i = lo; i < hi; ++i) { const int cc = PCNT[i]; PCNT[i] = runx; runx += cc; }
                H = tot;
                __syncthreads();
            }
            VMX_T(2);
            if (H > A.hit_cap) { status = VM_READ_CAPACITY_DEV; H = 0; npos = 0; }
            // pass B: hits in stream order (read pos asc; forward before reverse; ref pos asc).
            // key = (point + 2^36) << 26 | stream index, point = r - q (forward) or -(r + q) (reverse): ONE key space, like pointdict (Q1)
            for (int pi = (int)threadIdx.x; pi < npos; pi += (int)blockDim.x) {
                const int iloc = readstart + pi;
                const int c2 = PC2[pi];
                const int cf = c2 & 0xffff, cr = c2 >> 16;
                if (c2 == 0) continue;
                long long w = PCNT[pi];
                if (cf <= 1 && cr <= 1) {
                    if (cf) { const int t = (int)STG[2 * pi]; const long long rl = tpos_of(t); HKEY[w] = ((uint64_t)((rl - iloc) + (1LL << 36)) << 26) | (uint64_t)w; HVAL[w] = VMX_HV(iloc, t, 1); ++w; }
                    if (cr) { const int t = (int)STG[2 * pi + 1]; const long long rl = tpos_of(t); HKEY[w] = ((uint64_t)(-(rl + iloc) + (1LL << 36)) << 26) | (uint64_t)w; HVAL[w] = VMX_HV(iloc, t, 0); }
                    continue;
                }
                bool ok; const uint32_t fw = vmx_kmer_at(RD, iloc, k, ok);
                const uint32_t rv = vmx_kmer_rc(fw, k);
                int b0, b1, c0, c1; vmx_find_closest(GQ, mm, iloc, b0, b1, c0, c1);
                long long interval = (long long)b0 + b1 + 500; if (interval > 2000) interval = 2000;
                const long long ref1 = GR[c0], ref2 = GR[c1];
                long long rgap = (long long)iloc - GQ[c0]; if (rgap < 0) rgap = -rgap;
                const long long wf = w;
                const bool pf = !use_bm || ((BM[fw >> 5] >> (fw & 31)) & 1u), pr = iloc > 0 && (!use_bm || ((BM[rv >> 5] >> (rv & 31)) & 1u));
                for (int t = pf ? VMX_HEAD_IDX(HEAD[VMX_HB(fw)]) : -1; t >= 0;) { const int e = NEXT[t]; if (VMX_ENT_OK(e, fw)) { const long long rl = tpos_of(t); if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) HVAL[w++] = t; } t = VMX_ENT_NEXT(e); }
                const long long wr = w;
                if (pr) for (int t = VMX_HEAD_IDX(HEAD[VMX_HB(rv)]); t >= 0;) { const int e = NEXT[t]; if (VMX_ENT_OK(e, rv)) { const long long rl = tpos_of(t); if (vmx_local_accept(rl, ref1, ref2, interval, rgap)) HVAL[w++] = t; } t = VMX_ENT_NEXT(e); }
                if (wr - wf > 1) vmx_isort_i64(HVAL + wf, (int)(wr - wf));
                if (w - wr > 1) vmx_isort_i64(HVAL + wr, (int)(w - wr));
                for (long long x = wf; x < w; ++x) {
                    const int t = (int)HVAL[x]; const long long rl = tpos_of(t); const bool fwd = x < wr;      // (window indices ascend with the reference position)
                    const long long point = fwd ? rl - iloc : -(rl + iloc);
                    HKEY[x] = ((uint64_t)(point + (1LL << 36)) << 26) | (uint64_t)x; HVAL[x] = VMX_HV(iloc, t, fwd ? 1 : 0);
                }
            }
            __syncthreads();
            // the table is no longer needed; its entries die with the epoch
            // group the hits by diagonal, keeping stream order inside a diagonal: stable LSD radix sort. Only the grouping matters (the order of
            // the groups is restored by the emission keys), so the points are first mapped injectively to a dense range: forward points
            // (r - q > 0) and reverse points (-(r + q) < 0) form two clusters ~2r apart, each only as wide as the window
            {
                const unsigned long long MID = 1ULL << 36;
                unsigned long long mnL = ~0ULL, mxL = 0ULL, mnH = ~0ULL, mxH = 0ULL;
                for (long long i = threadIdx.x; i < H; i += blockDim.x) {
                    const unsigned long long pk = HKEY[i] >> 26;
                    if (pk < MID) { mnL = pk < mnL ? pk : mnL; mxL = pk > mxL ? pk : mxL; } else { mnH = pk < mnH ? pk : mnH; mxH = pk > mxH ? pk : mxH; }
                }
                for (int o = 32; o > 0; o >>= 1) {
                    unsigned long long a1 = __shfl_xor(mnL, o), b1 = __shfl_xor(mxL, o), a2 = __shfl_xor(mnH, o), b2 = __shfl_xor(mxH, o);
                    mnL = a1 < mnL ? a1 : mnL; mxL = b1 > mxL ? b1 : mxL; mnH = a2 < mnH ? a2 : mnH; mxH = b2 > mxH ? b2 : mxH;
                }
                if (threadIdx.x == 0) { s_min = ~0ULL; s_max = 0ULL; s_min2 = ~0ULL; s_max2 = 0ULL; }
                __syncthreads();
                if (vmx_lane() == 0) {
                    atomicMin((unsigned long long*)&s_min, mnL); atomicMax((unsigned long long*)&s_max, mxL);
                    atomicMin((unsigned long long*)&s_min2, mnH); atomicMax((unsigned long long*)&s_max2, mxH);
                }
                __syncthreads();
                const bool hasL = s_min != ~0ULL, hasH = s_min2 != ~0ULL;
                const unsigned long long baseL = hasL ? s_min : 0ULL, spanL = hasL ? s_max - s_min + 1 : 0ULL;
                const unsigned long long baseH = hasH ? s_min2 : 0ULL, spanH = hasH ? s_max2 - s_min2 + 1 : 0ULL;
                for (long long i = threadIdx.x; i < H; i += blockDim.x) {
                    const unsigned long long kk = HKEY[i], pk = kk >> 26;
                    const unsigned long long f = pk < MID ? pk - baseL : spanL + (pk - baseH);
                    HKEY[i] = (f << 26) | (kk & ((1ULL << 26) - 1));
                }
                __syncthreads();
                int nbits = 0; { unsigned long long range = H > 0 ? spanL + spanH : 0; while (nbits < 40 && (range >> nbits)) ++nbits; }
                // the stream index sits in the low 26 bits, so sorting whole keys = the stable sort by diagonal: up to VMX_SORT_LDS hits (most
                // reads) that is one bitonic sort in LDS instead of the radix passes through HBM
                int NP2 = 1; while (NP2 < H) NP2 <<= 1;
                if (H > 1 && NP2 <= A.hit_cap) {
                    // (beyond one LDS tile: tile-wise bitonic sort, only the few long-distance steps through HBM — vmx_block_sort_u64_tiled)
                    for (long long i = H + threadIdx.x; i < NP2; i += blockDim.x) HKEY[i] = ~0ULL;
                    __syncthreads();
                    vmx_block_sort_u64_tiled(HKEY, NP2, s_sort, VMX_SORT_LDS);
                } else {
                    uint64_t* res = vmx_block_radix_sort_u64(HKEY, HKEY2, (int)H, 26, (uint64_t)0, nbits, (int*)s_sort, s_scan);
                    if (res != HKEY) { for (long long i = threadIdx.x; i < H; i += blockDim.x) HKEY[i] = HKEY2[i]; }
                }
                __syncthreads();
            }
            int64_t* SV = (int64_t*)HKEY2;           // the sort's second buffer is free again: hit values (refloc << 1 | fwd) in sorted order
            // sorted-order copies of the read positions + start index of every diagonal group (block-wide running max)
            {
                int carry = -1;
                for (long long i0 = 0; i0 < H; i0 += blockDim.x) {
                    const long long i = i0 + threadIdx.x;
                    int v = -1;
                    if (i < H) {
                        const uint64_t sidx = HKEY[i] & ((1ULL << 26) - 1);
                        const int64_t hv = HVAL[sidx];              // one gather per hit: read position and hit value travel together
                        SQ[i] = (int)((uint64_t)hv >> 24);
                        SV[i] = (tpos_of((int)((hv >> 1) & 0x7fffff)) << 1) | (hv & 1);
                        if (i == 0 || (HKEY[i] >> 26) != (HKEY[i - 1] >> 26)) v = (int)i;
                    }
                    for (int o = 1; o < 64; o <<= 1) { int x = __shfl_up(v, o); if (vmx_lane() >= o) v = x > v ? x : v; }
                    if (vmx_lane() == 63) s_wmax[threadIdx.x >> 6] = v;
                    __syncthreads();
                    int base = carry;
                    for (int w2 = 0; w2 < (int)(threadIdx.x >> 6); ++w2) base = s_wmax[w2] > base ? s_wmax[w2] : base;
                    v = v > base ? v : base;
                    if (i < H) DST[i] = v;
                    int tot = carry; for (int w2 = 0; w2 < nw; ++w2) tot = s_wmax[w2] > tot ? s_wmax[w2] : tot;
                    carry = tot;
                    __syncthreads();
                }
            }
            VMX_T(3);
            // --- run-merge (:23232-23344). A RUN = hits of one diagonal whose read positions are <= k apart. Every thread owns a contiguous
            // slice of the sorted hits and walks the runs that START in it (to their end, which may lie in a later slice), so the work is
            // balanced by hit count. pass 0 counts the anchors a thread emits (-> offsets), pass 1 writes them with their emission key:
            //   flush inside a run: key = stream index of the triggering hit; leftover of a run: stream index of the next hit on the same
            //   diagonal (it flushes the leftover, :23248) or, for the last run of a diagonal, 2^26 + stream index of the diagonal's first hit
            //   (appended after all flushes in first-appearance order, :23343).
            {
                // one walk: output slots are claimed with an LDS counter as anchors are emitted (their order in OUT is free, see above), so
                // the runs are not walked a second time to learn the offsets
                if (threadIdx.x == 0) s_emit = 0;
                __syncthreads();
                {
                    const long long HH = status ? 0 : H;
                    // hits are dealt to the threads round-robin (neighbouring lanes test neighbouring hits: coalesced loads); a thread walks the runs
                    // that START at its hits. The order in which anchors are emitted is free: the emission keys restore the reference's order.
                    for (long long j0 = threadIdx.x; j0 < HH; j0 += blockDim.x) {
                        long long j = j0;
                        const uint64_t kj = HKEY[j]; const uint64_t pk = kj >> 26;
                        const int qj = SQ[j];
                        const bool starts = (j == 0) || (HKEY[j - 1] >> 26) != pk || (qj - SQ[j - 1] > k);
                        if (!starts) continue;
                        const long long i = j;
                        const long long hv0 = SV[j];
                        long long cq = qj, cr = hv0 >> 1, cl = k; int cs = (hv0 & 1) ? 1 : -1;
                        int prevq = qj;
                        for (++j; j < HH; ++j) {
                            const uint64_t k2 = HKEY[j]; const int q2 = SQ[j];
                            if ((k2 >> 26) != pk || q2 - prevq > k) break;
                            const uint64_t sidx = k2 & ((1ULL << 26) - 1);
                            const long long hv = SV[j];
                            const long long refloc = hv >> 1; const int strand = (hv & 1) ? 1 : -1;
                            const long long bouns = (long long)q2 - (cq + cl) + k;          // > 0 inside a run
                            if (cl + bouns < 20) { if (strand == 1) { cs = 1; cl += bouns; } else { cr = refloc; cs = -1; cl += bouns; } }
                            else {
                                const int o = n_out + atomicAdd(&s_emit, 1);
                                if (o < out_cap) { OUT[o] = vmx_mk_anchor(cq, cr, cs, cl); OKEY[o] = ebase + sidx; }
                                const long long nq = cq + cl;
                                if (strand == 1) { cr = cr + cl; cs = 1; } else { cr = refloc; cs = -1; }
                                cq = nq; cl = bouns;
                            }
                            prevq = q2;
                        }
                        {
                            uint64_t ek;
                            if (j < HH && (HKEY[j] >> 26) == pk) ek = ebase + (HKEY[j] & ((1ULL << 26) - 1));
                            else ek = ebase + (uint64_t)HH + (HKEY[DST[i]] & ((1ULL << 26) - 1));
                            const int o = n_out + atomicAdd(&s_emit, 1);
                            if (o < out_cap) { OUT[o] = vmx_mk_anchor(cq, cr, cs, cl); OKEY[o] = ek; }
                        }
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) s_tot = s_emit;
                    __syncthreads();
                    if (n_out + s_tot > out_cap) status = VM_READ_CAPACITY_DEV;
                    ebase += 2 * (uint64_t)HH;
                    if (ebase >> 37) status = VM_READ_CAPACITY_DEV;
                    __syncthreads();
                }
            }
            if (!status) n_out += (int)s_tot;
            __syncthreads();
            VMX_T(4);
        }
        // --- restore the reference's append order (emission key), then the stable argsort by q+l (:28585)
        {
            long long NO = 1; while (NO < n_out) NO <<= 1;
            if (status == 0 && n_out > 0 && NO > A.hit_cap) status = VM_READ_CAPACITY_DEV;
            const long long no = status ? 0 : n_out;
            const long long NP = no > 0 ? NO : 0;
            for (long long i = threadIdx.x; i < NP; i += blockDim.x) HKEY[i] = i < no ? ((OKEY[i] << 27) | (uint64_t)i) : ~0ULL;      // emission key (< 2^37: running base of the guide + final flag * H + stream index), then the anchor's index (< 2^27)
            __syncthreads();
            if (NP > 1) vmx_block_sort_u64_tiled(HKEY, (int)NP, s_sort, VMX_SORT_LDS);
            __syncthreads();
            for (long long e = threadIdx.x; e < no; e += blockDim.x) GOFF[e] = (int)(HKEY[e] & 0x7ffffffu);   // rank e -> anchor index
            __syncthreads();
            for (long long e = threadIdx.x; e < NP; e += blockDim.x) {
                uint64_t kk = ~0ULL;
                if (e < no) { const vmx_anchor a = OUT[GOFF[e]]; kk = ((uint64_t)(uint32_t)(A.sort_by_start ? a.q : a.q + a.l) << 32) | (uint64_t)e; }   // :28585; mode R sorts by read start
                HKEY[e] = kk;
            }
            __syncthreads();
            if (NP > 1) vmx_block_sort_u64_tiled(HKEY, (int)NP, s_sort, VMX_SORT_LDS);
            __syncthreads();
            for (long long x = threadIdx.x; x < no; x += blockDim.x) SORTED[x] = OUT[GOFF[(int)(HKEY[x] & 0xffffffffu)]];
            __syncthreads();
        }
        VMX_T(5);
        if (threadIdx.x == 0) { A.la_cnt[r] = status ? 0 : n_out; A.status[r] = status; }
        __syncthreads();
    }
    if (threadIdx.x == 0) A.epoch_pool[blockIdx.x] = (int)ep;
#undef VMX_HEAD_IDX
#undef VMX_HV
#undef VMX_HB
#undef VMX_ENT_OK
#undef VMX_ENT_NEXT
}
