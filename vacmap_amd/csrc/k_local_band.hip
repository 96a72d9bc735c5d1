// k_local_band.hip — local 9-mer re-seeding around the guide chains, guide-banded and LDS-tiled.
//
//   k_local_seed_band   get_localmap_multi_all_forDP_inv_guide_1 (/root/reference/src/vacmap/mammap_clrnano.py:23069-23345), one workgroup
//                       per read, same inputs / outputs / scratch pools as k_local_seed (k_local.hip), which stays as the general form: a
//                       read this kernel cannot take (VM_READ_BANDFALL_DEV) is re-run there by vmx_local_stage.
//
// Why a second form. The reference joins the 9-mers of the read window with those of the reference windows (guide +- 7000) and THEN drops
// every hit that is not near the two guide anchors closest to the read position (:23216-23231: |refloc - ref1| within `interval` <= 2000,
// the same around ref2, or |readgap - refgap| < 500). k_local_seed does literally that through HBM: a hash table of the windows per read,
// hit pools, three sorts over all hits of a guide — 145x the algorithmic bytes. Here the read window is cut into CHUNKS of
// LB_QC positions; for a chunk the filter itself bounds the reference positions that can be accepted to a few intervals around the guide
// anchors bracketing the chunk (the BAND, clipped to the reference windows: a position outside every window is not in the reference's
// table). The chunk's forward and reverse-complement 9-mers go into an LDS hash table (bucket heads + entries whose index IS the read
// position and strand), the band is streamed through it (rolling 9-mers, every in-band position is examined, so a repeated k-mer yields
// all its positions like local_lookuptable_m), candidates are tested with the exact filter of :23231, and the accepted hits
// (strand, diagonal, read position) are sorted in LDS.
//   The run merge (:23232-23344) couples only hits of one diagonal that are <= k read positions apart: inside a chunk a lane walks each run.
// Across chunks three things survive, all of them small:
//   * a run whose last hit lies in the chunk's last k positions may go on in the next chunk: its state waits in an LDS list ("open runs");
//   * the leftover anchor of a diagonal's last run is appended by the reference either when the NEXT hit on that diagonal arrives, however
//     much later (:23248), or at the very end in first-appearance order of the diagonals (:23343): chunks append (diagonal, chunk, HEAD =
//     emission key of the group's first hit | TAIL = the pending anchor) records to a log in HBM — two per (diagonal, chunk) group, ~0.25
//     per read base instead of ~25 B x 10 per hit — which is sorted once per guide and resolved pairwise;
//   * the order in which the reference appended the anchors ("emission key": the (read position, strand, window index) of the hit that
//     triggered the append, or FINAL | key of the diagonal's first hit) — restored by one sort per guide, then the stable argsort by
//     q + l (:28585) as before.
// Execution model: ONE WAVEFRONT per read (a 64-thread workgroup), as many as the LDS lets a CU hold. A chunk offers a few hundred to a
// couple of thousand independent items at a time (positions to hash, band positions to probe, hits to sort, runs to walk) and the chunks of a
// read depend on each other through the open runs, so a 512-thread workgroup per read spent its time in barriers and behind one-thread
// sections (first form of this kernel: 6 ms per batch, 39 % plan + join, 22 % walk, 19 % guide + windows). A wave alone needs no barrier at
// all (LDS operations of one wave execute in order), a serial step stalls only its own read, and the latency of the band's loads is hidden
// by the other reads resident on the CU.
// HBM traffic per read: the read and ~(1 + 4000 / LB_QC) x the window bases (L2 hits mostly), the log, the anchors and their two sorts.
// Deviation D1: a 9-mer holding a non-ACGT base never matches.
#define VMX_SORT_LOGR 3
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"
#include "vmx_local_dev.h"

#define LB_NBLOG VMX_LB_NBLOG
#define LB_BMMASK ((1u << VMX_LB_BMLOG) - 1u)
#define LB_NB (1 << LB_NBLOG)       // bucket heads of the chunk table
#define LB_OPEN 32                  // runs that may cross one chunk boundary
#define LB_PIECES 16                // disjoint band intervals of a chunk (after clipping to the windows)
#define LB_WIN 32                   // reference windows of a guide
#define LB_HULL_ONE 16384           // the <= 4 accept intervals of one guide segment are taken as their hull up to this many positions
#define LB_SLACK 64                 // band intervals closer than this are streamed as one
#define LB_QB 26                    // bits of a hit key that hold the read position (relative to the read window)
#define LB_SEQB 13                  // bits of a log key that hold the chunk number

static_assert(VMX_LB_GS <= 128 && VMX_LB_HCAP <= VMX_LB_SORTK && 2 * VMX_LB_QC < (1 << 13), "k_local_seed_band: LDS layout");
__device__ __forceinline__ int vmx_bits_u64(unsigned long long v) { int b = 0; while (b < 64 && (v >> b)) ++b; return b; }

// k-mers of the 8 consecutive positions x .. x + 7 of a 1-byte-per-base code array (any alignment; the arrays are padded by 64 bytes):
// km[j] = the k-mer starting at x + j, bit j of the returned mask = it holds no ambiguous base. Two 8-byte loads (three for k > 9).
__device__ __forceinline__ unsigned vmx_kmers8_w(uint64_t w0, uint64_t w1, uint64_t w2, int k, uint32_t KMASK, uint32_t km[8]) {
    uint32_t v = 0; int nval = 0; unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        if (i >= 7 + k) break;
        const uint32_t c = (uint32_t)((i < 8 ? w0 >> (8 * i) : (i < 16 ? w1 >> (8 * (i - 8)) : w2 >> (8 * (i - 16)))) & 0xffu);
        nval = c > 3u ? 0 : nval + 1;
        v = ((v << 2) | (c & 3u)) & KMASK;
        const int j = i - (k - 1);
        if (j >= 0 && j < 8) { km[j] = v; if (nval >= k) ok |= 1u << j; }
    }
    return ok;
}
// The same for k <= 9 in a dozen instructions per position less: the 16 bases are packed to 2 bits each first (base i at bits 2 i .. 2 i + 1), and
// the key of position j is bits 2 j .. 2 j + 2 k - 1 of that word — the k-mer with its FIRST base lowest. Keys never leave the kernel (bucket,
// occupancy bit, equality), so any one-to-one coding serves as long as the read and the band use the same one; the reverse complement is
// the same bit operation in either coding (complement, reverse the order of the 2-bit groups).
__device__ __forceinline__ uint32_t lb_pack4(uint32_t w) { uint32_t x = w & 0x03030303u; x = (x | (x >> 6)) & 0x000F000Fu; return (x | (x >> 12)) & 0xFFu; }
__device__ __forceinline__ uint32_t lb_bad4(uint32_t w) { uint32_t y = (w >> 2) & 0x01010101u; y = (y | (y >> 7)) & 0x00030003u; return (y | (y >> 14)) & 0xFu; }
__device__ __forceinline__ unsigned vmx_kmers8_le(uint64_t w0, uint64_t w1, int k, uint32_t KMASK, uint32_t km[8]) {
    const uint32_t a = (uint32_t)w0, b = (uint32_t)(w0 >> 32), c = (uint32_t)w1, d = (uint32_t)(w1 >> 32);
    const uint32_t P = lb_pack4(a) | (lb_pack4(b) << 8) | (lb_pack4(c) << 16) | (lb_pack4(d) << 24);
    uint32_t bad = 0;
    if ((a | b | c | d) & 0x04040404u) bad = lb_bad4(a) | (lb_bad4(b) << 4) | (lb_bad4(c) << 8) | (lb_bad4(d) << 12);     // codes above 3: ambiguous bases (rare)
    const uint32_t kb = (1u << k) - 1u;
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { km[j] = (P >> (2 * j)) & KMASK; if (((bad >> j) & kb) == 0u) ok |= 1u << j; }
    return ok;
}
__device__ __forceinline__ unsigned vmx_kmers8(const uint8_t* base, int k, uint32_t KMASK, uint32_t km[8]) {
    uint64_t w0, w1, w2 = 0;
    __builtin_memcpy(&w0, base, 8); __builtin_memcpy(&w1, base + 8, 8);
    if (k <= 9) return vmx_kmers8_le(w0, w1, k, KMASK, km);
    __builtin_memcpy(&w2, base + 16, 8);
    return vmx_kmers8_w(w0, w1, w2, k, KMASK, km);
}

// reverse complement of a k-mer in 2-bit codes: complement, reverse the bits, swap the two bits of every base, drop the low 32 - 2 k bits
__device__ __forceinline__ uint32_t vmx_kmer_rc_fast(uint32_t fw, int k) {
#ifdef VMX_EMU
    uint32_t x = ~fw, r = 0; for (int i = 0; i < 32; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
#else
    uint32_t r = __brev(~fw);
#endif
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return r >> (32 - 2 * k);
}
// the block sorts of vmx_device.h unroll their compare-exchange networks: called from five places they made 82 KB of code (the
// instruction cache shared by two CUs holds 64 KB, and nine wavefronts per CU sit in different phases). One copy each.
// (they take the LDS buffers from the kernel's dynamic shared array themselves: a pointer PASSED to a function that is not inlined is a generic
// pointer, and every access through it a flat_load / flat_store — measured 3.7x on the hit sort)
__device__ __attribute__((noinline)) int lb_sort_hbm(uint64_t* g, int N) {
    VMX_DYN_SHARED(uint64_t, s_u);
    return vmx_block_sort_u64_tiled(g, N, s_u, VMX_LB_SORTK);
}
__device__ __forceinline__ bool lb_sort_hits(int n) {      // s_hit[0 .. n) -> sort region, sorted; true: swizzled order
    VMX_DYN_SHARED(uint64_t, s_u);
    uint64_t* const lds = s_u; const uint64_t* const src = s_u + VMX_LB_REGION_U64;
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    int NH = 2; while (NH < n) NH <<= 1;
    const bool sw = vmx_bitonic_fast_ok(NH);
    for (int i = tid; i < NH; i += T) lds[sw ? vmx_sw(i) : i] = i < n ? src[i] : ~0ULL;
    __syncthreads();
    if (sw) vmx_bitonic_tile_sw(lds, NH, 0, NH); else vmx_block_bitonic_passes(lds, NH);
    return sw;
}

#ifndef VMX_LB_PRIO
#define VMX_LB_PRIO 2
#endif
__global__ void __launch_bounds__(64, VMX_LB_WAVES) k_local_seed_band(vmx_lseed_args A) {
    // Round 6: this kernel's waves hold the LDS-limited slots of the CUs (eight 17.8 KB wavefronts each) for 13.5 of a 16 ms HiFi step while the gap fill of the other
    // batches shares their SIMDs, and a lone wave per read is bound by instruction issue and LDS latency: at the default priority it ran 1.7x slower under load than
    // alone. Its instructions now go first (s_setprio 2, below the one-wave chain kernels' 3); the fill, which is throughput-bound, takes the issue slots left over.
    VMX_SETPRIO(VMX_LB_PRIO);
    VMX_DYN_SHARED(uint64_t, s_u);                 // VMX_LB_LDS_BYTES
    uint64_t* const s_sort = s_u;                  // [0, 8 VMX_LB_SORTK): sorts / the sorted hits of a chunk
    uint32_t* const s_head = (uint32_t*)s_u;       // LB_NB bucket heads (entry index + 1, 0 = empty) ...
    uint32_t* const s_ent = s_head + LB_NB;        // ... the entries: 2 i + s of chunk position i, strand s: (k-mer >> LB_NBLOG) << 13 | next ...
    uint32_t* const s_bm = s_ent + 2 * VMX_LB_QC;  // ... and a 2^14-bit occupancy map of the chunk's k-mers (their low 14 bits): together <= the sort region
    uint64_t* const s_hit = s_u + VMX_LB_REGION_U64;    // band intervals of the plan, then the chunk's hits
    uint32_t* const s_cq = (uint32_t*)(s_hit + VMX_LB_HCAP);   // candidates of one sweep over the band: (lane * 8 + j) << 22 | k-mer
    __shared__ int s_gq[VMX_LB_GS];
    __shared__ long long s_gr[VMX_LB_GS];
    __shared__ unsigned char s_seg[VMX_LB_QC];               // closest staged guide anchor of every chunk position << 1 | (the next one too)
    __shared__ long long s_iv[LB_WIN][2];
    __shared__ int s_ivbase[LB_WIN + 1];
    __shared__ long long s_pc[LB_PIECES][2];
    __shared__ unsigned long long s_opd[2][LB_OPEN];          // open runs: (strand | diagonal) ...
    __shared__ int s_opq[2][LB_OPEN], s_opp[2][LB_OPEN];      // ... start of the anchor in progress, read position of the run's last hit
    __shared__ int s_nop[2];
    __shared__ int s_niv, s_flag, s_next, s_fail, s_cnt, s_nit, s_npc, s_nhit, s_nlog, s_emit;
    __shared__ unsigned long long s_hlo, s_hhi;
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int k = A.k;
    const uint32_t KMASK = (1u << (2 * k)) - 1u;
    const int64_t hit_cap = A.hit_cap;
    uint64_t* HKEY = A.hkey_pool + (size_t)blockIdx.x * (size_t)hit_cap;        // log keys, then the emission / final sort keys
    uint64_t* LOGV0 = A.hkey2_pool + (size_t)blockIdx.x * (size_t)hit_cap;      // log values: HEAD emission key | TAIL q << 32 | l
    int64_t* LOGV1 = A.hval_pool + (size_t)blockIdx.x * (size_t)hit_cap;        //             TAIL reference position
    int32_t* GOFF = A.goff_pool + (size_t)blockIdx.x * (size_t)hit_cap;
    uint64_t* GKEY = A.gkey_pool + (size_t)blockIdx.x * (size_t)A.gkey_cap;
    long long vt0 = VMX_CLOCK();
#define VMX_T(ph) do { if (A.dbg && threadIdx.x == 0) { long long t1_ = VMX_CLOCK(); atomicAdd(&A.dbg[ph], (unsigned long long)(t1_ - vt0)); vt0 = t1_; } } while (0)
    while (true) {
        if (tid == 0) s_next = atomicAdd(A.queue, 1);
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
        const int out_cap = A.rd_off ? (int)(A.la_off[r + 1] - A.la_off[r]) : (int)(A.la_slot_len * VMX_LA_SLOT((int64_t)L));
        int n_out = 0;
        int status = 0, why = 0;                  // why: what made the kernel hand the read back (VMX_LSEED_TRACE), 0 = nothing
        int gbase = 0;
        for (int g = 0; g < ng; ++g) {
            // (as in k_local_seed: a failed guide sets `status`, every later phase runs on empty ranges and all waves meet every barrier)
            const vmx_anchor* G = A.guide_rows + a0 + gbase;     // descending read order
            const int m = A.guide_len[a0 + g];
            gbase += m;
            int N = 1; while (N < m) N <<= 1;
            if (N > A.gkey_cap) { status = VM_READ_BANDFALL_DEV; why = 1; }
            const int mm = status ? 0 : m;
            // --- :23095-23102 readgap
            int rg = 0;
            for (int i = 1 + tid; i < mm; i += T) { int d = G[i].q - G[i - 1].q; if (d < 0) d = -d; rg = d > rg ? d : rg; }
            rg = vmx_wave_max_i32(rg);
            if (tid == 0) { s_flag = 0; s_fail = 0; }
            __syncthreads();
            if (vmx_lane() == 0) atomicMax(&s_flag, rg);
            __syncthreads();
            long long readgap = (long long)s_flag + 1000; if (readgap < 5000) readgap = 5000;
            __syncthreads();
            // --- guide sorted by ref position (stable, :23103): key = r << 24 | original index
            const int NN = status ? 0 : N;
            for (int i = tid; i < NN; i += T) GKEY[i] = i < m ? (((uint64_t)G[i].r << 24) | (uint64_t)i) : ~0ULL;
            __syncthreads();
            int gk_lds = 0;
            if (NN > 1) gk_lds = lb_sort_hbm(GKEY, NN);
            // --- windows :23105-23180 — disjoint intervals of k-mer starts [lo, hi) in global reference coordinates. A window is a maximal stretch
            // of the sorted guide whose neighbours lie closer than `readgap` (on the retry: and on one contig); every lane tests its own
            // anchors, a running maximum carries the window's first index to its last anchor, whose lane turns (first, last) into an interval;
            // the few raw intervals are merged by one lane. (The lane-0 walk over a HiFi guide of ~3000 anchors was 16 % of the kernel.)
            {
                auto GK = [&](int i) -> uint64_t { return gk_lds == 2 ? s_sort[vmx_sw(i)] : (gk_lds == 1 ? s_sort[i] : GKEY[i]); };
                long long* const RW = (long long*)s_hit;                   // raw windows (a, b), ascending; at most VMX_LB_HCAP / 2
                bool retry = false, overflow = false;
                int nraw = 0;
                for (int attempt = 0; attempt < 2 && mm > 0; ++attempt) {
                    const bool split = attempt == 1;
                    nraw = 0; retry = false;
                    int carry = 0;                                          // first index of the window the previous block ended in
                    for (int i0 = 0; i0 < mm; i0 += T) {
                        const int i = i0 + tid;
                        const bool in = i < mm;
                        long long rr = 0, rp = 0, rn = 0;
                        if (in) { rr = (long long)(GK(i) >> 24); rp = i > 0 ? (long long)(GK(i - 1) >> 24) : rr; rn = i + 1 < mm ? (long long)(GK(i + 1) >> 24) : rr; }
                        int cr_ = 0;
                        bool first = in && (i == 0 || !((rr - rp) < readgap)), lastw = in && (i + 1 == mm || !((rn - rr) < readgap));
                        if (split && in) {
                            cr_ = vmx_pos2contig(A.coff, A.nseq, rr);
                            if (i > 0 && cr_ != vmx_pos2contig(A.coff, A.nseq, rp)) first = true;
                            if (i + 1 < mm && cr_ != vmx_pos2contig(A.coff, A.nseq, rn)) lastw = true;
                        }
                        int v = first ? i : -1;
                        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o); if (vmx_lane() >= o) v = x > v ? x : v; }
                        if (v < carry) v = carry;
                        carry = __shfl(v, 63);
                        // the window's last anchor makes the interval
                        bool emit = false; long long ia = 0, ib = 0; bool bad = false;
                        if (lastw) {
                            const long long ws = (long long)(GK(v) >> 24), we = rr;
                            if (ws != we) {                                // single-point windows are dropped (:23110, :23113)
                                const int c = vmx_pos2contig(A.coff, A.nseq, ws);
                                if (c != vmx_pos2contig(A.coff, A.nseq, we)) bad = true;                 // retry_diffcontig (:23142)
                                else {
                                    const long long cst = A.coff[c], clen = A.coff[c + 1] - cst;
                                    const long long lf = ws - cst < A.look_span ? ws - cst : A.look_span;
                                    long long lo = ws - lf - cst, hi = we + A.look_span - cst; if (hi > clen) hi = clen;
                                    const long long nk = (hi - lo) - k + 1;
                                    if (nk > 0) { emit = true; ia = cst + lo; ib = ia + nk; }
                                }
                            }
                        }
                        if (__ballot(bad)) {
                            // the reference stops at the first window that spans two contigs: everything before it would be thrown away with the retry
                            retry = true;
                        }
                        const unsigned long long bal = __ballot(emit);
                        if (emit) { const int o = nraw + __popcll(bal & ((1ULL << vmx_lane()) - 1ULL)); if (o < VMX_LB_HCAP / 2) { RW[2 * o] = ia; RW[2 * o + 1] = ib; } }
                        nraw += __popcll(bal);
                    }
                    if (nraw > VMX_LB_HCAP / 2) { overflow = true; nraw = 0; }
                    if (!retry || split) break;
                }
                __syncthreads();
                if (tid == 0) {
                    int niv = 0;
                    for (int w = 0; w < nraw; ++w) {
                        const long long a = RW[2 * w], b = RW[2 * w + 1];
                        if (niv > 0 && a <= s_iv[niv - 1][1]) { if (b > s_iv[niv - 1][1]) s_iv[niv - 1][1] = b; }     // (touching intervals are one: a run may cross)
                        else if (niv < LB_WIN) { s_iv[niv][0] = a; s_iv[niv][1] = b; ++niv; }
                        else overflow = true;
                    }
                    long long tot = 0;
                    for (int v = 0; v < niv; ++v) { s_ivbase[v] = (int)tot; tot += s_iv[v][1] - s_iv[v][0]; }
                    s_ivbase[niv] = (int)(tot < 0x7fffffff ? tot : 0x7fffffff);
                    if (tot >= 0x7fffffff) overflow = true;
                    s_niv = niv; s_flag = overflow ? 1 : 0;
                }
            }
            __syncthreads();
            if (s_flag) { status = VM_READ_BANDFALL_DEV; why = 2; }
            const int niv = status ? 0 : s_niv;
            // the window index of a reference position inside the windows (ascends with the position)
            auto t_of = [&](long long x) -> long long { int v = 0; while (v + 1 < niv && x >= s_iv[v + 1][0]) ++v; return (long long)s_ivbase[v] + (x - s_iv[v][0]); };
            // --- read window :23183-23191 (GQ ascending = the stored order reversed)
            int readstart = 0, readend = 0;
            if (mm > 0) {
                readstart = G[mm - 1].q - A.read_span; if (readstart < 0) readstart = 0;
                readend = G[0].q + A.read_span; if (readend > L - k + 1) readend = L - k + 1;
                if (A.r_st) { readstart = A.r_st[r]; readend = A.r_en[r] - k; }          // :22580, :22589
            }
            int npos = (status == 0 && niv > 0 && readend > readstart) ? readend - readstart : 0;
            const long long wlo = niv ? s_iv[0][0] : 0, whi = niv ? s_iv[niv - 1][1] : 0;
            // key layouts of this guide. hit: (strand | diagonal) << 26 | read position; log: ((strand | diagonal) << 13 | chunk) << 1 | TAIL, then the
            // record index; emission: read position, strand, window index, FINAL on top
            const int dbits = vmx_bits_u64((unsigned long long)((whi - wlo) + npos));
            const int tb = vmx_bits_u64((unsigned long long)s_ivbase[niv > 0 ? niv : 0]), pb = vmx_bits_u64((unsigned long long)npos);
            const int ilb = 64 - (1 + dbits + LB_SEQB + 1);
            const int ekb = pb + tb + 1;                                    // bit of the FINAL flag
            if (npos > 0 && (npos >= (1 << LB_QB) || 1 + dbits + LB_QB > 63 || ilb < 14 || ekb + 1 > 50)) { status = VM_READ_BANDFALL_DEV; why = 3; npos = 0; }
            const unsigned long long DM = dbits >= 64 ? ~0ULL : ((1ULL << dbits) - 1ULL);
            const unsigned long long QM = (1ULL << LB_QB) - 1ULL;
            auto ekey_of = [&](int q, int sb, long long x) -> uint64_t { return ((uint64_t)(unsigned)(q - readstart) << (tb + 1)) | ((uint64_t)sb << tb) | (uint64_t)t_of(x); };
            // lower-bound cursor of the guide: anchors with GQ < readstart
            if (tid == 0) { s_cnt = 0; s_nop[0] = 0; s_nop[1] = 0; s_nlog = 0; s_emit = 0; }
            __syncthreads();
            if (npos > 0) {
                int c = 0;
                for (int i = tid; i < mm; i += T) c += G[mm - 1 - i].q < readstart ? 1 : 0;
                c = vmx_wave_sum_i32(c);
                if (vmx_lane() == 0 && c) atomicAdd(&s_cnt, c);
            }
            __syncthreads();
            int cj = s_cnt;
            __syncthreads();
            VMX_T(0);
            int q0 = readstart, qc = VMX_LB_QC, seq = 0, cur = 0;
            const int qend = npos > 0 ? readend : readstart;
            while (q0 < qend) {
                // ---------------------------------------------------------------- plan: guide slice, closest anchors, band
                const int js = cj > 0 ? cj - 1 : 0;
                const int navail = mm - js < VMX_LB_GS ? mm - js : VMX_LB_GS;
                for (int i = tid; i < navail; i += T) { const vmx_anchor a = G[mm - 1 - (js + i)]; s_gq[i] = a.q; s_gr[i] = a.r; }
                if (tid == 0) { s_cnt = 0; s_nit = 0; s_hlo = ~0ULL; s_hhi = 0ULL; s_nop[cur ^ 1] = 0; }
                __syncthreads();
                int qb = q0 + qc < qend ? q0 + qc : qend;
                {
                    int c = 0;
                    for (int i = tid; i < navail; i += T) c += s_gq[i] <= qb - 1 ? 1 : 0;
                    c = vmx_wave_sum_i32(c);
                    if (vmx_lane() == 0 && c) atomicAdd(&s_cnt, c);
                }
                __syncthreads();
                int je = s_cnt;                                   // staged anchors at or before the chunk's last position
                if (je >= navail && js + navail < mm) { qb = s_gq[navail - 1]; je = navail - 1; }       // more anchors than the slice holds: the chunk ends at the last staged one
                const int ns = je + 1 < navail ? je + 1 : navail;
                const int cj_next = js + je;
                const int qlen = qb - q0;
                const bool g_first = js == 0, g_last = js + ns == mm;
                // closest guide anchors of every chunk position (findClosest_1 :17560 on the slice: same answer, the read positions are distinct)
                for (int i0 = 8 * tid; i0 < qlen; i0 += 8 * T) {                          // eight consecutive positions per lane: one bisection, then the slice is walked
                    int lo = 0, hi = ns;
                    { const int p = q0 + i0; while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_gq[mid] <= p) lo = mid + 1; else hi = mid; } }
                    int gcur = lo < ns ? s_gq[lo] : 0x7fffffff, gprev = lo > 0 ? s_gq[lo - 1] : -1;      // (kept in registers: the slice is read only when the walk advances)
                    for (int j = 0; j < 8 && i0 + j < qlen; ++j) {
                        const int p = q0 + i0 + j;
                        while (gcur <= p) { gprev = gcur; ++lo; gcur = lo < ns ? s_gq[lo] : 0x7fffffff; }      // first staged anchor beyond p
                        int c0, c1;
                        if (lo > 0 && gprev == p) c0 = c1 = lo - 1;
                        else if (lo == 0) c0 = c1 = 0;
                        else if (lo == ns) c0 = c1 = ns - 1;
                        else { c0 = lo - 1; c1 = lo; }
                        s_seg[i0 + j] = (unsigned char)((c0 << 1) | (c1 != c0 ? 1 : 0));
                    }
                }
                // the reference positions :23231 can accept for the chunk's positions, per guide anchor / segment between two anchors / head / tail
                {
                    auto push = [&](long long lo, long long hi) {
                        if (lo < 0) lo = 0;
                        if (hi < lo) return;
                        if (hi - lo >= (1LL << 28) - 1) { s_fail = 4; return; }
                        const int o = atomicAdd(&s_nit, 1);
                        if (o < VMX_LB_HCAP) s_hit[o] = ((uint64_t)lo << 28) | (uint64_t)(hi - lo + 1); else s_fail = 4;
                        atomicMin(&s_hlo, (unsigned long long)lo); atomicMax(&s_hhi, (unsigned long long)hi);
                    };
                    // positions whose closest anchors are (ref1, ref2), interval <= iv, |read gap to ref1| in [da, db]
                    auto item = [&](long long ref1, long long ref2, long long iv, long long da, long long db) {
                        const long long l1 = ref1 - iv, h1 = ref1 + iv, l2 = ref2 - iv, h2 = ref2 + iv;
                        const long long l3 = ref1 + da - 499, h3 = ref1 + db + 499, l4 = ref1 - db - 499, h4 = ref1 - da + 499;
                        long long lo = l1 < l2 ? l1 : l2; lo = lo < l4 ? lo : l4; lo = lo < l3 ? lo : l3;
                        long long hi = h1 > h2 ? h1 : h2; hi = hi > h3 ? hi : h3; hi = hi > h4 ? hi : h4;
                        if (hi - lo <= LB_HULL_ONE) push(lo, hi);
                        else { push(l1, h1); push(l2, h2); push(l3, h3); push(l4, h4); }
                    };
                    for (int x = tid; x < ns; x += T) {
                        const int gx = s_gq[x]; const long long rx = s_gr[x];
                        if (gx >= q0 && gx < qb) item(rx, rx, 500, 0, 0);
                        if (x >= 1) {
                            const int gp = s_gq[x - 1];
                            const int pa = gp + 1 > q0 ? gp + 1 : q0, pe = gx - 1 < qb - 1 ? gx - 1 : qb - 1;
                            if (pa <= pe) { long long iv = (long long)(gx - gp) + 500; if (iv > 2000) iv = 2000; item(s_gr[x - 1], rx, iv, pa - gp, pe - gp); }
                        }
                        if (x == 0 && g_first) {
                            const int pa = q0, pe = gx - 1 < qb - 1 ? gx - 1 : qb - 1;
                            if (pa <= pe) { long long iv = 2LL * (gx - pa) + 500; if (iv > 2000) iv = 2000; item(rx, rx, iv, gx - pe, gx - pa); }
                        }
                        if (x == ns - 1 && g_last) {
                            const int pa = gx + 1 > q0 ? gx + 1 : q0, pe = qb - 1;
                            if (pa <= pe) { long long iv = 2LL * (pe - gx) + 500; if (iv > 2000) iv = 2000; item(rx, rx, iv, pa - gx, pe - gx); }
                        }
                    }
                }
                __syncthreads();
                VMX_T(6);
                const int nit = s_nit < VMX_LB_HCAP ? s_nit : VMX_LB_HCAP;
                const bool one_piece = nit > 0 && (long long)(s_hhi - s_hlo) <= (long long)qlen + 12288;
                if (!one_piece && nit > 1) {
                    int NI = 1; while (NI < nit) NI <<= 1;
                    for (int i = nit + tid; i < NI; i += T) s_hit[i] = ~0ULL;
                    __syncthreads();
                    vmx_block_bitonic_passes(s_hit, NI);
                }
                if (tid == 0) {
                    int np = 0, wv = 0; bool over = false;
                    auto clip = [&](long long lo, long long hi) {            // [lo, hi) against the windows; pieces and windows both ascend
                        while (wv < niv && s_iv[wv][1] <= lo) ++wv;
                        for (int v = wv; v < niv && s_iv[v][0] < hi; ++v) {
                            const long long a = lo > s_iv[v][0] ? lo : s_iv[v][0], b = hi < s_iv[v][1] ? hi : s_iv[v][1];
                            if (a >= b) continue;
                            if (np > 0 && a <= s_pc[np - 1][1]) { if (b > s_pc[np - 1][1]) s_pc[np - 1][1] = b; }
                            else if (np < LB_PIECES) { s_pc[np][0] = a; s_pc[np][1] = b; ++np; }
                            else over = true;
                        }
                    };
                    if (one_piece) clip((long long)s_hlo, (long long)s_hhi + 1);
                    else if (nit > 0) {
                        long long clo = (long long)(s_hit[0] >> 28), chi = clo + (long long)(s_hit[0] & ((1ULL << 28) - 1));
                        for (int i = 1; i < nit; ++i) {
                            const long long lo = (long long)(s_hit[i] >> 28), hi = lo + (long long)(s_hit[i] & ((1ULL << 28) - 1));
                            if (lo <= chi + LB_SLACK) { if (hi > chi) chi = hi; }
                            else { clip(clo, chi); clo = lo; chi = hi; }
                        }
                        clip(clo, chi);
                    }
                    s_npc = np;
                    if (over) s_fail = 5;
                }
                // ---------------------------------------------------------------- the chunk's 9-mers (both strands) in the LDS table
                // (heads and entries lie below the interval list: disjoint)
                for (int i = tid; i < LB_NB; i += T) s_head[i] = 0u;
                for (int i = tid; i < (1 << (VMX_LB_BMLOG - 5)); i += T) s_bm[i] = 0u;
                __syncthreads();
                const int npc = s_fail ? 0 : s_npc;
                VMX_T(7);
                for (int i0 = 8 * tid; i0 < qlen; i0 += 8 * T) {                          // eight consecutive positions per lane: two 8-byte loads
                    uint32_t km[8]; const unsigned okm = vmx_kmers8(RD + q0 + i0, k, KMASK, km);
                    uint32_t rvs[8], of[8], orv[8]; unsigned use = 0, user = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        rvs[j] = vmx_kmer_rc_fast(km[j], k);
                        if (i0 + j < qlen && ((okm >> j) & 1u) && km[j] != rvs[j]) {         // :23213
                            use |= 1u << j;
                            if (q0 + i0 + j > 0) user |= 1u << j;                        // rc_testseq[-(iloc + k): -iloc] is '' at iloc == 0 (:23212)
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {                                        // all exchanges in flight together, the links afterwards
                        if ((use >> j) & 1u) { of[j] = atomicExch(&s_head[km[j] & (LB_NB - 1)], (uint32_t)(2 * (i0 + j) + 1)); atomicOr(&s_bm[(km[j] & LB_BMMASK) >> 5], 1u << (km[j] & 31u)); }
                        if ((user >> j) & 1u) { orv[j] = atomicExch(&s_head[rvs[j] & (LB_NB - 1)], (uint32_t)(2 * (i0 + j) + 2)); atomicOr(&s_bm[(rvs[j] & LB_BMMASK) >> 5], 1u << (rvs[j] & 31u)); }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if ((use >> j) & 1u) s_ent[2 * (i0 + j)] = ((km[j] >> LB_NBLOG) << 13) | of[j];
                        if ((user >> j) & 1u) s_ent[2 * (i0 + j) + 1] = ((rvs[j] >> LB_NBLOG) << 13) | orv[j];
                    }
                }
                __syncthreads();
                VMX_T(8);
                // ---------------------------------------------------------------- stream the band through it
                // the runs the previous chunk left open come back as one VIRTUAL hit each, at the read position of the run's last hit: the sort puts
                // it in front of the chunk's hits on its diagonal, and the run walk treats it like any other hit (a run it starts begins its
                // first anchor at the carried start)
                const int nprev = s_nop[cur];
                for (int e = tid; e < nprev; e += T) s_hit[e] = (s_opd[cur][e] << LB_QB) | (uint64_t)(unsigned)(s_opp[cur][e] - readstart);
                if (tid == 0) s_nhit = nprev;
                __syncthreads();
                if (A.dbg && tid == 0) { long long bl = 0; for (int pc = 0; pc < npc; ++pc) bl += s_pc[pc][1] - s_pc[pc][0]; atomicAdd(&A.dbg[9], (unsigned long long)bl); atomicAdd(&A.dbg[10], 1ULL); atomicAdd(&A.dbg[12], (unsigned long long)qlen); }
                // Sweep = 512 band positions, eight per lane: (1) every lane rolls its k-mers and tests the occupancy map — no divergence, the
                // passing (position, k-mer) pairs are packed into a queue with a ballot; (2) the queue is drained one candidate per lane: bucket
                // list, exact k-mer check, the filter of :23231. (As one loop per position the list walks of 64 lanes ran in lockstep behind the
                // longest of them: 60 us per chunk.)
                // Candidates of several sweeps share the queue when their positions fit next to the k-mer in 32 bits (relative to the band's first
                // position): after a sweep only whole groups of 64 are drained, the remainder (< 64) waits for the next sweep's — every lane busy
                // in the bucket walks. Otherwise (a band spread over > 2^(32 - 2k) positions, k > 9) the queue is emptied after every sweep.
                const int relbits = 32 - 2 * k;
                const long long xbase_all = npc ? s_pc[0][0] : 0;
                const bool accumulate = npc > 0 && k <= 9 && (s_pc[npc - 1][1] - xbase_all) < (1LL << relbits);
                int ncq = 0;
                auto drain = [&](int cnt, long long xb) {                                 // candidates [0, cnt) of the queue; x = xb + the entry's offset
                    for (int c = tid; c < cnt; c += T) {
                        const uint32_t cw = s_cq[c];
                        const uint32_t km = cw & KMASK;
                        const long long x = xb + (long long)(cw >> (2 * k));
                        const uint32_t chk = km >> LB_NBLOG;
                        for (uint32_t e = s_head[km & (LB_NB - 1)]; e != 0u;) {
                            const uint32_t ent = s_ent[e - 1];
                            if ((ent >> 13) == chk) {
                                const int i = (int)((e - 1) >> 1), sb = (int)((e - 1) & 1);
                                const int q = q0 + i;
                                const int sg = s_seg[i], c0 = sg >> 1, c1 = c0 + (sg & 1);
                                int b0 = s_gq[c0] - q; if (b0 < 0) b0 = -b0;
                                int b1 = s_gq[c1] - q; if (b1 < 0) b1 = -b1;
                                long long interval = (long long)b0 + b1 + 500; if (interval > 2000) interval = 2000;
                                if (vmx_local_accept(x, s_gr[c0], s_gr[c1], interval, (long long)b0)) {
                                    const unsigned long long drel = (unsigned long long)(sb ? (x - wlo) + (q - readstart) : (x - wlo) + (readend - 1 - q));
                                    const int o = atomicAdd(&s_nhit, 1);
                                    if (o < VMX_LB_HCAP) s_hit[o] = (((((uint64_t)sb << dbits) | drel) << LB_QB) | (uint64_t)(unsigned)(q - readstart));
                                }
                            }
                            e = ent & 0x1fffu;
                        }
                    }
                };
                for (int pc = 0; pc < npc; ++pc) {
                    const long long lo = s_pc[pc][0], hi = s_pc[pc][1];
                    uint64_t wn0 = 0, wn1 = 0;                                           // the next sweep's bases are on their way while this one is probed
                    if (lo + 8LL * tid < hi) { __builtin_memcpy(&wn0, A.ref + lo + 8LL * tid, 8); __builtin_memcpy(&wn1, A.ref + lo + 8LL * tid + 8, 8); }
                    for (long long xs = lo; xs < hi; xs += 8LL * T) {
                        const long long x0 = xs + 8LL * tid;
                        const long long xb = accumulate ? xbase_all : xs;
                        {
                            uint32_t kms[8]; unsigned okm = 0;
                            const uint64_t w0 = wn0, w1 = wn1; uint64_t w2 = 0;
                            if (x0 + 8LL * T < hi) { __builtin_memcpy(&wn0, A.ref + x0 + 8LL * T, 8); __builtin_memcpy(&wn1, A.ref + x0 + 8LL * T + 8, 8); }
                            if (x0 < hi) { if (k > 9) { __builtin_memcpy(&w2, A.ref + x0 + 16, 8); okm = vmx_kmers8_w(w0, w1, w2, k, KMASK, kms); } else okm = vmx_kmers8_le(w0, w1, k, KMASK, kms); }
                            uint32_t bw[8];
                            if (x0 >= hi) { for (int j = 0; j < 8; ++j) kms[j] = 0u; }
#pragma unroll
                            for (int j = 0; j < 8; ++j) bw[j] = s_bm[(kms[j] & LB_BMMASK) >> 5];          // all eight words in flight together
                            const uint32_t rel0 = (uint32_t)(x0 - xb);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const bool pass = x0 + j < hi && ((okm >> j) & 1u) && ((bw[j] >> (kms[j] & 31u)) & 1u);
                                const unsigned long long bal = __ballot(pass);
                                if (pass) s_cq[ncq + __popcll(bal & ((1ULL << vmx_lane()) - 1ULL))] = ((rel0 + (uint32_t)j) << (2 * k)) | kms[j];
                                ncq += __popcll(bal);
                            }
                        }
                        __syncthreads();
                        const int take = accumulate ? (ncq & ~63) : ncq;
                        drain(take, xb);
                        __syncthreads();
                        if (take < ncq) { uint32_t v = 0; if (tid < ncq - take) v = s_cq[take + tid]; __syncthreads(); if (tid < ncq - take) s_cq[tid] = v; }
                        ncq -= take;
                        __syncthreads();
                    }
                }
                drain(ncq, xbase_all);
                __syncthreads();
                const int nhit = s_nhit;
                if (A.dbg && tid == 0) atomicAdd(&A.dbg[11], (unsigned long long)nhit);
                if (s_fail) { status = VM_READ_BANDFALL_DEV; why = s_fail; break; }        // (uniform: s_fail is read after a barrier)
                if (nhit > VMX_LB_HCAP) {                                                 // more hits than the tile holds: the same positions in narrower chunks
                    if (qc <= 16) { status = VM_READ_BANDFALL_DEV; why = 6; break; }
                    qc >>= 1;
                    __syncthreads();
                    continue;
                }
                VMX_T(1);
                // ---------------------------------------------------------------- sort the hits by (strand, diagonal, read position)
                const bool sw = lb_sort_hits(nhit);
                VMX_T(2);
                // ---------------------------------------------------------------- the runs (:23232-23344), every hit on its own lane
                // A RUN = hits of one diagonal whose read positions are <= k apart. Along a run the reference's cache entry (q, r, s, l) only ever
                // does this: the anchor that starts at `st` takes every hit with q < st + 20 - k (:23241, l + bouns < 20), the first hit beyond
                // appends it with l = (its last hit's q) + k - st and the next anchor starts at (that last hit's q) + k. So the last hit b of an
                // anchor fixes the next anchor's last hit: NX(b) = the last hit of the run with q < q_b + 20 — a function of b alone. The anchors
                // of a run are the chain b1 = (last hit with q < q_first + 20 - k), b2 = NX(b1), ...: the chain is marked by pointer doubling
                // (J <- J o J, marks follow the current power), then every marked hit and every run start emits its anchor. Sequentially this
                // was one lane per run and one dependent LDS load per hit (47 % of the kernel on HiFi reads, whose runs fill the chunk).
                {
                    auto KH = [&](int i) -> uint64_t { return s_sort[sw ? vmx_sw(i) : i]; };
                    uint32_t* const QR = (uint32_t*)s_hit;                              // (index of the run's first hit) << 16 | (q - qbase): ascending
                    unsigned short* const NX = (unsigned short*)(QR + VMX_LB_HCAP);      // last hit of the anchor that follows hit i
                    unsigned short* const JP = NX + VMX_LB_HCAP;                         // its powers
                    uint32_t* const MK = s_cq;                                            // marks, one bit per hit
                    unsigned short* const TMP = (unsigned short*)(s_cq + 32);
                    const int qbase = q0 - 64;
                    auto emit = [&](long long cq, long long cr, int cs, long long cl, uint64_t ek) {
                        const int o = n_out + atomicAdd(&s_emit, 1);
                        if (o < out_cap) { OUT[o] = vmx_mk_anchor(cq, cr, cs, cl); OKEY[o] = ek; }
                    };
                    auto logrec = [&](unsigned long long dk, int sq, int tail, uint64_t v0, int64_t v1) {
                        const int o = atomicAdd(&s_nlog, 1);
                        if ((int64_t)o < hit_cap && ((unsigned long long)o >> ilb) == 0ULL) {
                            HKEY[o] = (((((uint64_t)dk << LB_SEQB) | (uint64_t)sq) << 1 | (uint64_t)tail) << ilb) | (uint64_t)o;
                            LOGV0[o] = v0; LOGV1[o] = v1;
                        } else s_fail = 8;
                    };
                    for (int i = tid; i < 32; i += T) MK[i] = 0u;
                    // (1) run starts, carried to every hit of the run by a running maximum
                    {
                        int carry = 0;
                        for (int i0 = 0; i0 < nhit; i0 += T) {
                            const int i = i0 + tid; const bool in = i < nhit;
                            int v = -1, qr = 0;
                            if (in) {
                                const uint64_t kj = KH(i); const int qj = readstart + (int)(kj & QM);
                                bool nr = i == 0;
                                if (i > 0) { const uint64_t kp = KH(i - 1); nr = (kp >> LB_QB) != (kj >> LB_QB) || qj - (readstart + (int)(kp & QM)) > k; }
                                v = nr ? i : -1; qr = qj - qbase;
                            }
                            for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(v, o); if (vmx_lane() >= o) v = x > v ? x : v; }
                            if (v < carry) v = carry;
                            carry = __shfl(v, 63);
                            if (in) QR[i] = ((uint32_t)v << 16) | (uint32_t)qr;
                        }
                    }
                    __syncthreads();
                    // last index j in [i, i + 19] with QR[j] < lim (QR ascends; QR[i] < lim)
                    auto last_below = [&](int i, uint32_t lim) -> int {
                        int lo = i, hi = i + 19 < nhit - 1 ? i + 19 : nhit - 1;
                        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (QR[mid] < lim) lo = mid; else hi = mid - 1; }
                        return lo;
                    };
                    auto carried_start = [&](unsigned long long dk) -> int {           // the open run a virtual hit stands for
                        for (int e = 0; e < nprev; ++e) if (s_opd[cur][e] == dk) return s_opq[cur][e];
                        return 0;
                    };
                    // first anchor of the run that starts at hit i: it starts at the hit (at the carried start for a virtual hit) and takes q < start + 20 - k
                    auto root_last = [&](int i, int qi, unsigned long long dk, int& st) -> int {
                        st = qi < q0 ? carried_start(dk) : qi;
                        return last_below(i, ((uint32_t)i << 16) | (uint32_t)(st - qbase + 20 - k));
                    };
                    // (2) NX, and the mark of every run's first chain element
                    for (int i = tid; i < nhit; i += T) {
                        const uint32_t w = QR[i];
                        const int nx = last_below(i, w + 20u);
                        NX[i] = (unsigned short)nx; JP[i] = (unsigned short)nx;
                        if ((int)(w >> 16) == i) {
                            const uint64_t kj = KH(i); int st;
                            const int b1 = root_last(i, readstart + (int)(kj & QM), kj >> LB_QB, st);
                            atomicOr(&MK[b1 >> 5], 1u << (b1 & 31));
                        }
                    }
                    __syncthreads();
                    // (3) pointer doubling: marks at distance < 2^r from the first element reach distance < 2^(r+1)
                    for (int round = 0; round < 12; ++round) {
                        bool ch = false;
                        for (int i = tid; i < nhit; i += T) {
                            const int j = JP[i];
                            if ((MK[i >> 5] >> (i & 31)) & 1u) atomicOr(&MK[j >> 5], 1u << (j & 31));
                            const int jj = JP[j];
                            TMP[i] = (unsigned short)jj; ch = ch || jj != j;
                        }
                        __syncthreads();
                        for (int i = tid; i < nhit; i += T) JP[i] = TMP[i];
                        __syncthreads();
                        if (!__any(ch)) break;
                    }
                    // (4) anchors: the one a run starts with, and the one behind every marked hit that is not the run's last
                    for (int i = tid; i < nhit; i += T) {
                        const uint64_t kj = KH(i); const unsigned long long dk = kj >> LB_QB;
                        const int qi = readstart + (int)(kj & QM);
                        const uint32_t w = QR[i]; const int rs = (int)(w >> 16);
                        const int sb = (int)(dk >> dbits); const long long drel = (long long)(dk & DM);
                        const int cs = sb ? -1 : 1;
                        auto rof = [&](int q) -> long long { return sb ? wlo + drel - (q - readstart) : wlo + drel - (readend - 1 - q); };
                        const bool newD = i == 0 || (KH(i - 1) >> LB_QB) != dk;
                        if (newD && qi >= q0) logrec(dk, seq, 0, ekey_of(qi, sb, rof(qi)), 0);          // first hit of the diagonal in this chunk (not a carried one)
                        // anchor [st, q_b + k) whose last hit is b
                        auto anchor = [&](int st, int b) {
                            const uint64_t kb = KH(b); const int qbh = readstart + (int)(kb & QM);
                            const long long len = (long long)qbh + k - st;
                            const long long cr = sb ? rof(qbh) : rof(st);
                            if (b + 1 < nhit && (KH(b + 1) >> LB_QB) == dk) {                              // the next hit on the diagonal appends it (:23241 / :23248)
                                const int q2 = readstart + (int)(KH(b + 1) & QM);
                                emit(st, cr, cs, len, ekey_of(q2, sb, rof(q2)));
                            } else if (qbh + k >= qb && qb < qend) {                                       // may go on in the next chunk
                                const int o = atomicAdd(&s_nop[cur ^ 1], 1);
                                if (o < LB_OPEN) { s_opd[cur ^ 1][o] = dk; s_opq[cur ^ 1][o] = st; s_opp[cur ^ 1][o] = qbh; }
                                else s_fail = 7;
                            } else logrec(dk, seq, 1, ((uint64_t)(unsigned)st << 32) | (uint64_t)(unsigned)len, cr);
                        };
                        if (rs == i) { int st; const int b1 = root_last(i, qi, dk, st); anchor(st, b1); }
                        const bool run_last = i + 1 == nhit || (int)(QR[i + 1] >> 16) != rs;
                        if (((MK[i >> 5] >> (i & 31)) & 1u) && !run_last) anchor(qi + k, (int)NX[i]);
                    }
                    __syncthreads();
                }
                if (s_fail || s_nop[cur ^ 1] > LB_OPEN) { status = VM_READ_BANDFALL_DEV; why = s_fail ? s_fail : 7; break; }
                cur ^= 1; q0 = qb; cj = cj_next; ++seq;
                {   // next chunk: as many positions as fill three quarters of the hit tile at this chunk's hit density (a HiFi read hits at nearly every
                    // position: fixed 512-position chunks overflowed the tile and were done twice, every time)
                    long long want = nhit > 0 ? (long long)qlen * (3 * VMX_LB_HCAP / 4) / nhit : VMX_LB_QC;
                    want &= ~7LL;
                    qc = (int)(want < 32 ? 32 : (want > VMX_LB_QC ? VMX_LB_QC : want));
                }
                if (seq >= (1 << LB_SEQB) - 1) { status = VM_READ_BANDFALL_DEV; why = 9; break; }
                __syncthreads();
                VMX_T(3);
            }
            __syncthreads();
            // ---------------------------------------------------------------- resolve the log: who appends every pending leftover
            {
                const int nlog = status ? 0 : s_nlog;
                int NL = 1; while (NL < nlog) NL <<= 1;
                if (nlog > 0 && (int64_t)NL > hit_cap) { status = VM_READ_BANDFALL_DEV; why = 10; }
                const int nl = status ? 0 : nlog;
                for (int i = nl + tid; i < (nl ? NL : 0); i += T) HKEY[i] = ~0ULL;
                __syncthreads();
                if (nl > 1) lb_sort_hbm(HKEY, NL);
                __syncthreads();
                const unsigned long long IM = (1ULL << ilb) - 1ULL;
                const int dsh = ilb + 1 + LB_SEQB;
                for (int i = tid; i < nl; i += T) {
                    const uint64_t key = HKEY[i];
                    if (!((key >> ilb) & 1ULL)) continue;                                // HEAD
                    const unsigned long long dk = key >> dsh;
                    const int idx = (int)(key & IM);
                    uint64_t ek;
                    if (i + 1 < nl && (HKEY[i + 1] >> dsh) == dk) ek = LOGV0[HKEY[i + 1] & IM];      // the next group's first hit appends it
                    else {                                                               // the diagonal's last: appended at the end, in first-appearance order (:23343)
                        int f = i; while (f > 0 && (HKEY[f - 1] >> dsh) == dk) --f;
                        ek = (1ULL << ekb) | LOGV0[HKEY[f] & IM];
                    }
                    const uint64_t v0 = LOGV0[idx];
                    const int sb = (int)(dk >> dbits);
                    const int o = n_out + atomicAdd(&s_emit, 1);
                    if (o < out_cap) { OUT[o] = vmx_mk_anchor((long long)(v0 >> 32), LOGV1[idx], sb ? -1 : 1, (long long)(v0 & 0xffffffffULL)); OKEY[o] = ek; }
                }
                __syncthreads();
            }
            VMX_T(4);
            // ---------------------------------------------------------------- the reference's append order inside this guide
            {
                const int ng_out = status ? 0 : s_emit;
                if (n_out + ng_out > out_cap) status = VM_READ_CAPACITY_DEV;
                const int ib = 64 - (ekb + 1);
                int NG = 1; while (NG < ng_out) NG <<= 1;
                if (!status && ng_out > 0 && ((int64_t)NG > hit_cap || ((unsigned long long)ng_out >> ib) != 0ULL)) { status = VM_READ_BANDFALL_DEV; why = 10; }
                const int no = status ? 0 : ng_out;
                for (int i = tid; i < (no ? NG : 0); i += T) HKEY[i] = i < no ? ((OKEY[n_out + i] << ib) | (uint64_t)i) : ~0ULL;
                __syncthreads();
                if (no > 1) lb_sort_hbm(HKEY, NG);
                __syncthreads();
                const unsigned long long IBM = (1ULL << ib) - 1ULL;
                for (int e = tid; e < no; e += T) GOFF[n_out + e] = n_out + (int)(HKEY[e] & IBM);   // rank -> anchor index
                __syncthreads();
                n_out += no;
            }
        }
        // --- the stable argsort by q + l (:28585; mode R sorts by read start)
        {
            long long NO = 1; while (NO < n_out) NO <<= 1;
            if (status == 0 && n_out > 0 && NO > hit_cap) { status = VM_READ_BANDFALL_DEV; why = 10; }
            const long long no = status ? 0 : n_out;
            const long long NP = no > 0 ? NO : 0;
            for (long long e = tid; e < NP; e += T) {
                uint64_t kk = ~0ULL;
                if (e < no) { const vmx_anchor a = OUT[GOFF[e]]; kk = ((uint64_t)(uint32_t)(A.sort_by_start ? a.q : a.q + a.l) << 32) | (uint64_t)e; }
                HKEY[e] = kk;
            }
            __syncthreads();
            if (NP > 1) lb_sort_hbm(HKEY, (int)NP);
            __syncthreads();
            for (long long x = tid; x < no; x += T) SORTED[x] = OUT[GOFF[(int)(HKEY[x] & 0xffffffffu)]];
            __syncthreads();
        }
        VMX_T(5);
        if (tid == 0) { A.la_cnt[r] = status ? -why : n_out; A.status[r] = status; }
        __syncthreads();
    }
}
