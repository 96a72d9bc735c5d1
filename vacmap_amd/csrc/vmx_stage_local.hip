// vmx_stage_local.hip — host orchestration of the local stage (L1-L4): vmx_local_stage() is shared by the stage entry
// vm_local_chain_batch and by vm_align_batch. Kernels: k_local.hip.
#include "vmx_host.h"
#include "vmx_local.h"
#include "vmx_stage.h"
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>

using namespace vmx;

__global__ void k_local_prep(const vmx_anchor* path_rows, const int32_t* path_len, const int32_t* n_paths, const int64_t* aoff, const double* gscore,
                             int n_reads, int mode, vmx_anchor* guide_rows, int32_t* guide_len, int32_t* n_guides_used, int32_t* n_guides_total, int32_t* ws_pool);
__global__ void k_la_sizes(const int32_t* la_cnt, int n, int64_t* size, int32_t* n_dev);
__global__ void k_local_seed(vmx_lseed_args A);
__global__ void k_local_seed_band(vmx_lseed_args A);
__global__ void k_chain_local_fast(const vmx_anchor* anchors, const int64_t* la_off, const int32_t* la_cnt, const int32_t* n_guides_total, int n_reads,
                                   const int64_t* roff, vmx_tables tab, const double* gapcost_list, double skip_exact, double skip_mm, int maxdiff, int maxgap,
                                   int mode, double* S_pool, int32_t* P_pool, int32_t* SA_pool, int32_t* si_pool, int64_t* t_pool, int32_t* cnt_pool,
                                   double* out_score, vmx_anchor* out_chain, int32_t* out_len, int32_t* out_variant, int32_t* status);
__global__ void k_chain_local_rows(const vmx_anchor* anchors, const int64_t* la_off, const int32_t* la_cnt, const int32_t* n_guides_total, const int32_t* rlist, int nlist,
                                   int want, vmx_tables tab, const double* gapcost_list, double skip_exact, double skip_mm, int maxdiff, int maxgap, int mode, double* S_pool,
                                   int32_t* P_pool, int32_t* SA_pool, double* out_score, vmx_anchor* out_chain, int32_t* out_len, int32_t* out_variant, int32_t* status,
                                   double* FP_pool, double* PP_pool, unsigned long long* dbg, const int32_t* nlist_dev);
__global__ void k_chain_local_rows_w3(const vmx_anchor* anchors, const int64_t* la_off, const int32_t* la_cnt, const int32_t* n_guides_total, const int32_t* rlist, int nlist,
                                      int want, vmx_tables tab, const double* gapcost_list, double skip_exact, double skip_mm, int maxdiff, int maxgap, int mode, double* S_pool,
                                      int32_t* P_pool, int32_t* SA_pool, double* out_score, vmx_anchor* out_chain, int32_t* out_len, int32_t* out_variant, int32_t* status,
                                      double* FP_pool, double* PP_pool, unsigned long long* dbg, const int32_t* nlist_dev);        // test kernel (VMX_RW_WIN=3)
__global__ void k_chain_local(const vmx_anchor* anchors, const int64_t* la_off, const int32_t* la_cnt, const int32_t* n_guides_total,
                              const int32_t* rlist, int nlist, int lds_cap, vmx_tables tab, const double* gapcost_list, double skip_exact,
                              double skip_mm, int maxdiff, int maxgap, int mode, double* S_pool, int32_t* P_pool, int32_t* SA_pool, double* out_score,
                              vmx_anchor* out_chain, int32_t* out_len, int32_t* out_variant, int32_t* status, double* FP_pool, double* PP_pool);

// Inputs on the device: oriented read codes + offsets; paths (rows at aoff[r], lengths at aoff[r]+p, n_paths[r]); gscore[r] (0 = unmapped).
// h_roff / h_aoff are the host copies of the offsets. Leaves in L: la_off (device+host), chain rows / len / score / variant / status.
int vmx_local_stage(vm_ctx* c, const vm_index_view& ix, const vm_params* prm, int64_t n, const uint8_t* d_ocodes, const int64_t* d_roff,
                    const std::vector<int64_t>& h_roff, const vmx_anchor* d_path_rows, const int32_t* d_path_len, const int32_t* d_npaths,
                    const int64_t* d_aoff, const std::vector<int64_t>& h_aoff, const double* d_gscore, vmx_local_bufs& L, bool fast) {
    // fast (round 6, the batched path after a context's first batch): NO host wait in this stage. The local-anchor counts stay on the device: the read list of the local
    // chain DP is built there (size-class order, most anchors first), a read the banded kernel hands back or whose anchors overflow its slot keeps that status and is
    // run again alone by align_device — through this function with fast = false, where the general kernel and the larger slots are.
    const int k = prm->local_kmersize;
    if (k < 5 || k > 11) { set_error("local k-mer size must be in [5,11]"); return VM_ERR_UNSUPPORTED; }
    if (prm->local_maxdiff > 62) { set_error("local_maxdiff > 62 unsupported"); return VM_ERR_UNSUPPORTED; }
    const int64_t tot_anchors = h_aoff[n];
    int64_t Lmax = 1;
    L.h_la_off.assign((size_t)n + 1, 0);
    for (int64_t r = 0; r < n; ++r) { int64_t len = h_roff[r + 1] - h_roff[r]; Lmax = std::max(Lmax, len); L.h_la_off[r + 1] = L.h_la_off[r] + VMX_LA_SLOT(len); }
    // the regular slot of a read holds len/2 + 4096 local anchors (a 10 % error read yields ~0.07 per base); reads that overflow it are
    // re-run below with slots carved from an overflow area at the end of the pools
    const int64_t la_regular = L.h_la_off[n];
    const int64_t la_overflow = std::max<int64_t>((int64_t)4 << 20, 64 * Lmax);
    const int64_t la_tot = la_regular + la_overflow;
    L.la_pool_rows = la_tot;
    VMX_TRY(L.guide_rows.reserve(sizeof(vmx_anchor) * (size_t)(tot_anchors + 1))); VMX_TRY(L.guide_len.reserve(4 * (size_t)(tot_anchors + 1)));
    VMX_TRY(L.ng_used.reserve(4 * (size_t)(n + 1))); VMX_TRY(L.ng_total.reserve(4 * (size_t)(n + 1))); VMX_TRY(L.prep_ws.reserve(4 * VMX_PREP_WS * (size_t)(tot_anchors + 1)));
    hipLaunchKernelGGL(k_local_prep, dim3((unsigned)std::min<int64_t>(n, (int64_t)c->num_cu * 32)), dim3(64), 0, c->stream, d_path_rows, d_path_len, d_npaths, d_aoff, d_gscore, (int)n,
                       prm->mode, L.guide_rows.as<vmx_anchor>(), L.guide_len.as<int32_t>(), L.ng_used.as<int32_t>(), L.ng_total.as<int32_t>(), L.prep_ws.as<int32_t>());
    // scratch per workgroup slot
    // exactly as many workgroups as the device keeps resident; they pull reads longest-first from a device-side queue
    const int TPB = 512;
    int occ = 1;
#ifndef VMX_EMU
    VMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_local_seed, TPB, 0));
    if (occ < 1) occ = 1;
    // a context that shares the GPU with others (batches in flight) leaves room for their kernels: measured with 3 in flight and the
    // anti-diagonal gap fill, 1 / 2 / 3 workgroups per CU give 34.8 / 33.0 / 35.2 ms per batch (with the slower striped fill one was
    // best: 52.5 / 53.5 / 56.4), while alone 3 is best
    { const int want = c->inflight >= 2 ? 2 : 3; if (want < occ) occ = want; }
    if (const char* e = getenv("VMX_LSEED_WGS")) { int v = atoi(e); if (v >= 1) occ = v; }                  // tuning knob: workgroups per CU
#endif
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)c->num_cu * occ));
    // the guide-banded form (k_local_band.hip) takes every read first; what it hands back (VM_READ_BANDFALL_DEV) runs through k_local_seed
    const bool band_on = [] { const char* e = getenv("VMX_LSEED_BAND"); return !e || atoi(e) != 0; }();      // (read per call: the tests run both kernels in one process)
    int occ_b = 8;
#ifndef VMX_EMU
    if (band_on) {
        VMX_HIP(hipFuncSetAttribute((const void*)k_local_seed_band, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VMX_LB_LDS_BYTES));
        VMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, (const void*)k_local_seed_band, 64, VMX_LB_LDS_BYTES));      // one wavefront per read: the LDS decides (~10 per CU)
        if (occ_b < 1) occ_b = 1;
        if (const char* e = getenv("VMX_LSEED_BAND_WGS")) { int v = atoi(e); if (v >= 1) occ_b = v; }
    }
#endif
    const int GB = (int)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)c->num_cu * occ_b));
    {
        std::vector<int32_t> ord((size_t)n + 1);
        for (int64_t r = 0; r < n; ++r) ord[(size_t)r + 1] = (int32_t)r;
        std::stable_sort(ord.begin() + 1, ord.end(), [&](int32_t a, int32_t b) { return h_roff[a + 1] - h_roff[a] > h_roff[b + 1] - h_roff[b]; });
        ord[0] = 0;                                               // [0] = queue head, [1..n] = read order
        VMX_TRY(vmx_push(c, L.rorder, ord.data(), ord.size()));
    }
    const int64_t nkey = (int64_t)1 << (2 * k);
    const int64_t head_stride = (2 * k > 14 && nkey <= (int64_t)VMX_SORT_LDS * 64) ? ((int64_t)1 << 14) : nkey;     // bucketed heads (k_local_seed)
    int64_t tpos_cap = 2 * Lmax + 5 * 14000 + 65536;
    int64_t hit_cap = 1; while (hit_cap < 4 * (Lmax + 14000)) hit_cap <<= 1;
    int64_t pcnt_cap = Lmax + 16;
    int64_t gkey_cap = 1; { int64_t mx = 1; for (int64_t r = 0; r < n; ++r) mx = std::max(mx, h_aoff[r + 1] - h_aoff[r]); while (gkey_cap < mx) gkey_cap <<= 1; }
    vmx_lseed_args A;
    A.rd_off = nullptr; A.rd_len = nullptr; A.r_st = nullptr; A.r_en = nullptr;
    static const bool dbg_on = getenv("VMX_DBG") != nullptr;
    // one launch over the reads listed in L.rorder[1 .. cnt] with `slots` workgroups and per-slot hit pools of `hcap` entries
    auto run_seed = [&](int cnt, int slots, int64_t hcap, bool band) -> int {
        const int64_t hit_cap = hcap;
        const int G = slots;
        if (!band) {   // head tables: entries are tagged with the slot's epoch (k_local_seed), so only fresh memory is filled (0xff = epoch 511, never current)
            const size_t need = 4 * (size_t)G * (size_t)head_stride;
            const void* before = L.cnt.p; const size_t cap_before = L.cnt.cap;
            const void* ebefore = L.epoch.p;
            VMX_TRY(L.cnt.reserve(need)); VMX_TRY(L.epoch.reserve(4 * (size_t)G + 64));
            if (L.cnt.p != before || L.cnt.cap != cap_before || L.epoch.p != ebefore) {
                VMX_HIP(hipMemsetAsync(L.cnt.p, 0xff, L.cnt.cap, c->stream)); VMX_HIP(hipMemsetAsync(L.epoch.p, 0, L.epoch.cap, c->stream));
            }
        }
        if (!band) {
            VMX_TRY(L.cur.reserve(4 * (size_t)G * (size_t)tpos_cap));
            VMX_TRY(L.sq.reserve(4 * (size_t)G * (size_t)hit_cap)); VMX_TRY(L.dst.reserve(4 * (size_t)G * (size_t)hit_cap));
            VMX_TRY(L.pcnt.reserve(4 * (size_t)G * (size_t)pcnt_cap)); VMX_TRY(L.gq.reserve(4 * (size_t)G * (size_t)gkey_cap)); VMX_TRY(L.gr.reserve(8 * (size_t)G * (size_t)gkey_cap));
            VMX_TRY(L.pc2.reserve(4 * (size_t)G * (size_t)pcnt_cap)); VMX_TRY(L.stg.reserve(16 * (size_t)G * (size_t)pcnt_cap));
        }
        VMX_TRY(L.hkey.reserve(8 * (size_t)G * (size_t)hit_cap));
        VMX_TRY(L.hval.reserve(8 * (size_t)G * (size_t)hit_cap));
        VMX_TRY(L.goff.reserve(4 * (size_t)G * (size_t)hit_cap));
        VMX_TRY(L.gkey.reserve(8 * (size_t)G * (size_t)gkey_cap));
        VMX_TRY(L.la_rows.reserve(sizeof(vmx_anchor) * (size_t)(la_tot + 1))); VMX_TRY(L.la_ekey.reserve(8 * (size_t)(la_tot + 1)));
        VMX_TRY(L.la_sorted.reserve(sizeof(vmx_anchor) * (size_t)(la_tot + 1)));
        VMX_TRY(vmx_push(c, L.la_off, L.h_la_off.data(), (size_t)n + 1));
        VMX_TRY(L.la_cnt.reserve(4 * (size_t)(n + 1))); VMX_TRY(L.status.reserve(4 * (size_t)(n + 1)));
        A.ocodes = d_ocodes; A.roff = d_roff; A.ref = ix.codes; A.coff = ix.coff; A.nseq = ix.nseq;
        A.guide_rows = L.guide_rows.as<vmx_anchor>(); A.guide_len = L.guide_len.as<int32_t>(); A.n_guides_used = L.ng_used.as<int32_t>(); A.aoff = d_aoff;
        A.n_reads = cnt; A.k = k; A.epoch_pool = L.epoch.as<int32_t>();
        const bool narrow = prm->mode == VM_MODE_R || prm->mode == VM_MODE_ASM;                   // mammap_noprefercloser.py:23631+ / mammap_asm.py:18012, :18059-18060, :18237
        A.look_span = narrow ? 2000 : 7000; A.read_span = narrow ? 500 : 7000;   // :23094, :23190
        A.sort_by_start = narrow ? 1 : 0;
        A.queue = L.rorder.as<int32_t>(); A.order = L.rorder.as<int32_t>() + 1;
        A.head_stride = head_stride; A.head_pool = L.cnt.as<int32_t>(); A.next_pool = L.cur.as<int32_t>(); A.sq_pool = L.sq.as<int32_t>(); A.dst_pool = L.dst.as<int32_t>(); A.tpos_pool = nullptr; A.tpos_cap = tpos_cap;          // window positions are implied by the interval list (k_local_seed)
        VMX_TRY(L.hkey2.reserve(8 * (size_t)G * (size_t)hit_cap));
        A.hkey2_pool = L.hkey2.as<uint64_t>();
        A.hkey_pool = L.hkey.as<uint64_t>(); A.hval_pool = L.hval.as<int64_t>(); A.goff_pool = L.goff.as<int32_t>(); A.hit_cap = hit_cap;
        A.pc2_pool = L.pc2.as<int32_t>(); A.stg_pool = L.stg.as<int64_t>();
        A.pcnt_pool = L.pcnt.as<int32_t>(); A.pcnt_cap = pcnt_cap; A.gkey_pool = L.gkey.as<uint64_t>(); A.gq_pool = L.gq.as<int32_t>(); A.gr_pool = L.gr.as<int64_t>();
        A.gkey_cap = gkey_cap;
        A.la_rows = L.la_rows.as<vmx_anchor>(); A.la_ekey = L.la_ekey.as<uint64_t>(); A.la_sorted = L.la_sorted.as<vmx_anchor>(); A.la_off = L.la_off.as<int64_t>();
        A.la_cnt = L.la_cnt.as<int32_t>(); A.status = L.status.as<int32_t>();
        A.dbg = nullptr;
        if (dbg_on) { VMX_TRY(L.dbg.reserve(128)); VMX_HIP(hipMemsetAsync(L.dbg.p, 0, 128, c->stream)); A.dbg = L.dbg.as<unsigned long long>(); }
        if (band) hipLaunchKernelGGL(k_local_seed_band, dim3((unsigned)G), dim3(64), VMX_LB_LDS_BYTES, c->stream, A);
        else hipLaunchKernelGGL(k_local_seed, dim3((unsigned)G), dim3(TPB), 0, c->stream, A);
        if (dbg_on) { unsigned long long h[16]; VMX_TRY(vmx_fetch(c, h, L.dbg.p, 16)); VMX_HIP(vmx_stream_sync(c));
                      fprintf(stderr, band ? "k_local_seed_band phase ticks (100MHz, summed over blocks): guide+windows %llu stream %llu hit sort %llu walk %llu log %llu emission+final sorts %llu | slice+closest+items %llu pieces %llu table %llu | band positions %llu chunks %llu hits %llu read positions %llu\n"
                                           : "k_local_seed phase ticks (100MHz, summed over blocks): guide+windows %llu table %llu passA %llu passB+sort %llu merge %llu finalsort %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12]); }
        return 0;
    };
    A.la_slot_len = 1;
    (void)hipEventRecord(c->kev[0], c->stream);
    // (the banded form keeps no hits in HBM: its per-slot pools hold the chunk log and the anchors' sort keys only)
    int64_t hit_cap_b = 1; while (hit_cap_b < Lmax / 2 + 8192) hit_cap_b <<= 1;        // >= the read's anchor slot (len / 2 + 4096) and its chunk log (~0.25 records per base)
    VMX_TRY(run_seed((int)n, band_on ? GB : G, band_on ? hit_cap_b : hit_cap, band_on));
    (void)hipEventRecord(c->kev[1], c->stream); c->kev_set |= 1;
    const bool rows_kernel = vmx_chain_rows_on() && prm->mode != VM_MODE_ASM;      // four reads per wavefront (k_chain_rows.hip); -mode asm keeps the one-wave kernel
    if (fast && !rows_kernel) { set_error("vmx_local_stage: the fast form needs the row chain kernels"); return VM_ERR_ARG; }
    if (!fast) {
    // sizing sync #2: local anchor counts decide the LDS bucket of every read in the local chain DP
    L.h_la_cnt.resize((size_t)n);
    VMX_TRY(vmx_fetch(c, L.h_la_cnt.data(), L.la_cnt.p, (size_t)n));
    std::vector<int32_t> h_lstatus((size_t)n);
    VMX_TRY(vmx_fetch(c, h_lstatus.data(), L.status.p, (size_t)n));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    if (band_on) {   // reads the banded form handed back (hit tile / open-run list / key width exceeded): the general kernel, same slots
        std::vector<int32_t> ord(1, 0);
        for (int64_t r = 0; r < n; ++r) if (h_lstatus[r] == VM_READ_BANDFALL_DEV) ord.push_back((int32_t)r);
        const int cnt = (int)ord.size() - 1;
        c->n_bandfall += cnt;
        if (getenv("VMX_LSEED_TRACE")) {
            int why[12] = {0};
            for (int i = 1; i <= cnt; ++i) { const int w = -L.h_la_cnt[ord[i]]; ++why[w >= 0 && w < 12 ? w : 11]; }
            fprintf(stderr, "k_local_seed_band: %lld reads, %d handed to k_local_seed (guide sort %d, windows %d, key width %d, intervals %d, pieces %d, hit tile %d, open runs %d, log %d, chunks %d, sorts %d)\n",
                    (long long)n, cnt, why[1], why[2], why[3], why[4], why[5], why[6], why[7], why[8], why[9], why[10]);
        }
        if (cnt) {
            std::stable_sort(ord.begin() + 1, ord.end(), [&](int32_t a, int32_t b) { return h_roff[a + 1] - h_roff[a] > h_roff[b + 1] - h_roff[b]; });
            VMX_TRY(vmx_push(c, L.rorder, ord.data(), ord.size()));
            VMX_TRY(run_seed(cnt, std::min(cnt, G), hit_cap, false));
            VMX_TRY(vmx_fetch(c, L.h_la_cnt.data(), L.la_cnt.p, (size_t)n));
            VMX_TRY(vmx_fetch(c, h_lstatus.data(), L.status.p, (size_t)n));
            VMX_HIP(vmx_stream_sync(c));
            VMX_HIP(hipGetLastError());
        }
    }
    {   // capacity retries: 8x, 64x, 512x, 4096x the regular slot / hit pool, as long as the overflow area lasts
        int64_t ovf_used = 0;
        for (int round = 1; round <= 4; ++round) {
            std::vector<int32_t> ord(1, 0);
            const int64_t mult = (int64_t)1 << (3 * round);
            int64_t lmax_f = 1;
            for (int64_t r = 0; r < n; ++r) {
                if (h_lstatus[r] != VM_READ_CAPACITY_DEV) continue;
                const int64_t len = h_roff[r + 1] - h_roff[r];
                const int64_t want = mult * VMX_LA_SLOT(len);
                if (ovf_used + want > la_overflow) continue;                 // stays a capacity failure
                L.h_la_off[r] = la_regular + ovf_used; ovf_used += want;
                ord.push_back((int32_t)r); lmax_f = std::max(lmax_f, len);
            }
            const int cnt = (int)ord.size() - 1;
            if (!cnt) break;
            VMX_TRY(vmx_push(c, L.rorder, ord.data(), ord.size()));
            VMX_TRY(vmx_push(c, L.la_off, L.h_la_off.data(), (size_t)n + 1));
            int64_t hcap = 1; while (hcap < mult * 4 * (lmax_f + 14000)) hcap <<= 1;
            if (hcap > ((int64_t)1 << 26)) hcap = (int64_t)1 << 26;          // stream indices are 26-bit
            A.la_slot_len = mult;                                            // slot length = mult * VMX_LA_SLOT(len) for the listed reads
            VMX_TRY(run_seed(cnt, std::min(cnt, 8), hcap, false));
            VMX_TRY(vmx_fetch(c, L.h_la_cnt.data(), L.la_cnt.p, (size_t)n));
            VMX_TRY(vmx_fetch(c, h_lstatus.data(), L.status.p, (size_t)n));
            VMX_HIP(vmx_stream_sync(c));
            VMX_HIP(hipGetLastError());
        }
    }
    } else L.h_la_cnt.clear();
    // LC DP
    const HostTables& T = host_tables();
    std::vector<double> gap(64, 0.0);
    for (int g = 1; g <= prm->local_maxdiff; ++g)
        gap[g] = (g <= 10 || prm->mode == VM_MODE_R || prm->mode == VM_MODE_ASM) ? (0.01 * k * g + 0.5 * T.log2int[g]) : (0.01 * k * g + 2 * T.log2int[g]);   // :27317-27322; _scar: 0.5*log2 throughout (mammap_noprefercloser.py:23432)
    VMX_TRY(vmx_push(c, L.gap, gap.data(), 64));
    // LDS buckets by anchor count (24 B per anchor): a workgroup claims only what its read needs, so 4-12 reads share a CU.
    // Inside a bucket the reads are ordered longest first and every read is its own workgroup: the dispatcher hands them out in
    // that order, which is the longest-first dynamic schedule.
    constexpr int NB = 10;
    const int caps[NB] = {512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 13056};
    std::vector<int32_t> lists[NB + 1];
    // reads above lds_max anchors keep their working arrays in HBM (cap 0): a 100 KB LDS claim leaves one wavefront per CU, and a batch
    // of long reads then runs 256 reads at a time (tuning knob VMX_LC_LDS_MAX)
    // Measured on the hg38-size workload, ms per step with 3 batches in flight for limits 13056 / 4096 / 2048 / 1024 / 512 / 0:
    // 46.3 / 46.1 / 44.9 / 43.9 / 44.0 / 43.4 — LDS residency pays for a context that runs alone, not when batches share the GPU.
    static const int lds_env = [] { const char* e = getenv("VMX_LC_LDS_MAX"); return e ? atoi(e) : -1; }();
    const int lds_max = lds_env >= 0 ? lds_env : VMX_CHAIN_LDS_MAX_SHARED;
    for (int64_t r = 0; r < n && !fast; ++r) {
        int m = L.h_la_cnt[r];
        if (m <= 0) continue;   // no guide / capacity failure: nothing to chain (status already set by the seeding kernel or stays 0 for unmapped reads)
        int bk = NB; for (int q = 0; q < NB; ++q) if (m <= caps[q] && caps[q] <= lds_max) { bk = q; break; }
        if (rows_kernel) bk = NB;                                         // k_chain_local_rows: one list, nothing in LDS
        lists[bk].push_back((int32_t)r);
    }
    std::vector<int32_t> rl; int64_t rl_off[NB + 2];
    for (int q = 0; q <= NB; ++q) {
        std::stable_sort(lists[q].begin(), lists[q].end(), [&](int32_t a, int32_t b) { return L.h_la_cnt[a] > L.h_la_cnt[b]; });
        rl_off[q] = (int64_t)rl.size(); rl.insert(rl.end(), lists[q].begin(), lists[q].end());
    }
    rl_off[NB + 1] = (int64_t)rl.size();
    const int32_t* d_nlist = nullptr;
    if (fast) {
        // the list on the device: L.rlist = [read order (n) | queue range (4) | counters (4) | n (4) | scratch (513, zeroed)], keys in L.hq
        VMX_TRY(L.rlist.reserve(4 * (size_t)(n + 16 + 520))); VMX_TRY(L.hq.reserve(8 * (size_t)(n + 1)));
        int32_t* ord = L.rlist.as<int32_t>(); int32_t* rng = ord + n; int32_t* ctr = rng + 4; int32_t* ndev = ctr + 4; int32_t* scratch = ndev + 4;
        VMX_HIP(hipMemsetAsync(scratch, 0, 4 * 513, c->stream));
        hipLaunchKernelGGL(k_la_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, L.la_cnt.as<int32_t>(), (int)n, L.hq.as<int64_t>(), ndev);
        const unsigned Gs = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 1023) / 1024, (int64_t)c->num_cu * 2));
        hipLaunchKernelGGL(k_size_hist, dim3(Gs), dim3(256), 0, c->stream, L.hq.as<int64_t>(), (const int32_t*)ndev, (int64_t)1 << 62, scratch);
        hipLaunchKernelGGL(k_size_scatter, dim3(Gs), dim3(256), 0, c->stream, L.hq.as<int64_t>(), (const int32_t*)ndev, (const int32_t*)scratch, scratch + 257, ord, rng, ctr);
        d_nlist = rng + 1;                                            // reads queued (those with local anchors)
        lists[NB].assign(1, 0);                                       // (one launch per variant below, over the upper bound n)
        rl_off[NB] = 0;
    } else
    VMX_TRY(vmx_push(c, L.rlist, rl.data(), rl.size()));
    VMX_TRY(L.S.reserve(8 * (size_t)(la_tot + 1))); VMX_TRY(L.P.reserve(4 * (size_t)(la_tot + 1))); VMX_TRY(L.SA.reserve(4 * (size_t)(la_tot + 1)));
    VMX_TRY(L.chain.reserve(sizeof(vmx_anchor) * (size_t)(la_tot + 1)));
    VMX_TRY(L.chain_len.reserve(4 * (size_t)(n + 1))); VMX_TRY(L.score.reserve(8 * (size_t)(n + 1))); VMX_TRY(L.variant.reserve(4 * (size_t)(n + 1)));
    VMX_HIP(hipMemsetAsync(L.chain_len.p, 0, 4 * (size_t)n, c->stream)); VMX_HIP(hipMemsetAsync(L.score.p, 0, 8 * (size_t)n, c->stream));
    VMX_HIP(hipMemsetAsync(L.variant.p, 0, 4 * (size_t)n, c->stream));
    const int maxgap = prm->mode == VM_MODE_L ? 50 : 99;                                        // :24061 / mammap_ccs.py:24061
    const double skip_exact = prm->local_skipcost;
    const double skip_mm = prm->mode == VM_MODE_L ? std::min(prm->local_skipcost, 40.0) : prm->local_skipcost;   // mammap_ccs.py:28587
#ifndef VMX_EMU
    VMX_HIP(hipFuncSetAttribute((const void*)k_chain_local, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)caps[NB - 1] * VMX_LC_BYTES_PER_ANCHOR + 64)));
#endif
    if (prm->mode == VM_MODE_R) { VMX_TRY(L.fp.reserve(8 * (size_t)(la_tot + 1))); VMX_TRY(L.pp.reserve(8 * (size_t)(la_tot + 1))); }   // _scar's fixed_penatly / pre_penatly
    vmx_fork fk(c);                                                   // independent LDS buckets side by side, largest reads first
    for (int q = NB; q >= 0; --q) {
        int cnt = (int)lists[q].size();
        if (!cnt) continue;
        if (fast) cnt = (int)n;
        int cap = q < NB ? caps[q] : 0;
        size_t shmem = (size_t)cap * VMX_LC_BYTES_PER_ANCHOR + 64;
        if (rows_kernel) {
            // one launch per variant (different code; a row whose read wants the other variant leaves at once): LC-exact and LC-mm, or `_scar` alone in mode R
            for (int want = (prm->mode == VM_MODE_R ? 2 : 0); want <= (prm->mode == VM_MODE_R ? 2 : 1); ++want)
                hipLaunchKernelGGL((vmx_chain_rows_win3() ? k_chain_local_rows_w3 : k_chain_local_rows), dim3((unsigned)((cnt + 3) / 4)), dim3(64), 0, fk.next(), L.la_sorted.as<vmx_anchor>(),
                                   L.la_off.as<int64_t>(), L.la_cnt.as<int32_t>(), L.ng_total.as<int32_t>(), L.rlist.as<int32_t>() + rl_off[q], cnt, want, c->tables,
                                   L.gap.as<double>(), skip_exact, skip_mm, prm->local_maxdiff, maxgap, prm->mode, L.S.as<double>(), L.P.as<int32_t>(), L.SA.as<int32_t>(),
                                   L.score.as<double>(), L.chain.as<vmx_anchor>(), L.chain_len.as<int32_t>(), L.variant.as<int32_t>(), L.status.as<int32_t>(),
                                   L.fp.as<double>(), L.pp.as<double>(), vmx_chain_dbg(), d_nlist);
            continue;
        }
        hipLaunchKernelGGL(k_chain_local, dim3((unsigned)cnt), dim3(64), shmem, fk.next(), L.la_sorted.as<vmx_anchor>(),
                           L.la_off.as<int64_t>(), L.la_cnt.as<int32_t>(), L.ng_total.as<int32_t>(), L.rlist.as<int32_t>() + rl_off[q], cnt, cap, c->tables,
                           L.gap.as<double>(), skip_exact, skip_mm, prm->local_maxdiff, maxgap, prm->mode, L.S.as<double>(), L.P.as<int32_t>(), L.SA.as<int32_t>(),
                           L.score.as<double>(), L.chain.as<vmx_anchor>(), L.chain_len.as<int32_t>(), L.variant.as<int32_t>(), L.status.as<int32_t>(),
                           L.fp.as<double>(), L.pp.as<double>());
    }
    fk.join();
    vmx_chain_dbg_report(c->stream);
    // L5: reads whose LC launch hit the opcount switch (:27380 / :28333) take the *_fast twin. One wave per read; all other reads
    // return at once.
    VMX_TRY(L.si.reserve(4 * (size_t)(la_tot + 1))); VMX_TRY(L.tg.reserve(8 * (size_t)(la_tot + 1))); VMX_TRY(L.cntp.reserve(4 * (size_t)(h_roff[n] + 50 * n + 64)));
    hipLaunchKernelGGL(k_chain_local_fast, dim3((unsigned)n), dim3(64), 0, c->stream, L.la_sorted.as<vmx_anchor>(), L.la_off.as<int64_t>(), L.la_cnt.as<int32_t>(),
                       L.ng_total.as<int32_t>(), (int)n, d_roff, c->tables, L.gap.as<double>(), skip_exact, skip_mm, prm->local_maxdiff, maxgap, prm->mode,
                       L.S.as<double>(), L.P.as<int32_t>(), L.SA.as<int32_t>(), L.si.as<int32_t>(), L.tg.as<int64_t>(), L.cntp.as<int32_t>(),
                       L.score.as<double>(), L.chain.as<vmx_anchor>(), L.chain_len.as<int32_t>(), L.variant.as<int32_t>(), L.status.as<int32_t>());
    return 0;
}

extern "C" {

void vm_local_out_free(vm_local_out* o) {
    free(o->status); free(o->variant); free(o->score); free(o->chain_off); free(o->chain); free(o->raw_off); free(o->raw);
    memset(o, 0, sizeof(*o));
}

int vm_local_chain_batch(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets,
                         const int64_t* read_path_off, const int64_t* path_off, const int64_t* path_anchors, vm_local_out* out) {
    memset(out, 0, sizeof(*out));
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    vm_index_view ix; vmx_index_view(mi, &ix);
    // reads
    DevBuf &raw = c->b[0], &codes = c->b[1], &roff = c->b[2];
    const int64_t tot = offsets[n];
    VMX_TRY(upload(raw, seqs, (size_t)tot, c->stream)); VMX_TRY(codes.reserve((size_t)tot + 64)); VMX_TRY(upload(roff, offsets, (size_t)n + 1, c->stream));
    if (tot) hipLaunchKernelGGL(k_encode, dim3((unsigned)std::min<int64_t>((tot + 255) / 256, 4096)), dim3(256), 0, c->stream, raw.as<char>(), codes.as<uint8_t>(), tot);
    // paths -> device layout (rows at aoff[r], lengths at aoff[r]+p)
    std::vector<int64_t> h_roff(offsets, offsets + n + 1), h_aoff((size_t)n + 1, 0);
    for (int64_t r = 0; r < n; ++r) { int64_t rows = path_off[read_path_off[r + 1]] - path_off[read_path_off[r]]; h_aoff[r + 1] = h_aoff[r] + rows; }
    const int64_t ta = h_aoff[n];
    std::vector<vmx_anchor> rows((size_t)ta + 1); std::vector<int32_t> plen((size_t)ta + 1, 0), npaths((size_t)n); std::vector<double> gscore((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        int64_t p0 = read_path_off[r], p1 = read_path_off[r + 1];
        npaths[r] = (int32_t)(p1 - p0); gscore[r] = p1 > p0 ? 1.0 : 0.0;
        int64_t w = h_aoff[r];
        for (int64_t p = p0; p < p1; ++p) {
            plen[h_aoff[r] + (p - p0)] = (int32_t)(path_off[p + 1] - path_off[p]);
            for (int64_t x = path_off[p]; x < path_off[p + 1]; ++x) { vmx_anchor a; a.q = (int32_t)path_anchors[4 * x]; a.r = path_anchors[4 * x + 1]; a.s = (int16_t)path_anchors[4 * x + 2]; a.l = (int16_t)path_anchors[4 * x + 3]; rows[w++] = a; }
        }
    }
    DevBuf &d_rows = c->b[3], &d_plen = c->b[4], &d_np = c->b[5], &d_aoff = c->b[6], &d_gs = c->b[7];
    VMX_TRY(upload(d_rows, rows.data(), (size_t)ta + 1, c->stream)); VMX_TRY(upload(d_plen, plen.data(), (size_t)ta + 1, c->stream));
    VMX_TRY(upload(d_np, npaths.data(), (size_t)n, c->stream)); VMX_TRY(upload(d_aoff, h_aoff.data(), (size_t)n + 1, c->stream));
    VMX_TRY(upload(d_gs, gscore.data(), (size_t)n, c->stream));
    vmx_local_bufs& L = *vmx_ctx_local_bufs(c);
    VMX_TRY(vmx_local_stage(c, ix, prm, n, codes.as<uint8_t>(), roff.as<int64_t>(), h_roff, d_rows.as<vmx_anchor>(), d_plen.as<int32_t>(), d_np.as<int32_t>(),
                            d_aoff.as<int64_t>(), h_aoff, d_gs.as<double>(), L));
    // download
    const int64_t la_tot = L.la_pool_rows;
    std::vector<int32_t> clen((size_t)n); std::vector<vmx_anchor> chain((size_t)la_tot), sorted((size_t)la_tot);
    out->status = (int32_t*)malloc(4 * (size_t)std::max<int64_t>(n, 1)); out->variant = (int32_t*)malloc(4 * (size_t)std::max<int64_t>(n, 1));
    out->score = (double*)malloc(8 * (size_t)std::max<int64_t>(n, 1));
    VMX_TRY(download(out->status, L.status.p, (size_t)n, c->stream)); VMX_TRY(download(out->variant, L.variant.p, (size_t)n, c->stream));
    VMX_TRY(download(out->score, L.score.p, (size_t)n, c->stream)); VMX_TRY(download(clen.data(), L.chain_len.p, (size_t)n, c->stream));
    VMX_TRY(download(chain.data(), L.chain.p, (size_t)la_tot, c->stream)); VMX_TRY(download(sorted.data(), L.la_sorted.p, (size_t)la_tot, c->stream));
    VMX_HIP(vmx_stream_sync(c)); VMX_HIP(hipGetLastError());
    int64_t tc = 0, tr = 0;
    for (int64_t r = 0; r < n; ++r) { tc += clen[r]; tr += L.h_la_cnt[r]; }
    out->chain_off = (int64_t*)malloc(8 * (size_t)(n + 1)); out->raw_off = (int64_t*)malloc(8 * (size_t)(n + 1));
    out->chain = (int64_t*)malloc(32 * (size_t)std::max<int64_t>(tc, 1)); out->raw = (int64_t*)malloc(32 * (size_t)std::max<int64_t>(tr, 1));
    int64_t oc = 0, orw = 0;
    for (int64_t r = 0; r < n; ++r) {
        out->chain_off[r] = oc; out->raw_off[r] = orw;
        for (int x = 0; x < clen[r]; ++x) { const vmx_anchor& a = chain[L.h_la_off[r] + x]; int64_t* o = out->chain + 4 * oc++; o[0] = a.q; o[1] = a.r; o[2] = a.s; o[3] = a.l; }
        for (int x = 0; x < L.h_la_cnt[r]; ++x) { const vmx_anchor& a = sorted[L.h_la_off[r] + x]; int64_t* o = out->raw + 4 * orw++; o[0] = a.q; o[1] = a.r; o[2] = a.s; o[3] = a.l; }
    }
    out->chain_off[n] = oc; out->raw_off[n] = orw;
    return VM_OK;
}

}  // extern "C"

vmx_local_bufs* vmx_ctx_local_bufs(vm_ctx* c) {
    if (!c->lbufs) c->lbufs = new vmx_local_bufs();
    return c->lbufs;
}
void vmx_ctx_free_local_bufs(vm_ctx* c) {
    if (c->lbufs) { c->lbufs->release(); delete c->lbufs; c->lbufs = nullptr; }
}
