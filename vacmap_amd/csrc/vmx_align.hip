// vmx_align.hip — the batched path: vm_align_batch / vm_reads_upload / vm_align_resident (include/vacmapx.h).
// Replaces get_readmap_DP_test (/root/reference/src/vacmap/mammap_clrnano.py:24023-24084) for a whole batch of reads: seed -> global
// chain -> local re-seed + chain -> extend, every stage a HIP kernel on the context's stream; the host only sizes buffers
// (a few small device->host reads of totals per batch) and launches. No CPU compute path exists.
#include "vmx_host.h"
#include "vmx_stage.h"
#include "vmx_select.h"
#include "vmx_ext_state.h"
#include "vmx_round.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <string>

using namespace vmx;

__global__ void k_orient(const uint8_t* codes, const int64_t* roff, const double* gscore, int n_reads, uint8_t* ocodes);
__global__ void k_scan_i64(const int64_t* in, int64_t* out, int64_t n, int pow2_round);
__global__ void k_ext_phase(vmx_ext_args A, int phase);
__global__ void k_ed_anchor_bound(vmx_ext_args A, const int32_t* probread, const uint8_t* qpool, const int64_t* qoff, const uint8_t* tpool, const int64_t* toff,
                                  int64_t* ub_out);
__global__ void k_desc_lens(const vmx_pair_desc* desc, const int32_t* n_prob, int64_t* tl, int64_t* ql);
__global__ void k_gather(const vmx_pair_desc* desc, const int32_t* n_prob, const int32_t* prob_read, const uint8_t* ocodes, const int64_t* roff,
                         const uint8_t* ref, const int64_t* t_off, const int64_t* q_off, uint8_t* tpool, uint8_t* qpool, int64_t pool_cap, vmx_ext_read* er);
__global__ void k_prob_owner(const vmx_ext_read* er, int n_reads, int use_dp, int32_t* prob_read);
__global__ void k_dp_sizes(const vmx_pair_desc* desc, const int32_t* n_prob, int64_t* tb_sz, int64_t* bnd_sz, int64_t* run_sz, int64_t* cig_sz);
__global__ void k_dp_table(const vmx_pair_desc* desc, const int32_t* n_prob, const int64_t* t_off, const int64_t* q_off, const int64_t* tb_off,
                           const int64_t* bnd_off, const int64_t* run_off, const int64_t* cig_off, vmx_dp_prob* probs);
__global__ void k_ext_records(vmx_ext_args A, const vmx_dp_prob* probs, const char* cig_pool, const int32_t* cig_len, const int32_t* cig_q);
__global__ void k_res_sizes(const vmx_ext_read* er, const vm_record* rec, const int64_t* soff, int n_reads, int64_t* recn, int64_t* blobn);
__global__ void k_res_pack(const vmx_ext_read* er, const vm_record* rec, const char* blob, const int64_t* soff, const int64_t* blob_off, int n_reads,
                           const int64_t* rec_o, const int64_t* blob_o, vm_record* out_rec, char* out_blob);

// ---- small utility kernels
__global__ void k_compact_rows(const int64_t* __restrict__ rows, const int64_t* __restrict__ koff, const int64_t* __restrict__ aoff, int n_reads, int64_t* __restrict__ out) {
    for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const int64_t n4 = 4 * (aoff[r + 1] - aoff[r]);
        const int64_t* s = rows + 4 * koff[r]; int64_t* d = out + 4 * aoff[r];
        for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) d[i] = s[i];
    }
}
__global__ void k_i32_to_i64(const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_readlens(const int64_t* __restrict__ roff, int64_t* __restrict__ lens, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) lens[i] = roff[i + 1] - roff[i];
}

// scalars the host wants are gathered by one-thread kernels into one block and read back with ONE copy (a hipMemcpyAsync per scalar is a
// runtime copy kernel each: ~100 of them per batch before)
__global__ void k_stat_put(const int32_t* __restrict__ src, int32_t* __restrict__ dst) { *dst = *src; }
// (gap-fill sizing on the device — pool totals and the cut of the problems into traceback chunks — is the last step of k_round_prep<true>, k_round.hip)
__global__ void k_final_scalars(const int64_t* nr, const int64_t* nb, const int32_t* oflow, const int32_t* edc, const int32_t* rounds, int64_t* __restrict__ out) {
    out[0] = *nr; out[1] = *nb; out[2] = *oflow; out[3] = edc[0]; out[4] = edc[1]; out[5] = edc[2];
    for (int i = 0; i < 8; ++i) out[6 + i] = rounds[i];
}

struct vm_reads { vm_ctx* ctx; int64_t n; std::vector<int64_t> h_off; DevBuf raw, codes, off; };

struct vmx_batch_bufs {
    DevBuf seed[13];
    DevBuf nanc64, aoff, rows, lens, keys, koff, sorted, flip, S, P, SA, cov, gmax, opc, rl, gap, scr, soff, res, plen, prow, ocodes;
    // extend stage
    DevBuf er, coff3, soff2, segA, st, en, segA_s, st_s, en_s, segprob, dup, desc[2], rcount, oflow, probread, tl, ql, toff, qoff, tpool, qpool;
    DevBuf edout, carry, ext3, dpsz[4], dpoff[4], dptab, tb, tbredo, bnd, run, cig, ciglen, dpscore, rec, blob, bloboff, reccoff, recclen, dupd, totals;
    DevBuf raw, codes, off, order, qrange, scanpart, scanoff, si, tg, cntp, fp, pp, chunkn, sellist, cigq, statblk, szh, side_codes, side_off, roundpart, gfctl, geotot;
    void release() { DevBuf* p = (DevBuf*)this; for (size_t i = 0; i < sizeof(*this) / sizeof(DevBuf); ++i) p[i].release(); }
};
static vmx_batch_bufs* batch_bufs(vm_ctx* c) { if (!c->bbufs) c->bbufs = new vmx_batch_bufs(); return c->bbufs; }
void vmx_ctx_free_batch_bufs(vm_ctx* c) { if (c->bbufs) { c->bbufs->release(); delete c->bbufs; c->bbufs = nullptr; } }

// k_chain_select by LDS size class: a read claims 17 B of LDS per anchor for its serial peel, so a launch whose reads have at most `cap`
// anchors asks for exactly that (a fixed 52 KB claim kept three waves on a CU — and next to another batch's kernels often none);
// reads beyond the largest class keep everything in HBM. Inside a class the reads with most anchors go first.
int vmx_launch_chain_select(vm_ctx* c, int64_t n, const int64_t* h_aoff, DevBuf& d_list, const vmx_anchor* sorted, const int64_t* d_aoff, const int64_t* d_lens,
                            const double* S, const int32_t* P, const int32_t* SA, const int64_t* gmax, const int32_t* flip, int mode, char* scr, const int64_t* soff,
                            int32_t* d_mapq, double* d_score, int32_t* d_np, int32_t* plen, vmx_anchor* prow) {
    if (n <= 0) return 0;
    constexpr int NC = 6;
    const int caps[NC] = {192, 384, 768, 1536, 3072, 0};
    static const bool serial_peel = getenv("VMX_SELECT_SERIAL") != nullptr;      // A/B knob: the one-lane peel of round 3
    std::vector<int32_t> lists[NC];
    for (int64_t r = 0; r < n; ++r) {
        const int64_t m = h_aoff[r + 1] - h_aoff[r];
        int q = NC - 1; for (int i = 0; i < NC - 1; ++i) if (m <= caps[i]) { q = i; break; }
        lists[q].push_back((int32_t)r);
    }
    std::vector<int32_t> rl; size_t off[NC + 1];
    for (int q = 0; q < NC; ++q) {
        std::stable_sort(lists[q].begin(), lists[q].end(), [&](int32_t a, int32_t b) { return h_aoff[a + 1] - h_aoff[a] > h_aoff[b + 1] - h_aoff[b]; });
        off[q] = rl.size(); rl.insert(rl.end(), lists[q].begin(), lists[q].end());
    }
    VMX_TRY(vmx_push(c, d_list, rl.data(), rl.size()));
    vmx_fork fk(c);
    for (int q = NC - 1; q >= 0; --q) {                          // the classes with the longest walks first
        const int cnt = (int)lists[q].size(); if (!cnt) continue;
        hipLaunchKernelGGL(k_chain_select, dim3((unsigned)cnt), dim3(64), (size_t)caps[q] * 17 + 64, fk.next(), sorted, d_aoff, serial_peel ? nullptr : d_lens, d_list.as<int32_t>() + off[q], cnt, caps[q],
                           S, P, SA, gmax, flip, mode, scr, soff, d_mapq, d_score, d_np, plen, prow);
    }
    fk.join();
    return 0;
}

#define LAUNCH1D(kernel, n, ...) hipLaunchKernelGGL(kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(((n) + 255) / 256, 4096))), dim3(256), 0, c->stream, __VA_ARGS__)

__global__ void k_ext_geometry(const int32_t* la_cnt, const int64_t* roff, int n, int mul, int tmask, long long tdiv, long long capA, long long capS, long long capB, int64_t* coff3,
                               int64_t* soff2, int64_t* bloboff, int64_t* tot);
__global__ void k_scan_part_dev(const int64_t* in, int64_t* part, const int32_t* n_ptr);
__global__ void k_scan_apply_dev(const int64_t* in, int64_t* out, const int64_t* part_off, const int32_t* n_ptr);
__global__ void k_scan_part(const int64_t* in, int64_t* part, int64_t n, int64_t chunk);
__global__ void k_scan_apply(const int64_t* in, int64_t* out, const int64_t* part_off, int64_t n, int64_t chunk);

// exclusive scan on the stream (out[n] = total): one workgroup for small n, three phases otherwise
static int dev_scan(vm_ctx* c, vmx_batch_bufs& B, const int64_t* in, int64_t* out, int64_t n) {
    if (n <= 16384) { hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, in, out, n, 0); return 0; }
    int64_t chunk = (n + 511) / 512; const int64_t nb = (n + chunk - 1) / chunk;
    VMX_TRY(B.scanpart.reserve(8 * (size_t)(nb + 2))); VMX_TRY(B.scanoff.reserve(8 * (size_t)(nb + 2)));
    hipLaunchKernelGGL(k_scan_part, dim3((unsigned)nb), dim3(256), 0, c->stream, in, B.scanpart.as<int64_t>(), n, chunk);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, B.scanpart.as<int64_t>(), B.scanoff.as<int64_t>(), nb, 0);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, in, out, B.scanoff.as<int64_t>(), n, chunk);
    return 0;
}

// the same with the element count on the device (*n_ptr <= cap): no host read-back; out[n] = total
static int dev_scan_dev(vm_ctx* c, vmx_batch_bufs& B, const int64_t* in, int64_t* out, const int32_t* n_ptr) {
    const int nb = 512;
    VMX_TRY(B.scanpart.reserve(8 * (size_t)(nb + 2))); VMX_TRY(B.scanoff.reserve(8 * (size_t)(nb + 2)));
    hipLaunchKernelGGL(k_scan_part_dev, dim3(nb), dim3(256), 0, c->stream, in, B.scanpart.as<int64_t>(), n_ptr);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, B.scanpart.as<int64_t>(), B.scanoff.as<int64_t>(), (int64_t)nb, 0);
    hipLaunchKernelGGL(k_scan_apply_dev, dim3(nb), dim3(256), 0, c->stream, in, out, B.scanoff.as<int64_t>(), n_ptr);
    return 0;
}

// one DP round of the extend stage: descriptors (count on device) -> offsets -> gathered pools. The round's problem count stays on the
// device (B.rcount; every consumer reads it there and launches a grid sized for the hardware, not for the count); it is copied into
// slot `stat_slot` of the batch's counter block, which the host reads once at the end. want_cnt: also return a host copy (one wait) —
// only the gap-fill rounds, whose pools are sized by it, ask for that.
static int ext_gather_round(vm_ctx* c, vmx_batch_bufs& B, const vm_index_view& ix, int64_t n, const uint8_t* d_ocodes, const int64_t* d_roff, int cur, int redo_only,
                            int64_t round_cap, int64_t pool_cap, int stat_slot, bool dp, int64_t tb_limit, const int64_t* caps = nullptr, int64_t* plan_out = nullptr, int ad_pct = 0) {
    (void)n; (void)redo_only; (void)round_cap;    // (the slots' owners are written when the phase kernel allocates them; vmx_alloc_probs never lets the published count pass the capacity)
    if (!B.roundpart.p) { VMX_TRY(B.roundpart.reserve(8 * (size_t)VMX_ROUND_PART_WORDS)); VMX_HIP(hipMemsetAsync(B.roundpart.p, 0, 8 * (size_t)VMX_ROUND_PART_WORDS, c->stream)); }
    vmx_round_args R; memset(&R, 0, sizeof R);
    R.desc = B.desc[cur].as<vmx_pair_desc>(); R.n_prob = c->rc_cur;
    R.off[0] = B.toff.as<int64_t>(); R.off[1] = B.qoff.as<int64_t>();
    R.part = B.roundpart.as<int64_t>(); R.stat_out = B.statblk.as<int32_t>() + stat_slot; R.epoch = ++c->round_epoch;
    if (dp) {
        for (int i = 0; i < 4; ++i) R.off[2 + i] = B.dpoff[i].as<int64_t>();
        R.tb_size = B.dpsz[0].as<int64_t>(); R.probs = B.dptab.as<vmx_dp_prob>(); R.plan_out = plan_out; R.tb_limit = tb_limit;
        if (caps) for (int i = 0; i < 4; ++i) R.cap[i] = caps[i];          // unplanned pass: every problem is checked against the pools (k_round.hip)
        R.ad_on = 1; R.ad_match = 2; R.ad_o1 = 4; R.ad_e1 = 2; R.ad_o2 = 24; R.ad_e2 = 1; R.ad_pct = ad_pct;      // the fill's own scoring and band-width rule (gf_chunk)
        hipLaunchKernelGGL(k_round_prep<true>, dim3(VMX_ROUND_WGS), dim3(256), 0, c->stream, R);
    } else
        hipLaunchKernelGGL(k_round_prep<false>, dim3(VMX_ROUND_WGS), dim3(256), 0, c->stream, R);
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((int64_t)c->num_cu * 8)), dim3(256), 0, c->stream, B.desc[cur].as<vmx_pair_desc>(), c->rc_cur,
                       B.probread.as<int32_t>(), d_ocodes, d_roff, ix.codes, B.toff.as<int64_t>(), B.qoff.as<int64_t>(), B.tpool.as<uint8_t>(), B.qpool.as<uint8_t>(), pool_cap,
                       B.er.as<vmx_ext_read>());
    return 0;
}

// diagnostic capture for the stage tests (vm_align_trace): the segment lists of every read as they stand after one phase of the
// extend stage in the first (filtering) pass: rows (segment, q, r, s, l), off[n + 1]
struct vmx_seg_trace { int stage; std::vector<int64_t> rows, off; };
int vmx_asm_resolve_ties(vm_ctx* c, const vm_index* mi, int64_t n, const std::vector<int64_t>& h_roff, const std::vector<int64_t>& h_aoff, const uint8_t* d_codes,
                         const vmx_anchor* d_sorted, const double* d_S, const int32_t* d_P, const int32_t* d_SA, const int64_t* d_gmax, const int32_t* d_flip,
                         int32_t* d_mapq, double* d_gscore, int32_t* d_np, int32_t* d_plen, vmx_anchor* d_prow, std::vector<int32_t>& status_override);
int vmx_align_batch_asm_mixed(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets, vm_record** recs, int64_t* n_recs,
                              char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats);
int align_device(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, const std::vector<int64_t>& h_roff,
                 vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats, vmx_seg_trace* trace = nullptr, const vmx_preset* preset = nullptr);

#define VMX_EXT_EXACT_INTERNAL (-9002)   // ... "this read needs a part of the path that a batch does not run: the later tiers of the divergence filter, or pass 1" (never leaves align_device)
#define VMX_RETRY_BATCH (-9003)          // align_device_once -> align_device: an assumption the batch ran on instead of a host wait did not hold (a pool sized from the context's
                                        // history, the unplanned pass 1): the pools have been grown / the assumption dropped, run the batch again
#define VMX_EXT_SHORT_INTERNAL (-9001)   // align_device_once's per-read status for "an extend-stage pool was too small for this read" (never leaves align_device)
static int align_device_once(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, const std::vector<int64_t>& h_roff,
                             vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats, vmx_seg_trace* trace, const vmx_preset* preset);
// The pools of the extend stage are sized from bounds that hold by construction for the reference's segment surgery (a chain never has more segments than
// anchors, a round never more problems than anchors, ...) — and where a bound is an estimate (the problem strings of a round: 6 bytes per read base) or a
// construction turns out wrong for some input, THE READ reports it (VMX_EXT_CAPACITY_DEV; round 6: k_gather's string pools mark the owning read as well).
// Round 6: the batch is NOT run again as a whole with every pool x4 — that multiplied ~10 GB of grow-only pools for all 4096 reads because of one
// read, could exhaust the HBM next to the other contexts, and left the pools inflated for the life of the context. The reads that reported it (a handful, if
// any) are run again ALONE as a side batch with their pools x4, x16, ... x1024: a few reads' pools times 1024 still fit inside what the whole batch holds, so
// nothing grows; their records are spliced into the batch's. A read that is still short at x1024 — or whose side batch finds no device memory — is reported as
// VM_READ_CAPACITY; the rest of the batch is unaffected. VMX_TEST_EXT_POOL=<mask>:<div> (tests) divides the pools the mask names (1 segment anchors, 2 segments,
// 4 record blob, 8 problems per round, 16 problem strings) so that each of these ends is exercised.
__global__ void k_side_codes(const uint8_t* __restrict__ codes, const int64_t* __restrict__ roff, const int32_t* __restrict__ pick, const int64_t* __restrict__ soff, int n, uint8_t* __restrict__ out) {
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t a = roff[pick[j]], len = roff[pick[j] + 1] - a; const uint8_t* s = codes + a; uint8_t* d = out + soff[j];
        for (int64_t i = threadIdx.x; i < len; i += blockDim.x) d[i] = s[i];
    }
}
int align_device(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, const std::vector<int64_t>& h_roff,
                 vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats, vmx_seg_trace* trace, const vmx_preset* preset) {
    c->ext_mul = 1;
    std::vector<int32_t> status((size_t)n + 1, 0);
    vm_batch_stats st; memset(&st, 0, sizeof st);
    int rc = VMX_RETRY_BATCH; int64_t batch_retries = 0;
    for (int attempt = 0; attempt < 6 && rc == VMX_RETRY_BATCH; ++attempt) {
        if (attempt) { ++batch_retries; if (trace) { trace->rows.clear(); trace->off.clear(); } }
        rc = align_device_once(c, mi, prm, n, d_codes, d_roff, h_roff, recs, n_recs, cigar_blob, status.data(), &st, trace, preset);
    }
    if (rc == VMX_RETRY_BATCH) { set_error("the batch was run again five times and its pools still did not hold it"); return VM_ERR_OOM; }
    if (rc < 0) return rc;
    st.n_batch_retries = batch_retries;
    std::vector<int32_t> sub;                                    // reads of the batch that were short of an extend-stage pool
    for (int64_t r = 0; r < n; ++r) if (status[(size_t)r] == VMX_EXT_SHORT_INTERNAL || status[(size_t)r] == VMX_EXT_EXACT_INTERNAL) sub.push_back((int32_t)r);
    if (!sub.empty()) {
        vmx_batch_bufs& B = *batch_bufs(c);
        std::vector<vm_record> all; all.reserve((size_t)*n_recs);
        {   // (a read that is run again delivers its records from there: the batch's own — none, for every real reason to run a read again — are dropped)
            std::vector<char> again((size_t)n + 1, 0); for (int32_t r : sub) again[(size_t)r] = 1;
            for (int64_t i = 0; i < *n_recs; ++i) if (!again[(size_t)(*recs)[i].read_idx]) all.push_back((*recs)[i]);
        }
        int64_t blob_n = 0; for (const vm_record& x : all) blob_n = std::max<int64_t>(blob_n, x.cigar_off + x.cigar_len + 1);
        std::string blob(*cigar_blob, (size_t)blob_n);
        free(*recs); free(*cigar_blob); *recs = nullptr; *cigar_blob = nullptr; *n_recs = 0;
        int64_t retries = 0;
        while (!sub.empty()) {
            bool any_short = false; for (int32_t r : sub) any_short = any_short || status[(size_t)r] == VMX_EXT_SHORT_INTERNAL;
            if (any_short) { if (c->ext_mul >= 1024) break; c->ext_mul *= 4; }       // (reads that only need the filter's later tiers keep the pools as they are)
            c->force_exact = true; c->run_pass1 = true; ++retries;
            if (getenv("VMX_DBG_POOLS")) fprintf(stderr, "[pools] extend stage: %zu read(s) run again alone (pools x%d, every tier of the divergence filter)\n", sub.size(), c->ext_mul);
            const int64_t m = (int64_t)sub.size();
            std::vector<int64_t> s_off((size_t)m + 1, 0); std::vector<vmx_preset> s_pre;
            for (int64_t j = 0; j < m; ++j) { s_off[(size_t)j + 1] = s_off[(size_t)j] + (h_roff[(size_t)sub[(size_t)j] + 1] - h_roff[(size_t)sub[(size_t)j]]); if (preset) s_pre.push_back(preset[sub[(size_t)j]]); }
            int src = 0;
            if ((src = B.side_codes.reserve((size_t)s_off[(size_t)m] + 64)) < 0 || (src = vmx_push(c, B.side_off, s_off.data(), (size_t)m + 1)) < 0 ||
                (src = vmx_push(c, B.sellist, sub.data(), (size_t)m)) < 0) { if (src == VM_ERR_OOM) break; c->ext_mul = 1; c->force_exact = false; c->run_pass1 = false; return src; }
            hipLaunchKernelGGL(k_side_codes, dim3((unsigned)std::min<int64_t>(m, (int64_t)c->num_cu * 8)), dim3(256), 0, c->stream, d_codes, d_roff, B.sellist.as<int32_t>(), B.side_off.as<int64_t>(), (int)m,
                               B.side_codes.as<uint8_t>());
            vm_record* r2 = nullptr; int64_t n2 = 0; char* b2 = nullptr; vm_batch_stats st2; std::vector<int32_t> status2((size_t)m + 1, 0);
            src = VMX_RETRY_BATCH;
            for (int attempt = 0; attempt < 6 && src == VMX_RETRY_BATCH; ++attempt)
                src = align_device_once(c, mi, prm, m, B.side_codes.as<uint8_t>(), B.side_off.as<int64_t>(), s_off, &r2, &n2, &b2, status2.data(), &st2, nullptr, preset ? s_pre.data() : nullptr);
            if (src == VMX_RETRY_BATCH) src = VM_ERR_OOM;
            if (src == VM_ERR_OOM) { free(r2); free(b2); break; }                 // no room for the larger pools: these reads are reported, the batch stands
            if (src < 0) { free(r2); free(b2); c->ext_mul = 1; c->force_exact = false; c->run_pass1 = false; return src; }
            std::vector<int32_t> still;
            for (int64_t j = 0; j < m; ++j) { status[(size_t)sub[(size_t)j]] = status2[(size_t)j]; if (status2[(size_t)j] == VMX_EXT_SHORT_INTERNAL) still.push_back(sub[(size_t)j]); }
            for (int64_t i = 0; i < n2; ++i) { vm_record x = r2[i]; x.read_idx = sub[(size_t)x.read_idx]; x.cigar_off += (int64_t)blob.size(); all.push_back(x); }
            int64_t bl = 0; for (int64_t i = 0; i < n2; ++i) bl = std::max<int64_t>(bl, r2[i].cigar_off + r2[i].cigar_len + 1);
            blob.append(b2, (size_t)bl);
            free(r2); free(b2);
            st.ms_total += st2.ms_total; for (int i = 0; i < 14; ++i) st.ms_stage[i] += st2.ms_stage[i];
            st.n_host_syncs += st2.n_host_syncs; st.n_ed_tier2 += st2.n_ed_tier2; st.n_ed_full += st2.n_ed_full; st.n_local_anchors += st2.n_local_anchors; st.n_local_general += st2.n_local_general;
            sub.swap(still);
        }
        c->ext_mul = 1; c->force_exact = false; c->run_pass1 = false;
        for (int32_t r : sub) status[(size_t)r] = VM_READ_CAPACITY;
        std::stable_sort(all.begin(), all.end(), [](const vm_record& a, const vm_record& b) { return a.read_idx < b.read_idx; });      // records stay grouped by read, in read order
        *recs = (vm_record*)malloc(sizeof(vm_record) * std::max<size_t>(all.size(), 1)); *cigar_blob = (char*)malloc(std::max<size_t>(blob.size(), 1));
        if (!*recs || !*cigar_blob) { free(*recs); free(*cigar_blob); *recs = nullptr; *cigar_blob = nullptr; set_error("out of host memory"); return VM_ERR_OOM; }
        memcpy(*recs, all.data(), sizeof(vm_record) * all.size()); memcpy(*cigar_blob, blob.data(), blob.size());
        *n_recs = (int64_t)all.size();
        // the batch's counters over the spliced result
        st.n_ext_retries = retries; st.n_records = *n_recs; st.aligned_bases = 0; st.cigar_bytes = 0; st.n_failed = 0; st.n_unmapped = 0;
        std::vector<char> has((size_t)n + 1, 0);
        for (const vm_record& x : all) { st.aligned_bases += x.q_en - x.q_st; st.cigar_bytes += x.cigar_len; has[(size_t)x.read_idx] = 1; }
        for (int64_t r = 0; r < n; ++r) { if (status[(size_t)r] != 0) st.n_failed++; else if (!has[(size_t)r]) st.n_unmapped++; }
    }
    if (status_per_read) memcpy(status_per_read, status.data(), 4 * (size_t)n);
    if (stats) *stats = st;
    return VM_OK;
}

static int align_device_once(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, const std::vector<int64_t>& h_roff,
                             vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats, vmx_seg_trace* trace, const vmx_preset* preset) {
    *recs = nullptr; *n_recs = 0; *cigar_blob = nullptr;
    vmx_batch_bufs& B = *batch_bufs(c);
    vm_index_view ix; vmx_index_view(mi, &ix);
    const int64_t total_bases = h_roff[n];
    vm_batch_stats st; memset(&st, 0, sizeof st);
    st.n_reads = n; st.read_bases = total_bases;
    hipEvent_t* ev = c->ev; int nev = 0;
    c->n_syncs = 0; c->n_bandfall = 0; c->kev_set = 0; c->sync_wait_ns = 0; download_wait_ns() = 0;
    c->mb.pend.clear(); c->mb.dn_used = 0; c->mb.big_used = 0;      // (fetches a failed call left behind must not be delivered into its dead buffers)
    c->call_t0_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));
    if (n == 0) { *recs = (vm_record*)malloc(sizeof(vm_record)); *cigar_blob = (char*)malloc(1); if (stats) *stats = st; return VM_OK; }

    std::vector<int64_t> h_aoff((size_t)n + 1, 0);
    std::vector<int32_t> asm_override;                           // -mode asm: per-contig status set by the tie-break step
    const int rmode = prm->mode == VM_MODE_R ? 1 : (prm->mode == VM_MODE_ASM ? 2 : 0);      // the GC kernels' variant: H / L / S, R, the asm fork
    double* d_gscore = nullptr; int32_t* d_mapq = nullptr; int32_t* d_np = nullptr;
    vmx_local_bufs& L = *vmx_ctx_local_bufs(c);
    // Round 6: after a context's first batch the LOCAL stage runs without a host wait (vmx_local_stage's fast form) and the extend stage's per-read pool geometry is made
    // on the device (k_ext_geometry) inside pools sized from the context's history — one wait less per batch. A read whose slice does not fit, whose anchors overflow
    // their slot or that the banded seeding kernel hands back is run again alone on the waiting path (side batches, below). VMX_LOCAL_SYNC=1: the waiting path always.
    int tmask = 0; int64_t tdiv = 1;                            // test hook: VMX_TEST_EXT_POOL=<mask>:<div> (pools of exact size divided: the waiting path)
    if (const char* e = getenv("VMX_TEST_EXT_POOL")) { long long m_ = 0, d_ = 1; if (sscanf(e, "%lld:%lld", &m_, &d_) == 2 && d_ >= 1) { tmask = (int)m_; tdiv = d_; } }
    static const bool local_sync_env = getenv("VMX_LOCAL_SYNC") != nullptr;
    const bool fast_local = !preset && !trace && !c->force_exact && !local_sync_env && vmx_chain_rows_on() && rmode != 2 && c->geo_valid && c->ext_mul == 1 && tmask == 0;
    // the stages in front of the extension: seed, global chain, local chain
    auto front = [&]() -> int {
    // ---------------- S1 seed
    std::vector<int64_t> h_koff, h_nhits;
    int64_t* d_seed_rows = nullptr;
    static const bool seed_arena = [] { const char* e = getenv("VMX_SEED_ARENA"); return !(e && atoi(e) == 0); }();
    VMX_TRY(vmx_seed_stage(c, mi, prm->check_num, prm->mid_occ, n, d_codes, d_roff, total_bases, B.seed, h_koff, h_nhits, seed_arena ? &B.tb : nullptr, &d_seed_rows));
    for (int64_t r = 0; r < n; ++r) st.n_hits += h_nhits[r];
    st.n_minimizers = c->last_n_minimizers;
    // compact anchors: aoff = scan(n_anchors)
    VMX_TRY(B.nanc64.reserve(8 * (size_t)(n + 2))); VMX_TRY(B.aoff.reserve(8 * (size_t)(n + 2)));
    LAUNCH1D(k_i32_to_i64, n, B.seed[11].as<int32_t>(), B.nanc64.as<int64_t>(), n);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, B.nanc64.as<int64_t>(), B.aoff.as<int64_t>(), n, 0);
    VMX_TRY(vmx_fetch(c, h_aoff.data(), B.aoff.p, (size_t)n + 1));
    VMX_HIP(vmx_stream_sync(c));
    const int64_t tot = h_aoff[n];
    st.n_anchors = tot;
    VMX_TRY(B.rows.reserve(32 * (size_t)(tot + 1)));
    hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)std::min<int64_t>(n, (int64_t)c->num_cu * 8)), dim3(128), 0, c->stream, d_seed_rows, B.seed[7].as<int64_t>(),
                       B.aoff.as<int64_t>(), (int)n, B.rows.as<int64_t>());
    VMX_TRY(B.lens.reserve(8 * (size_t)(n + 1)));
    LAUNCH1D(k_readlens, n, d_roff, B.lens.as<int64_t>(), n);
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));

    // ---------------- S2 + G1/G2 global chain
    std::vector<int64_t> koff((size_t)n + 1), soff((size_t)n + 1);
    int64_t kt = 0, stt = 0;
    for (int64_t r = 0; r < n; ++r) {
        int64_t m = h_aoff[r + 1] - h_aoff[r]; int64_t N = 1; while (N < m) N <<= 1;
        koff[r] = kt; kt += N; soff[r] = stt; stt += (vmx_select_scratch_bytes(m) + 15) & ~(int64_t)15;
    }
    koff[n] = kt; soff[n] = stt;
    VMX_TRY(B.keys.reserve(8 * (size_t)(kt + 1))); VMX_TRY(vmx_push(c, B.koff, koff.data(), (size_t)n + 1));
    VMX_TRY(B.sorted.reserve(sizeof(vmx_anchor) * (size_t)(tot + 1))); VMX_TRY(B.flip.reserve(4 * (size_t)(n + 1)));
    hipLaunchKernelGGL(k_flip_sort, dim3((unsigned)std::min<int64_t>(n, (int64_t)c->num_cu * 8)), dim3(256), 0, c->stream, B.rows.as<int64_t>(), B.aoff.as<int64_t>(), B.lens.as<int64_t>(),
                       (int)n, B.keys.as<uint64_t>(), B.koff.as<int64_t>(), B.sorted.as<vmx_anchor>(), B.flip.as<int32_t>());
    VMX_TRY(B.S.reserve(8 * (size_t)(tot + 1))); VMX_TRY(B.P.reserve(4 * (size_t)(tot + 1))); VMX_TRY(B.SA.reserve(4 * (size_t)(tot + 1)));
    VMX_TRY(B.cov.reserve((size_t)tot + 16)); VMX_TRY(B.gmax.reserve(8 * (size_t)(n + 1))); VMX_TRY(B.opc.reserve(8 * (size_t)(n + 1)));
    const HostTables& T = host_tables();
    {
        if (prm->global_maxdiff > 62) { set_error("global_maxdiff > 62 unsupported"); return VM_ERR_UNSUPPORTED; }
        std::vector<double> gap(64, 0.0);
        for (int g = 1; g <= prm->global_maxdiff; ++g) gap[g] = (0.01 * ix.k * g + 0.5 * T.log2int[g]);
        VMX_TRY(vmx_push(c, B.gap, gap.data(), 64));
    }
    if (rmode == 1) { VMX_TRY(B.fp.reserve(8 * (size_t)(tot + 1))); VMX_TRY(B.pp.reserve(8 * (size_t)(tot + 1))); }   // fixed_penatly / pre_penatly of mode R's chain
    // LDS buckets by anchor count (25 B per anchor), reads longest-first inside a bucket, one workgroup per read (see vmx_local_stage)
    constexpr int NB = 10;
    const int caps[NB] = {384, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192, 13056};
    std::vector<int32_t> lists[NB + 1];
    std::vector<int64_t> asm_long;
    static const int gc_lds_env = [] { const char* e = getenv("VMX_GC_LDS_MAX"); return e ? atoi(e) : -1; }();     // see vmx_stage_local.hip
    const int gc_lds_max = gc_lds_env >= 0 ? gc_lds_env : VMX_CHAIN_LDS_MAX_SHARED;
    const bool rows_kernel = vmx_chain_rows_on() && rmode != 2;          // four reads per wavefront (k_chain_rows.hip); -mode asm keeps the one-wave kernel
    for (int64_t r = 0; r < n; ++r) {
        int64_t m = h_aoff[r + 1] - h_aoff[r]; const int64_t L = h_roff[r + 1] - h_roff[r];
        if (m <= 2) continue;                                             // :23986 unmapped
        if (prm->mode == VM_MODE_ASM && L >= 500000) { asm_long.push_back(r); continue; }      // mammap_asm.py:23205: the linked path, not built on the device
        if ((double)m / (double)L > 5.0) continue;                        // fast_enable (:23570): gmax stays -1 -> k_chain_global_fast below
        int bk = NB; for (int q = 0; q < NB; ++q) if (m <= caps[q] && caps[q] <= gc_lds_max) { bk = q; break; }
        if (rows_kernel) bk = NB;                                         // k_chain_global_rows: one list, nothing in LDS
        lists[bk].push_back((int32_t)r);
    }
    {
        std::vector<int32_t> rl; int64_t rl_off[NB + 2];
        for (int q = 0; q <= NB; ++q) {
            std::stable_sort(lists[q].begin(), lists[q].end(), [&](int32_t a, int32_t b) { return h_aoff[a + 1] - h_aoff[a] > h_aoff[b + 1] - h_aoff[b]; });
            rl_off[q] = (int64_t)rl.size(); rl.insert(rl.end(), lists[q].begin(), lists[q].end());
        }
        VMX_TRY(vmx_push(c, B.rl, rl.data(), rl.size()));
        VMX_HIP(hipMemsetAsync(B.gmax.p, 0xff, 8 * (size_t)n, c->stream));
        { static const int64_t kUnsupported = -4; for (int64_t r : asm_long) VMX_HIP(hipMemcpyAsync(B.gmax.as<int64_t>() + r, &kUnsupported, 8, hipMemcpyHostToDevice, c->stream)); }
#ifndef VMX_EMU
        VMX_HIP(hipFuncSetAttribute((const void*)k_chain_global, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)caps[NB - 1] * VMX_GC_BYTES_PER_ANCHOR + 64)));
#endif
        vmx_fork fk(c);                                               // the LDS buckets are independent: run them side by side
        for (int q = NB; q >= 0; --q) {                               // slowest (largest reads) first
            int cnt = (int)lists[q].size(); if (!cnt) continue;
            int cap = q < NB ? caps[q] : 0; size_t shmem = (size_t)cap * VMX_GC_BYTES_PER_ANCHOR + 64;
            if (rows_kernel) {
                hipLaunchKernelGGL((vmx_chain_rows_win3() ? k_chain_global_rows_w3 : k_chain_global_rows), dim3((unsigned)((cnt + 3) / 4)), dim3(64), 0, fk.next(), B.sorted.as<vmx_anchor>(), B.aoff.as<int64_t>(),
                                   B.rl.as<int32_t>() + rl_off[q], cnt, c->tables, B.gap.as<double>(), prm->global_skipcost, prm->global_maxdiff, 1000, B.S.as<double>(),
                                   B.P.as<int32_t>(), B.SA.as<int32_t>(), B.cov.as<uint8_t>(), B.gmax.as<int64_t>(), B.opc.as<int64_t>(), rmode, B.fp.as<double>(), B.pp.as<double>(),
                                   vmx_chain_dbg());
                continue;
            }
            hipLaunchKernelGGL(k_chain_global, dim3((unsigned)cnt), dim3(64), shmem, fk.next(), B.sorted.as<vmx_anchor>(), B.aoff.as<int64_t>(),
                               B.rl.as<int32_t>() + rl_off[q], cnt, cap, c->tables, B.gap.as<double>(), prm->global_skipcost, prm->global_maxdiff, 1000, B.S.as<double>(),
                               B.P.as<int32_t>(), B.SA.as<int32_t>(), B.cov.as<uint8_t>(), B.gmax.as<int64_t>(), B.opc.as<int64_t>(), rmode, B.fp.as<double>(), B.pp.as<double>());
        }
        fk.join();
        vmx_chain_dbg_report(c->stream);
        // G3: reads left at gmax = -1 (more than 5 anchors per base, :23570, or GC-exact's opcount bail-out, :24914) take GC-fast.
        // One wave per read; reads that do not need it return at once.
        VMX_TRY(B.si.reserve(4 * (size_t)(tot + 1))); VMX_TRY(B.tg.reserve(8 * (size_t)(tot + 1))); VMX_TRY(B.cntp.reserve(4 * (size_t)(total_bases + 50 * n + 64)));
        hipLaunchKernelGGL(k_chain_global_fast, dim3((unsigned)n), dim3(64), 0, c->stream, B.sorted.as<vmx_anchor>(), B.aoff.as<int64_t>(), (int)n, d_roff, c->tables,
                           B.gap.as<double>(), prm->global_skipcost, prm->global_maxdiff, 1000, B.S.as<double>(), B.P.as<int32_t>(), B.SA.as<int32_t>(), B.cov.as<uint8_t>(),
                           B.si.as<int32_t>(), B.tg.as<int64_t>(), B.cntp.as<int32_t>(), B.gmax.as<int64_t>(), (int32_t*)nullptr, rmode, B.fp.as<double>(), B.pp.as<double>());
    }
    VMX_TRY(B.scr.reserve((size_t)stt + 64)); VMX_TRY(vmx_push(c, B.soff, soff.data(), (size_t)n + 1));
    VMX_TRY(B.res.reserve(16 * (size_t)(n + 2) + 64));
    d_gscore = B.res.as<double>(); d_mapq = (int32_t*)(d_gscore + n + 1); d_np = d_mapq + n + 1;
    VMX_TRY(B.plen.reserve(4 * (size_t)(tot + 1))); VMX_TRY(B.prow.reserve(sizeof(vmx_anchor) * (size_t)(tot + 1)));
    VMX_TRY(vmx_launch_chain_select(c, n, h_aoff.data(), B.sellist, B.sorted.as<vmx_anchor>(), B.aoff.as<int64_t>(), B.lens.as<int64_t>(), B.S.as<double>(), B.P.as<int32_t>(), B.SA.as<int32_t>(),
                                    B.gmax.as<int64_t>(), B.flip.as<int32_t>(), prm->mode, B.scr.as<char>(), B.soff.as<int64_t>(), d_mapq, d_gscore, d_np, B.plen.as<int32_t>(), B.prow.as<vmx_anchor>()));
    if (prm->mode == VM_MODE_ASM)        // decode_hit's edlib tie-break among equal chains (mammap_asm.py:21302-21326): marked contigs are settled here
        VMX_TRY(vmx_asm_resolve_ties(c, mi, n, h_roff, h_aoff, d_codes, B.sorted.as<vmx_anchor>(), B.S.as<double>(), B.P.as<int32_t>(), B.SA.as<int32_t>(), B.gmax.as<int64_t>(),
                                     B.flip.as<int32_t>(), d_mapq, d_gscore, d_np, B.plen.as<int32_t>(), B.prow.as<vmx_anchor>(), asm_override));
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));

    // ---------------- orient + L1-L4 local stage
    VMX_TRY(B.ocodes.reserve((size_t)total_bases + 64));
    hipLaunchKernelGGL(k_orient, dim3((unsigned)std::min<int64_t>(n, (int64_t)c->num_cu * 8)), dim3(256), 0, c->stream, d_codes, d_roff, d_gscore, (int)n, B.ocodes.as<uint8_t>());
    VMX_TRY(vmx_local_stage(c, ix, prm, n, B.ocodes.as<uint8_t>(), d_roff, h_roff, B.prow.as<vmx_anchor>(), B.plen.as<int32_t>(), d_np, B.aoff.as<int64_t>(), h_aoff, d_gscore, L, fast_local));
    if (!fast_local) for (int64_t r = 0; r < n; ++r) st.n_local_anchors += L.h_la_cnt[r];
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));

    return 0;
    };
    // -mode asm, contig of 500 kb and more (vmx_asm.hip): the chain comes from the linked DPs; the extend stage below runs on it as
    // ass_extend_func does (mammap_asm.py:23423-23460): forward orientation, MAPQ 60
    auto front_preset = [&]() -> int {
        int64_t tot_chain = 0;
        for (int64_t r = 0; r < n; ++r) { if (preset[r].len < 2) { set_error("preset chain: every contig needs a chain of two anchors or more"); return VM_ERR_ARG; } tot_chain += preset[r].len; }
        VMX_TRY(B.ocodes.reserve((size_t)total_bases + 64));
        VMX_HIP(hipMemcpyAsync(B.ocodes.p, d_codes, (size_t)total_bases, hipMemcpyDeviceToDevice, c->stream));
        VMX_TRY(B.res.reserve(16 * (size_t)(n + 2) + 64));
        d_gscore = B.res.as<double>(); d_mapq = (int32_t*)(d_gscore + n + 1); d_np = d_mapq + n + 1;
        std::vector<double> gs((size_t)n, 1.0); std::vector<int32_t> mq((size_t)n, 60), one((size_t)n, 1), zero((size_t)n, 0), len32((size_t)n);
        std::vector<vmx_anchor> chains; chains.reserve((size_t)tot_chain);
        L.h_la_off.assign((size_t)n + 1, 0); L.h_la_cnt.assign((size_t)n, 0);
        for (int64_t r = 0; r < n; ++r) {
            len32[(size_t)r] = (int32_t)preset[r].len; L.h_la_cnt[(size_t)r] = (int32_t)preset[r].len; L.h_la_off[(size_t)r + 1] = L.h_la_off[(size_t)r] + preset[r].len;
            chains.insert(chains.end(), preset[r].chain_desc, preset[r].chain_desc + preset[r].len);
            h_aoff[(size_t)r + 1] = h_aoff[(size_t)r] + 3;
        }
        VMX_HIP(hipMemcpyAsync(d_gscore, gs.data(), 8 * (size_t)n, hipMemcpyHostToDevice, c->stream)); VMX_HIP(hipMemcpyAsync(d_mapq, mq.data(), 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        VMX_HIP(hipMemcpyAsync(d_np, one.data(), 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        VMX_TRY(B.gmax.reserve(8 * (size_t)(n + 1))); VMX_HIP(hipMemsetAsync(B.gmax.p, 0, 8 * (size_t)n, c->stream));
        VMX_TRY(vmx_push(c, L.la_off, L.h_la_off.data(), (size_t)n + 1));
        VMX_TRY(vmx_push(c, L.chain, chains.data(), chains.size()));
        VMX_TRY(vmx_push(c, L.chain_len, len32.data(), (size_t)n));
        VMX_TRY(vmx_push(c, L.status, zero.data(), (size_t)n));
        VMX_HIP(vmx_stream_sync(c));                              // the host vectors above are on their way
        for (int e = 0; e < 3; ++e) VMX_HIP(hipEventRecord(ev[nev++], c->stream));
        return 0;
    };
    { const int rcf = preset ? front_preset() : front(); if (rcf < 0) return rcf; }

    // ---------------- E1-E6 extend stage
    // per-read pool geometry from the local anchor count (the chain is never longer than that)
    std::vector<int64_t> coff3((size_t)n + 1), soff2((size_t)n + 1), bloboff((size_t)n + 1);
    auto pool = [&](int bit, int64_t x) -> int64_t { const int64_t y = x * c->ext_mul; return (tmask & bit) ? std::max<int64_t>(y / tdiv, 1) : y; };
    int64_t cA = 0, cS = 0, cB = 0, cS_full = 0;
    if (!fast_local) {
        for (int64_t r = 0; r < n; ++r) {
            const int64_t cl = L.h_la_cnt[r], len = h_roff[r + 1] - h_roff[r];
            coff3[r + 1] = coff3[r] + (cl > 0 ? pool(1, 3 * cl + 8) : 0); soff2[r + 1] = soff2[r] + (cl > 0 ? pool(2, cl + 2) : 0);
            bloboff[r + 1] = bloboff[r] + (cl > 0 ? ((pool(4, 3 * len + 64 * (cl / 2 + 2) + 56) + 7) & ~(int64_t)7) : 0);
            if (cl > 0) cS_full += cl + 2;
        }
        cA = coff3[n]; cS = soff2[n]; cB = bloboff[n];
        if (!preset && c->ext_mul == 1 && tmask == 0) {      // the pools now hold at least this much: what the batches that do not ask may use
            c->geo_cap[0] = std::max<long long>(c->geo_cap[0], cA); c->geo_cap[1] = std::max<long long>(c->geo_cap[1], cS); c->geo_cap[2] = std::max<long long>(c->geo_cap[2], cB);
            c->geo_cap[3] = std::max<long long>(c->geo_cap[3], cS_full);
            c->geo_valid = true;
        }
    } else {
        // exactly what the pools already hold (no pool grows on this path: a growth is a hipMalloc of gigabytes with the GPU idle); a batch that needs more has its last
        // reads cut (k_ext_geometry) and run again alone, and this context's next batch takes the waiting path once, which sizes the pools for it
        cA = c->geo_cap[0]; cS = c->geo_cap[1]; cB = c->geo_cap[2]; cS_full = c->geo_cap[3];
    }
    VMX_TRY(B.er.reserve(sizeof(vmx_ext_read) * (size_t)(n + 1)));
    if (!fast_local) { const vmx_push_req rq[3] = {{&B.coff3, coff3.data(), 8 * ((size_t)n + 1)}, {&B.soff2, soff2.data(), 8 * ((size_t)n + 1)}, {&B.bloboff, bloboff.data(), 8 * ((size_t)n + 1)}}; VMX_TRY(vmx_push_many(c, rq, 3)); }
    else {
        VMX_TRY(B.coff3.reserve(8 * ((size_t)n + 2))); VMX_TRY(B.soff2.reserve(8 * ((size_t)n + 2))); VMX_TRY(B.bloboff.reserve(8 * ((size_t)n + 2))); VMX_TRY(B.geotot.reserve(64));
        hipLaunchKernelGGL(k_ext_geometry, dim3(1), dim3(1024), 0, c->stream, L.la_cnt.as<int32_t>(), d_roff, (int)n, c->ext_mul, tmask, (long long)tdiv, (long long)cA, (long long)cS, (long long)cB,
                           B.coff3.as<int64_t>(), B.soff2.as<int64_t>(), B.bloboff.as<int64_t>(), B.geotot.as<int64_t>());
    }
    VMX_TRY(B.segA.reserve(sizeof(vmx_anchor) * (size_t)(cA + 1))); VMX_TRY(B.segA_s.reserve(sizeof(vmx_anchor) * (size_t)(cA + 1)));
    VMX_TRY(B.st.reserve(4 * (size_t)(cS + 1))); VMX_TRY(B.en.reserve(4 * (size_t)(cS + 1))); VMX_TRY(B.st_s.reserve(4 * (size_t)(cS + 1))); VMX_TRY(B.en_s.reserve(4 * (size_t)(cS + 1)));
    VMX_TRY(B.segprob.reserve(4 * (size_t)(cS + 1))); VMX_TRY(B.dup.reserve(4 * (size_t)(cS + 1)));
    const int64_t round_cap = pool(8, cS_full + 16);         // one problem per anchor at most in any round
    const int64_t pool_cap = pool(16, 6 * total_bases + (1 << 20));
    int64_t Lmax_b = 1; for (int64_t r = 0; r < n; ++r) Lmax_b = std::max(Lmax_b, h_roff[r + 1] - h_roff[r]);
    const int64_t carry_stride = preset ? 32768 : 2 * Lmax_b + 32768;      // (ass_extend_func has no divergence filter: a preset chain never reaches the exact kernel, and a ring per
                                                                           //  workgroup sized by a 100 Mb contig asked for 617 GB)
    const int64_t ed_wgs = std::min<int64_t>(c->num_cu, 64);   // workgroups of the exact kernel (x1 long, x2 short patterns): the last tier of the filter, hardly ever
                                                               // reached (0 problems per step on the bench workload), so its carry rings are kept small (0.8 instead of 3.2 GB)        // exact edit-distance kernel: one carry ring per workgroup (1 + 2 per CU), longest text it takes
    VMX_TRY(B.desc[0].reserve(sizeof(vmx_pair_desc) * (size_t)round_cap)); VMX_TRY(B.desc[1].reserve(sizeof(vmx_pair_desc) * (size_t)round_cap));
    VMX_TRY(B.rcount.reserve(256)); VMX_HIP(hipMemsetAsync(B.rcount.p, 0, 256, c->stream)); c->rc_next = 0; c->rc_cur = B.rcount.as<int32_t>(); VMX_TRY(B.oflow.reserve(64)); VMX_TRY(B.probread.reserve(4 * (size_t)round_cap));
    VMX_TRY(B.tl.reserve(8 * (size_t)(round_cap + 1))); VMX_TRY(B.ql.reserve(8 * (size_t)(round_cap + 1))); VMX_TRY(B.toff.reserve(8 * (size_t)(round_cap + 1))); VMX_TRY(B.qoff.reserve(8 * (size_t)(round_cap + 1)));
    VMX_TRY(B.tpool.reserve((size_t)pool_cap + 64)); VMX_TRY(B.qpool.reserve((size_t)pool_cap + 64));
    VMX_TRY(B.edout.reserve(8 * (size_t)(round_cap + 1))); VMX_TRY(B.carry.reserve((size_t)VMX_ED_WAVES * (size_t)carry_stride * (size_t)ed_wgs * 3 + 64)); VMX_TRY(B.ext3.reserve(12 * (size_t)(round_cap + 1)));
    VMX_TRY(B.rec.reserve(sizeof(vm_record) * (size_t)(cS + 1))); VMX_TRY(B.blob.reserve((size_t)cB + 64)); VMX_TRY(B.reccoff.reserve(8 * (size_t)(cS + 1)));
    VMX_TRY(B.recclen.reserve(4 * (size_t)(cS + 1))); VMX_TRY(B.dupd.reserve((size_t)cB + 64));
    VMX_HIP(hipMemsetAsync(B.oflow.p, 0, 4, c->stream));
    VMX_TRY(B.statblk.reserve(2048)); VMX_HIP(hipMemsetAsync(B.statblk.p, 0, 2048, c->stream));     // [0..7] i32 round counts | i64 [32..45] final scalars | i64 [64..121] chunk plan of pass 0 | i64 [128..185] of pass 1
    const size_t gfctl_bytes = (size_t)2 * (VMX_MAX_CHUNKS + 1) * (32 + 544) * 4;
    VMX_TRY(B.gfctl.reserve(gfctl_bytes)); VMX_HIP(hipMemsetAsync(B.gfctl.p, 0, gfctl_bytes, c->stream));      // the gap fill's per-chunk control blocks and size-order scratch (below), cleared once
    vmx_ext_args A; memset(&A, 0, sizeof A);
    A.n_reads = (int)n; A.nseq = ix.nseq; A.local_maxdiff = preset ? 50 : prm->local_maxdiff /* ass_extend_func: large_cost 50, mammap_asm.py:23426 */; A.asm_long = preset ? 1 : 0; A.nodiscard = prm->nodiscard; A.hardclip = prm->hardclip; A.redo_only = 0; A.mode = prm->mode; A.maxdivergence = prm->maxdivergence;
    A.ocodes = B.ocodes.as<uint8_t>(); A.roff = d_roff; A.ref = ix.codes; A.coff = ix.coff;
    A.chain = L.chain.as<vmx_anchor>(); A.chain_len = L.chain_len.as<int32_t>(); A.la_off = L.la_off.as<int64_t>(); A.lstatus = L.status.as<int32_t>();
    A.gscore = d_gscore; A.mapq = d_mapq; A.er = B.er.as<vmx_ext_read>(); A.coff3 = B.coff3.as<int64_t>(); A.soff = B.soff2.as<int64_t>();
    A.segA = B.segA.as<vmx_anchor>(); A.st = B.st.as<int32_t>(); A.en = B.en.as<int32_t>(); A.segA_snap = B.segA_s.as<vmx_anchor>(); A.st_snap = B.st_s.as<int32_t>(); A.en_snap = B.en_s.as<int32_t>();
    A.seg_prob = B.segprob.as<int32_t>(); A.dup = B.dup.as<int32_t>(); A.round_count = B.rcount.as<int32_t>(); A.round_cap = round_cap; A.overflow = B.oflow.as<int32_t>(); A.prob_read = B.probread.as<int32_t>();
    A.ed_out = B.edout.as<int64_t>(); A.ext_te = B.ext3.as<int32_t>(); A.ext_qe = B.ext3.as<int32_t>() + round_cap;
    A.rec = B.rec.as<vm_record>(); A.rec_blob = B.blob.as<char>(); A.blob_off = B.bloboff.as<int64_t>(); A.rec_coff = B.reccoff.as<int64_t>(); A.rec_clen = B.recclen.as<int32_t>();
    A.dup_d = B.dupd.as<double>();
    const unsigned gridR = (unsigned)((n + 63) / 64);
    // k_ext_phase / k_ext_records: one lane in `spread` works. Alone, spreading the reads over more waves shortens the batch (a batch of
    // 40-100 kb reads, 1 / 4 / 16 / 64 lanes per read: phases 19.7 / 16.5 / 13.9 / 14.4 ms, records 8.7 / 7.6 / 8.6 / 15.6 ms; batch 130 -> 123 ms);
    // with three batches in flight it does not raise the throughput (34.3 vs 33.7 ms per step), so a shared context keeps one lane per read
    static const int ext_env = [] { const char* e = getenv("VMX_EXT_SPREAD"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 0; }();
    static const int rec_env = [] { const char* e = getenv("VMX_REC_SPREAD"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 0; }();
    const int ext_spread = ext_env ? ext_env : (c->inflight >= 2 ? 1 : 16), rec_spread = rec_env ? rec_env : (c->inflight >= 2 ? 1 : 4);
    const unsigned gridX = (unsigned)((n * ext_spread + 63) / 64), gridXR = (unsigned)((n * rec_spread + 63) / 64);
    static const bool ext_wave = [] { const char* e = getenv("VMX_EXT_WAVE"); return !(e && atoi(e) == 0); }();
    int cur = 0;
    auto phase = [&](int ph) { A.desc = B.desc[cur].as<vmx_pair_desc>(); A.desc_prev = B.desc[cur ^ 1].as<vmx_pair_desc>();
                               c->rc_cur = B.rcount.as<int32_t>() + (c->rc_next++ & 63); A.round_count = c->rc_cur;      // a fresh, already cleared counter per phase (one clearing per batch)
                               // the phases that walk every anchor (0 rebuild, 3 snapshot copy, 5 checkpoints) run one wavefront per read; VMX_EXT_WAVE=0: the one-lane form
                               const bool wv = ext_wave && (ph == 0 || ph == 3 || ph == 5);
                               A.spread = wv ? 64 : ext_spread; hipLaunchKernelGGL(k_ext_phase, dim3(wv ? (unsigned)n : gridX), dim3(64), 0, c->stream, A, ph);
                               if (trace && trace->stage == ph && !A.redo_only && trace->off.empty()) {
                                   std::vector<vmx_ext_read> er((size_t)n); std::vector<vmx_anchor> sa((size_t)cA + 1); std::vector<int32_t> st_((size_t)cS + 1), en_((size_t)cS + 1);
                                   (void)hipMemcpyAsync(er.data(), B.er.p, sizeof(vmx_ext_read) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
                                   (void)hipMemcpyAsync(sa.data(), B.segA.p, sizeof(vmx_anchor) * (size_t)cA, hipMemcpyDeviceToHost, c->stream);
                                   (void)hipMemcpyAsync(st_.data(), B.st.p, 4 * (size_t)cS, hipMemcpyDeviceToHost, c->stream);
                                   (void)hipMemcpyAsync(en_.data(), B.en.p, 4 * (size_t)cS, hipMemcpyDeviceToHost, c->stream);
                                   (void)vmx_stream_sync(c);
                                   trace->off.assign(1, 0);
                                   for (int64_t r = 0; r < n; ++r) {
                                       if (er[(size_t)r].active && er[(size_t)r].status == 0)
                                           for (int sg = 0; sg < er[(size_t)r].nseg; ++sg)
                                               for (int t = st_[(size_t)(soff2[r] + sg)]; t < en_[(size_t)(soff2[r] + sg)]; ++t) {
                                                   const vmx_anchor& a = sa[(size_t)(coff3[r] + t)];
                                                   const int64_t row[5] = {sg, a.q, a.r, a.s, a.l}; trace->rows.insert(trace->rows.end(), row, row + 5);
                                               }
                                       trace->off.push_back((int64_t)trace->rows.size() / 5);
                                   }
                               } };
    int ext_rounds = 0;
    auto ext_round = [&](int redo_only) -> int {   // x-drop extension of the problems of the current round (their count stays on the device)
        int rc = ext_gather_round(c, B, ix, n, B.ocodes.as<uint8_t>(), d_roff, cur, redo_only, round_cap, pool_cap, 1 + ext_rounds++, false, 0);
        if (rc < 0) return rc;
        hipLaunchKernelGGL(k_extend, dim3((unsigned)((int64_t)c->num_cu * 16)), dim3(64), 0, c->stream, B.tpool.as<uint8_t>(), B.toff.as<int64_t>(), B.qpool.as<uint8_t>(),
                           B.qoff.as<int64_t>(), 0, 2, -4, 4, 4, 100, 50, B.ext3.as<int32_t>(), B.ext3.as<int32_t>() + round_cap, B.ext3.as<int32_t>() + 2 * round_cap, c->rc_cur);
        cur ^= 1;
        return 0;
    };
    std::vector<int64_t> dp_tot(5, 0);
    // ---- gap fill of one pass (E5). Round 6: ONE host wait per batch in here (the chunk plan of pass 0), none per chunk and none in pass 1.
    //  * every chunk has its own control block (queue range / counters / redo list length / redo bytes: 32 ints) and its own zeroed scratch for the size ordering
    //    in B.gfctl, cleared once per batch: nothing is cleared between chunks, and the host reads all of them with the batch's final results;
    //  * the full-matrix traceback space of the problems whose band was not proven (the SECOND launch's pool) is sized from the context's history (largest chunk
    //    seen x 1.3, 512 MB to begin with) instead of by a read-back between the two launches; the band pass checks every allocation against the pool and a problem
    //    that does not fit is emptied (tl = ql = 0: no fill, empty CIGAR) — the host sees the need at the end, grows the pool and runs the batch again;
    //  * pass 1 (the nofilter re-run of a read whose segment filter removed something and whose CIGARs carry paired indels, :24079-24080) is not part of a batch at all:
    //    hardly any read asks for it (none of the 150 golden reads, a handful per 100 k synthetic reads), and its eleven launches and one host wait were paid by every batch.
    //    A read that asks for it (E.redo) is run again alone by align_device with both passes (c->run_pass1), like the reads that need the later tiers of the divergence filter.
    constexpr int GF_SLOT = 32 + 544;                                  // ints per chunk slot of B.gfctl: control block, then the size-order scratch (513 used)
    auto gf_slot = [&](int pass, int q) -> int32_t* { return B.gfctl.as<int32_t>() + (size_t)(pass * (VMX_MAX_CHUNKS + 1) + q) * GF_SLOT; };
    // The band-width rule (vmx_ad_ns: the narrowest band whose margin covers pct % of the problem) only decides which problems are TRIED in a band and how wide — the
    // proof decides what is kept, so the records do not depend on it. Round 6: pct follows the reads instead of the mode alone. Each context starts at the mode's
    // default (90; mode L 40) and steps down (to 20 at the least) by 20 / 10 while fewer than 1 / 2.5 % of a batch's problems are tried in a band and not proven (they are
    // filled again in full: ~4x a band attempt), back up when more than 3 % are, and then holds that floor for 256 batches. HiFi-shape reads settle at 20 (a 270-base
    // problem runs on ONE diagonal pair per lane: margin 104 against ~8 points of errors), ONT reads at 80-90 (70 fails 6 %, 55 fails 29 %):
    // `profiles/r06_q_band_width_rule_sweep.txt`. VMX_AD_PCT / VMX_AD_PCT_MIN pin the rule (tuning runs).
    static std::mutex ad_m; static struct { int pct = 0, floor = 20, hold = 0; } ad_tab[16][8];      // per device and read mode, shared by the contexts of the process
    auto& adr = ad_tab[c->device & 15][prm->mode & 7];
    int ad_cur; { std::lock_guard<std::mutex> g(ad_m); if (adr.pct == 0) adr.pct = vmx_ad_pct_env(prm->mode) & 0xffff; ad_cur = adr.pct; }
    const bool ad_pinned = getenv("VMX_AD_PCT") != nullptr || getenv("VMX_AD_PCT_MIN") != nullptr;
    const int ad_pct = ad_pinned ? vmx_ad_pct_env(prm->mode) : (ad_cur | (std::max(20, ad_cur - 25) << 16));
    int fill_waves = 16;                                              // waves per CU of the fill kernel (tuning knob: VMX_FILL_WAVES)
    if (const char* e = getenv("VMX_FILL_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 32) fill_waves = v; }
    static const int tr_spread = [] { const char* e = getenv("VMX_TRACE_SPREAD"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 64 ? v : 1; }();
    static const int64_t tb_chunk = [] { const char* e = getenv("VMX_TB_CHUNK_GB"); const double v = e ? atof(e) : 0.0; return v > 0.5 ? (int64_t)(v * (double)(1 << 30)) : (int64_t)VMX_TB_CHUNK; }();     // tuning knob
    int gf_chunks[2] = {0, 0}; int64_t gf_redo_cap = 0;
    // one chunk: problems [p0, p0 + pn) — pn an upper bound when n_ptr names the count on the device
    auto gf_chunk = [&](int pass, int q, int p0, int64_t pn, const int32_t* n_ptr, int64_t tb_off0) -> int {
        int32_t* ctl = gf_slot(pass, q); int32_t* d_range = ctl; int32_t* d_cnt = ctl + 4; int32_t* d_redo_cnt = ctl + 12;
        unsigned long long* d_redo_bytes = (unsigned long long*)(ctl + 16); int32_t* scratch = ctl + 32;
        int32_t* d_redo_list = B.order.as<int32_t>() + round_cap + 32;
        uint8_t* tb_base = B.tb.as<uint8_t>() - tb_off0;           // the problems' absolute traceback offsets index a buffer that holds this chunk only
        hipEvent_t* ke = q < 8 ? c->gev + (pass ? 24 : 0) + 3 * q : nullptr;      // HIP events around the dominant kernel, on the stream it runs on
        const unsigned Gs = (unsigned)std::max<int64_t>(1, std::min<int64_t>((pn + 1023) / 1024, (int64_t)c->num_cu * 2));
        hipLaunchKernelGGL(k_size_hist, dim3(Gs), dim3(256), 0, c->stream, B.dpsz[0].as<int64_t>() + p0, n_ptr, (int64_t)((1LL << 42) - 1), scratch);        // (queue keys: k_round.hip)
        hipLaunchKernelGGL(k_size_scatter, dim3(Gs), dim3(256), 0, c->stream, B.dpsz[0].as<int64_t>() + p0, n_ptr, (const int32_t*)scratch, scratch + 257, B.order.as<int32_t>(), d_range, d_cnt);
        if (ke) (void)hipEventRecord(ke[0], c->stream);
        {
            vmx_lowprio lp(c);                                    // the long launch at the lowest dispatch priority (vmx_host.h)
            hipLaunchKernelGGL(k_gapfill_fill_ns, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(pn, (int64_t)c->num_cu * fill_waves))), dim3(64), 0, lp.stream(), B.tpool.as<uint8_t>(), B.qpool.as<uint8_t>(),
                               B.dptab.as<vmx_dp_prob>() + p0, (int)pn, 2, -4, 4, 2, 24, 1, tb_base, B.bnd.as<int32_t>(), B.dpscore.as<int32_t>() + p0, B.order.as<int32_t>(), d_range, d_cnt,
                               d_redo_list, d_redo_cnt, 0, ad_pct, (uint8_t*)nullptr, d_redo_bytes, n_ptr, (unsigned long long)gf_redo_cap, 1);
            lp.join();
        }
        // second launch: the problems whose band was not proven (a few per cent), in full: the larger ones on a whole wave, the others four per wave (its queue is the list the
        // first launch left; any grid works)
        hipLaunchKernelGGL(k_gapfill_fill_ns, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(pn, (int64_t)c->num_cu * 4))), dim3(64), 0, c->stream, B.tpool.as<uint8_t>(), B.qpool.as<uint8_t>(),
                           B.dptab.as<vmx_dp_prob>() + p0, (int)pn, 2, -4, 4, 2, 24, 1, tb_base, B.bnd.as<int32_t>(), B.dpscore.as<int32_t>() + p0, B.order.as<int32_t>(), d_range, d_cnt,
                           d_redo_list, d_redo_cnt, 1, ad_pct, B.tbredo.as<uint8_t>(), d_redo_bytes, n_ptr, (unsigned long long)gf_redo_cap, 1);
        if (ke) (void)hipEventRecord(ke[1], c->stream);
        hipLaunchKernelGGL(k_gapfill_trace, dim3((unsigned)std::max<int64_t>(1, (pn * tr_spread + 63) / 64)), dim3(64), 0, c->stream, B.tpool.as<uint8_t>(), B.qpool.as<uint8_t>(), B.dptab.as<vmx_dp_prob>() + p0, (int)pn, prm->eqx,
                           tb_base, B.run.as<uint32_t>(), B.cig.as<char>(), B.ciglen.as<int32_t>() + p0, B.dpscore.as<int32_t>() + p0, B.tbredo.as<uint8_t>(), tr_spread, B.cigq.as<int32_t>() + p0, n_ptr);
        if (ke) { (void)hipEventRecord(ke[2], c->stream); c->n_gev[pass] = q + 1; }
        gf_chunks[pass] = q + 1;
        return 0;
    };
    auto gapfill = [&](int redo_only) -> int {
        // per-problem tables are sized for the round's capacity
        for (int i = 0; i < 4; ++i) { VMX_TRY(B.dpsz[i].reserve(8 * (size_t)(round_cap + 2))); VMX_TRY(B.dpoff[i].reserve(8 * (size_t)(round_cap + 2))); }
        VMX_TRY(B.dptab.reserve(sizeof(vmx_dp_prob) * (size_t)(round_cap + 1))); VMX_TRY(B.ciglen.reserve(4 * (size_t)(round_cap + 1))); VMX_TRY(B.cigq.reserve(4 * (size_t)(round_cap + 1))); VMX_TRY(B.dpscore.reserve(4 * (size_t)(round_cap + 1)));
        VMX_TRY(B.order.reserve(4 * (size_t)(2 * round_cap + 64)));
        if (!redo_only) {
            // the second launch's pool, from the context's history
            gf_redo_cap = std::max<int64_t>((int64_t)((double)c->redo_need_max * 1.3), std::min<int64_t>((int64_t)512 << 20, std::max<int64_t>(tb_chunk / 8, 1 << 20)));
            if (const char* e = getenv("VMX_TEST_REDO_POOL_BYTES")) { const long long v = atoll(e); if (v > 0 && c->redo_need_max == 0) gf_redo_cap = v; }      // test hook: the first guess of a fresh context
            VMX_TRY(B.tbredo.reserve((size_t)gf_redo_cap + 64));
            gf_redo_cap = (int64_t)B.tbredo.cap - 64;
        }
        c->n_gev[redo_only ? 1 : 0] = 0;
        {
            // ONE launch: string offsets, the four pool-size scans, the problem table and the chunk plan (at most VMX_TB_CHUNK traceback bytes per chunk), then the gather
            int rcg = ext_gather_round(c, B, ix, n, B.ocodes.as<uint8_t>(), d_roff, cur, redo_only, round_cap, pool_cap, 6 + redo_only, true, tb_chunk, nullptr, B.statblk.as<int64_t>() + 64 + (redo_only ? 64 : 0), ad_pct);
            if (rcg < 0) return rcg;
            // the sizing wait of the batch: problem count, pool totals, chunk cuts
            int64_t plan[8 + 2 * (VMX_MAX_CHUNKS + 1)];
            VMX_TRY(vmx_fetch(c, plan, B.statblk.as<int64_t>() + 64 + (redo_only ? 64 : 0), sizeof(plan) / 8));
            VMX_HIP(vmx_stream_sync(c));
            const int cnt = (int)plan[0];
            int64_t totals[4] = {plan[1], plan[2], plan[3], plan[4]};
            if (plan[7] < 0) { set_error("gap fill: more traceback chunks than VMX_MAX_CHUNKS"); return VM_ERR_OOM; }
            st.n_dp_problems += cnt; st.dp_cells += totals[0]; st.dp_string_bytes += plan[5] + plan[6];
            std::vector<int32_t> cuts; std::vector<int64_t> h_tboff_at;            // chunk q = problems [cuts[q], cuts[q+1]); traceback offset at every cut
            for (int q = 0; q <= (int)plan[7]; ++q) { cuts.push_back((int32_t)plan[8 + q]); h_tboff_at.push_back(plan[8 + VMX_MAX_CHUNKS + 1 + q]); }
            if (cuts.size() < 2) { cuts.assign(2, 0); h_tboff_at.assign(2, 0); }
            int64_t tbmax = 0;
            for (size_t q = 0; q + 1 < cuts.size(); ++q) tbmax = std::max(tbmax, h_tboff_at[q + 1] - h_tboff_at[q]);
            VMX_TRY(B.tb.reserve((size_t)tbmax + 64)); VMX_TRY(B.bnd.reserve(4 * (size_t)(totals[1] + 4))); VMX_TRY(B.run.reserve(4 * (size_t)(totals[2] + 4))); VMX_TRY(B.cig.reserve((size_t)totals[3] + 16));
            if (cnt) {
                std::vector<int32_t> csz; for (size_t q = 0; q + 1 < cuts.size(); ++q) csz.push_back(cuts[q + 1] - cuts[q]);
                VMX_TRY(vmx_push(c, B.chunkn, csz.data(), csz.size()));
                for (size_t q = 0; q + 1 < cuts.size(); ++q) VMX_TRY(gf_chunk(redo_only, (int)q, cuts[q], cuts[q + 1] - cuts[q], B.chunkn.as<int32_t>() + q, h_tboff_at[q]));
            }
        }
        A.redo_only = redo_only;
        A.spread = rec_spread;
        hipLaunchKernelGGL(k_ext_records, dim3((unsigned)n), dim3(64), 0, c->stream, A, B.dptab.as<vmx_dp_prob>(), B.cig.as<char>(), B.ciglen.as<int32_t>(), B.cigq.as<int32_t>());     // one wavefront per read
        cur ^= 1;
        return 0;
    };
    // pass 0
    phase(0);
    {   // divergence filter: edit distance of every segment (:19251)
        int rc0 = ext_gather_round(c, B, ix, n, B.ocodes.as<uint8_t>(), d_roff, cur, 0, round_cap, pool_cap, 0, false, 0);
        if (rc0 < 0) return rc0;
        const int64_t cnt = round_cap;                           // launch widths only: every kernel below reads the real count on the device
        {
            VMX_TRY(B.order.reserve(4 * (size_t)(2 * round_cap + 64))); VMX_TRY(B.qrange.reserve(128)); VMX_TRY(B.szh.reserve(4 * 520));
            int32_t* d_range = B.qrange.as<int32_t>(); int32_t* d_cnt = d_range + 4; int32_t* d_nfull = d_range + 8;
            VMX_TRY(B.dpsz[0].reserve(8 * (size_t)(round_cap + 2))); VMX_TRY(B.dpsz[1].reserve(8 * (size_t)(round_cap + 2)));
            // tier 0 (k_ed_anchor_bound, k_ext.hip): the segment's own anchors cut the pair into independent short pieces whose summed cost
            // bounds the edit distance from above; then (k_ed_band.hip) four problems per wave in a +-320-row band, one problem per wave in
            // a +-768-row band, and the exact unbanded kernel, each only for the problems the tier before could not prove "keep" for
            int32_t* d_n2 = d_range + 9; int32_t* d_n1 = d_range + 10;
            (void)hipMemsetAsync(d_nfull, 0, 12, c->stream);
            hipLaunchKernelGGL(k_ed_anchor_bound, dim3((unsigned)std::min<int64_t>(cnt, (int64_t)c->num_cu * 16)), dim3(64), 0, c->stream, A, B.probread.as<int32_t>(),
                               B.qpool.as<uint8_t>(), B.qoff.as<int64_t>(), B.tpool.as<uint8_t>(), B.toff.as<int64_t>(), B.dpsz[1].as<int64_t>());
            // Round 6: the tiers behind the anchor bound are hardly ever needed (problems left after tier 0 per batch: 0 on ONT and HiFi reads at hg38 size, 0.07 on the
            // vacsim workload), yet every batch paid for them: two size-ordered banded launches, three flag passes, and a host wait to learn whether the exact kernel —
            // whose wide workgroups wait for room on a full GPU even when empty — had anything to do. Now a batch runs tier 0 and ONE flag pass that marks the READ of every
            // problem it could not settle (VMX_EXT_NEED_EXACT_DEV); align_device runs those reads again alone with all tiers (c->force_exact), exact kernel included.
            static const bool always_all = getenv("VMX_FORCE_EXACT") != nullptr;        // A/B and test knob: every batch runs every tier
            const bool all_tiers = c->force_exact || always_all;
            hipLaunchKernelGGL(k_ed_flag, dim3((unsigned)std::min<int64_t>((cnt + 255) / 256, 1024)), dim3(256), 0, c->stream, B.dpsz[1].as<int64_t>(), B.qoff.as<int64_t>(), B.toff.as<int64_t>(),
                               c->rc_cur, prm->maxdivergence, B.dpsz[0].as<int64_t>(), B.edout.as<int64_t>(), d_n1, 1, B.probread.as<int32_t>(), all_tiers ? (vmx_ext_read*)nullptr : B.er.as<vmx_ext_read>());
            if (all_tiers) {
            vmx_size_order_wide(c, B.dpsz[0].as<int64_t>(), c->rc_cur, cnt, (int64_t)VMX_ED_LONG, B.order.as<int32_t>(), d_range, d_cnt, B.szh.as<int32_t>());
            hipLaunchKernelGGL(k_ed_banded4, dim3((unsigned)std::min<int64_t>((cnt + 3) / 4, (int64_t)c->num_cu * 16)), dim3(64), 0, c->stream, B.qpool.as<uint8_t>(), B.qoff.as<int64_t>(),
                               B.tpool.as<uint8_t>(), B.toff.as<int64_t>(), B.order.as<int32_t>(), d_range, d_cnt + 2, B.dpsz[1].as<int64_t>());
            hipLaunchKernelGGL(k_ed_flag, dim3((unsigned)std::min<int64_t>((cnt + 255) / 256, 1024)), dim3(256), 0, c->stream, B.dpsz[1].as<int64_t>(), B.qoff.as<int64_t>(), B.toff.as<int64_t>(),
                               c->rc_cur, prm->maxdivergence, B.dpsz[0].as<int64_t>(), B.edout.as<int64_t>(), d_n2, 0, (const int32_t*)nullptr, (vmx_ext_read*)nullptr);
            vmx_size_order_wide(c, B.dpsz[0].as<int64_t>(), c->rc_cur, cnt, (int64_t)VMX_ED_LONG, B.order.as<int32_t>(), d_range, d_cnt, B.szh.as<int32_t>());
            hipLaunchKernelGGL(k_ed_banded, dim3((unsigned)std::min<int64_t>(cnt, (int64_t)c->num_cu * 16)), dim3(64), 0, c->stream, B.qpool.as<uint8_t>(), B.qoff.as<int64_t>(),
                               B.tpool.as<uint8_t>(), B.toff.as<int64_t>(), B.order.as<int32_t>(), d_range, d_cnt + 2, B.dpsz[1].as<int64_t>());
            hipLaunchKernelGGL(k_ed_flag, dim3((unsigned)std::min<int64_t>((cnt + 255) / 256, 1024)), dim3(256), 0, c->stream, B.dpsz[1].as<int64_t>(), B.qoff.as<int64_t>(), B.toff.as<int64_t>(),
                               c->rc_cur, prm->maxdivergence, B.dpsz[0].as<int64_t>(), B.edout.as<int64_t>(), d_nfull, 0, (const int32_t*)nullptr, (vmx_ext_read*)nullptr);
            vmx_size_order_wide(c, B.dpsz[0].as<int64_t>(), c->rc_cur, cnt, (int64_t)VMX_ED_LONG, B.order.as<int32_t>(), d_range, d_cnt, B.szh.as<int32_t>());
            // the exact tier, unconditionally (no read-back of the count: these are side batches of a few reads, or the A/B knob)
            for (int which = 0; which < 2; ++which)
                hipLaunchKernelGGL(k_edit_distance, dim3((unsigned)std::min<int64_t>(cnt, ed_wgs * (which == 0 ? 1 : 2))), dim3(which == 0 ? 64 * VMX_ED_WAVES : 256), 0, c->stream,
                                   B.qpool.as<uint8_t>(), B.qoff.as<int64_t>(), B.tpool.as<uint8_t>(), B.toff.as<int64_t>(),
                                   B.carry.as<int8_t>() + (which ? (size_t)VMX_ED_WAVES * (size_t)carry_stride * (size_t)ed_wgs : 0), B.order.as<int32_t>(), d_range, d_cnt, which,
                                   B.edout.as<int64_t>(), carry_stride, B.oflow.as<int32_t>());
            }
        }
        cur ^= 1;
    }
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));
    phase(1); VMX_TRY(ext_round(0));
    phase(2); VMX_TRY(ext_round(0));
    phase(3); VMX_TRY(ext_round(0));
    phase(4); VMX_TRY(ext_round(0));
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));
    phase(5); VMX_TRY(gapfill(0));
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));
    // pass 1: reads whose CIGARs carry paired indels are re-run with nofilter (:24079-24080)
    int64_t test_side_every = 0; if (const char* e = getenv("VMX_TEST_SIDE_EVERY")) test_side_every = atoll(e);      // test hook: every k-th read of a batch is sent through the side batch
    static const bool always_pass1 = getenv("VMX_FORCE_PASS1") != nullptr;        // A/B and test knob: every batch runs pass 1 (rounds 1-5)
    const bool do_pass1 = c->run_pass1 || always_pass1;
    if (do_pass1) {
        A.redo_only = 1;
        phase(3); phase(4); phase(5); VMX_TRY(gapfill(1));
        A.redo_only = 0;
    }
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));

    // ---------------- results
    // pack the records of the batch on the device, then copy exactly those bytes
    VMX_TRY(B.dpsz[0].reserve(8 * (size_t)(n + 2))); VMX_TRY(B.dpsz[1].reserve(8 * (size_t)(n + 2))); VMX_TRY(B.dpoff[0].reserve(8 * (size_t)(n + 2))); VMX_TRY(B.dpoff[1].reserve(8 * (size_t)(n + 2)));
    hipLaunchKernelGGL(k_res_sizes, dim3(gridR), dim3(64), 0, c->stream, B.er.as<vmx_ext_read>(), B.rec.as<vm_record>(), B.soff2.as<int64_t>(), (int)n, B.dpsz[0].as<int64_t>(), B.dpsz[1].as<int64_t>());
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, B.dpsz[0].as<int64_t>(), B.dpoff[0].as<int64_t>(), n, 0);
    hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(256), 0, c->stream, B.dpsz[1].as<int64_t>(), B.dpoff[1].as<int64_t>(), n, 0);
    int64_t nr = 0, nb = 0; int32_t oflow = 0;
    int32_t n_full = 0, n_t2 = 0, n_t1 = 0;
    int64_t fin[14]; int64_t hstat[8];
    hipLaunchKernelGGL(k_final_scalars, dim3(1), dim3(1), 0, c->stream, B.dpoff[0].as<int64_t>() + n, B.dpoff[1].as<int64_t>() + n, B.oflow.as<int32_t>(), B.qrange.as<int32_t>() + 8,
                       B.statblk.as<int32_t>(), B.statblk.as<int64_t>() + 32);
    // ONE wait for the results: the packed buffers on the device have room for the worst case (cS records, cB CIGAR bytes), so the pack
    // kernel runs before the totals are known on the host, and the host copies as many records / bytes as the context's history
    // suggests for a batch of this size (records per read, CIGAR bytes per base, + 15 %). Only when a batch exceeds the guess — the
    // first batch of a context always does — the missing tail costs a second copy and wait.
    VMX_TRY(B.totals.reserve(sizeof(vm_record) * (size_t)(cS + 1))); VMX_TRY(B.dupd.reserve((size_t)cB + 64));
    hipLaunchKernelGGL(k_res_pack, dim3((unsigned)std::min<int64_t>(n, (int64_t)c->num_cu * 8)), dim3(64), 0, c->stream, B.er.as<vmx_ext_read>(), B.rec.as<vm_record>(), B.blob.as<char>(),
                       B.soff2.as<int64_t>(), B.bloboff.as<int64_t>(), (int)n, B.dpoff[0].as<int64_t>(), B.dpoff[1].as<int64_t>(), B.totals.as<vm_record>(), B.dupd.as<char>());
    int64_t g_nr = std::min<int64_t>(cS, (int64_t)(c->res_rec_per_read * 2.0 * (double)n) + 256), g_nb = std::min<int64_t>(cB, (int64_t)(c->res_blob_per_base * 1.5 * (double)total_bases) + 65536);      // (records are 40 B: a generous guess costs nothing; a short one costs a second wait)
    if (c->res_rec_per_read <= 0.0) { g_nr = 0; g_nb = 0; }
    std::vector<vmx_ext_read> er((size_t)n);
    std::vector<int64_t> h_gmax((size_t)n);
    *recs = (vm_record*)malloc(sizeof(vm_record) * (size_t)std::max<int64_t>(g_nr, 1)); *cigar_blob = (char*)malloc((size_t)std::max<int64_t>(g_nb, 1));
    // every error return below hands the caller NULL outputs (callers raise without calling vm_free: ADVICE r3)
    struct OutGuard { vm_record** r; char** b; bool keep = false; ~OutGuard() { if (!keep) { free(*r); free(*b); *r = nullptr; *b = nullptr; } } } out_guard{recs, cigar_blob};
    if (!*recs || !*cigar_blob) { set_error("out of host memory"); return VM_ERR_OOM; }
    VMX_TRY(vmx_fetch(c, fin, B.statblk.as<int64_t>() + 32, 14));          // record / blob totals, overflow flag, tier counters, per-round problem counts: one copy
    int64_t geo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (fast_local) VMX_TRY(vmx_fetch(c, geo, B.geotot.p, 6));             // what the geometry kernel found: totals (history), local anchors (statistics)
    std::vector<int32_t> h_ctl(gfctl_bytes / 4);
    VMX_TRY(vmx_fetch(c, h_ctl.data(), B.gfctl.p, h_ctl.size()));          // the gap fill's per-chunk control blocks (second-launch queue length and pool need)
    VMX_TRY(vmx_fetch(c, er.data(), B.er.p, (size_t)n)); VMX_TRY(vmx_fetch(c, h_gmax.data(), B.gmax.p, (size_t)n));
    VMX_TRY(vmx_fetch(c, *recs, B.totals.p, (size_t)g_nr)); VMX_TRY(vmx_fetch(c, *cigar_blob, B.dupd.p, (size_t)g_nb));
    VMX_HIP(hipEventRecord(ev[nev++], c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    nr = fin[0]; nb = fin[1]; oflow = (int32_t)fin[2]; n_full = (int32_t)fin[3]; n_t2 = (int32_t)fin[4]; n_t1 = (int32_t)fin[5];
    if (fast_local) {
        st.n_local_anchors = geo[4];
        if (geo[5] > 0 || geo[3] + 16 > c->geo_cap[3]) c->geo_valid = false;       // this batch outgrew the pools: the next one of this context asks (and grows them)
        if (geo[5] && getenv("VMX_DBG_POOLS")) fprintf(stderr, "[pools] extend stage: %lld read(s) did not fit the history-sized pools (run again alone)\n", (long long)geo[5]);
    }
    {   // what the gap fill assumed instead of asking (above): did it hold?
        bool again = false; int64_t need_max = 0, ad_failed = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int q = 0; q < gf_chunks[pass]; ++q) {
                const int32_t* ctl = h_ctl.data() + (size_t)(pass * (VMX_MAX_CHUNKS + 1) + q) * GF_SLOT;
                unsigned long long rb = 0; memcpy(&rb, ctl + 16, 8);
                st.n_dp_redo += ctl[12]; st.dp_redo_tb_bytes += (int64_t)rb; st.dp_cells += (int64_t)rb; ad_failed += ctl[15];
                need_max = std::max<int64_t>(need_max, (int64_t)rb);
            }
        c->redo_need_max = std::max<long long>(c->redo_need_max, need_max);
        if (need_max > gf_redo_cap) { again = true; if (getenv("VMX_DBG_POOLS")) fprintf(stderr, "[pools] gap fill: a chunk's second launch needs %.3f GB of traceback space, the pool held %.3f: running the batch again\n", need_max / 1e9, gf_redo_cap / 1e9); }
        if (again) return VMX_RETRY_BATCH;
        if (!ad_pinned && !c->force_exact && st.n_dp_problems >= 2000) {      // adapt the band-width rule (above) to the rate of band attempts that were not proven
            const double f = (double)ad_failed / (double)st.n_dp_problems;
            std::lock_guard<std::mutex> g(ad_m);
            if (ad_cur == adr.pct) {                                      // (a batch that ran on an older value does not vote on the current one)
                if (f > 0.03 && adr.pct < 100) { adr.pct = std::min(100, adr.pct + 10); adr.floor = adr.pct; adr.hold = 256; }
                else if (adr.hold > 0) { if (--adr.hold == 0) adr.floor = 20; }
                else if (f < 0.025 && adr.pct > adr.floor) adr.pct = std::max(adr.floor, adr.pct - (f < 0.01 ? 20 : 10));
            }
        }
    }
    for (int i = 0; i < 8; ++i) hstat[i] = fin[6 + i];
    st.n_segments = hstat[0]; st.n_ed_problems = hstat[0]; st.n_ext_problems = (int64_t)hstat[1] + hstat[2] + hstat[3] + hstat[4];
    st.n_ed_full = n_full; st.n_ed_tier2 = n_t2; st.n_ed_tier1 = n_t1;
    if (oflow) { set_error("extend stage: a per-batch work pool overflowed"); return VM_ERR_OOM; }
    if (nr > g_nr || nb > g_nb) {                                 // the guess was short: fetch the tails
        if (nr > g_nr) { vm_record* p2 = (vm_record*)realloc(*recs, sizeof(vm_record) * (size_t)nr); if (!p2) { set_error("out of host memory"); return VM_ERR_OOM; } *recs = p2;
                         VMX_TRY(vmx_fetch(c, *recs + g_nr, B.totals.as<vm_record>() + g_nr, (size_t)(nr - g_nr))); }
        if (nb > g_nb) { char* p2 = (char*)realloc(*cigar_blob, (size_t)nb); if (!p2) { set_error("out of host memory"); return VM_ERR_OOM; } *cigar_blob = p2;
                         VMX_TRY(vmx_fetch(c, *cigar_blob + g_nb, B.dupd.as<char>() + g_nb, (size_t)(nb - g_nb))); }
        VMX_HIP(vmx_stream_sync(c));
    }
    if (n > 0 && total_bases > 0) { c->res_rec_per_read = std::max(c->res_rec_per_read, (double)nr / (double)n); c->res_blob_per_base = std::max(c->res_blob_per_base, (double)nb / (double)total_bases); }
    for (int64_t r = 0; r < n; ++r) {
        int stt2 = er[r].status == VMX_EXT_CAPACITY_DEV ? VMX_EXT_SHORT_INTERNAL : (er[r].status == VMX_EXT_NEED_EXACT_DEV ? VMX_EXT_EXACT_INTERNAL : er[r].status);
        if (fast_local && (er[r].status == VM_READ_CAPACITY_DEV || er[r].status == VM_READ_BANDFALL_DEV)) stt2 = VMX_EXT_EXACT_INTERNAL;      // the local stage's retries live on the waiting path
        if (stt2 == 0 && !do_pass1 && er[r].active && er[r].redo == 1 && er[r].pass == 1) stt2 = VMX_EXT_EXACT_INTERNAL;      // asks for pass 1, which this batch did not run: again, alone
        if (stt2 == 0 && test_side_every > 0 && !c->force_exact && r % test_side_every == 0) stt2 = VMX_EXT_EXACT_INTERNAL;       // (test hook)      // (align_device runs these reads again alone: larger pools / all tiers of the divergence filter)
        if (h_gmax[r] == -2 && (h_aoff[r + 1] - h_aoff[r]) > 2) stt2 = VM_READ_RAISED;     // GC-fast: the reference raises on this read (k_chain_fast.hip)
        if (h_gmax[r] == -4) stt2 = VM_READ_UNSUPPORTED;                                    // -mode asm: a contig of 500 kb or more inside align_device (vm_align_batch routes those to vmx_asm.hip)
        if (!asm_override.empty() && asm_override[(size_t)r] != 0) stt2 = asm_override[(size_t)r];
        if (rmode == 1 && (h_aoff[r + 1] - h_aoff[r]) <= 2) stt2 = VM_READ_RAISED;              // mode R returns an unbound `factor` for <= 2 anchors (mammap_noprefercloser.py:24417)
        if (status_per_read) status_per_read[r] = stt2;
        if (stt2 != 0) { st.n_failed++; continue; }
        if (er[r].nrec == 0) st.n_unmapped++;
    }
    for (int64_t i = 0; i < nr; ++i) { st.aligned_bases += (*recs)[i].q_en - (*recs)[i].q_st; st.cigar_bytes += (*recs)[i].cigar_len; }
    *n_recs = nr; st.n_records = nr;
    float ms = 0;
    hipEventElapsedTime(&ms, ev[0], ev[nev - 1]); st.ms_total = ms;
    for (int i = 1; i < nev && i < 16; ++i) { hipEventElapsedTime(&ms, ev[i - 1], ev[i]); st.ms_stage[i - 1] = ms; }
    if ((c->kev_set & 1) && hipEventElapsedTime(&ms, c->kev[0], c->kev[1]) == hipSuccess) st.ms_local_seed = ms;
    if ((c->kev_set & 2) && hipEventElapsedTime(&ms, c->kev[2], c->kev[3]) == hipSuccess) st.ms_cluster = ms;
    st.n_host_syncs = c->n_syncs; st.n_local_general = c->n_bandfall;
    // host-side view of the batch (tuning): [14] ms the thread spent inside its waits for the stream, [15] wall ms of the call up to here — the difference is host
    // work done while this context's stream was EMPTY (sizing, vector work, launches between a wait's return and the next kernel)
    st.ms_stage[14] = (float)((c->sync_wait_ns + download_wait_ns()) * 1e-6);
    st.ms_stage[15] = (float)((std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->call_t0_ns) * 1e-6);
    for (int pass = 0; pass < 2; ++pass)
        for (int q = 0; q < c->n_gev[pass]; ++q) {
            hipEvent_t* ke = c->gev + 24 * pass + 3 * q;
            if (hipEventElapsedTime(&ms, ke[0], ke[1]) == hipSuccess) st.ms_gapfill_fill += ms;
            if (hipEventElapsedTime(&ms, ke[1], ke[2]) == hipSuccess) st.ms_gapfill_trace += ms;
            st.n_gapfill_launches++;
        }
    if (stats) *stats = st;
    if (getenv("VMX_DBG_POOLS")) {      // tuning aid: which grow-only pools hold the context's HBM (index in declaration order of vmx_batch_bufs / vmx_local_bufs)
        const DevBuf* bb = (const DevBuf*)&B; size_t tot = 0;
        for (size_t i = 0; i < sizeof(B) / sizeof(DevBuf); ++i) { tot += bb[i].cap; if (bb[i].cap > ((size_t)128 << 20)) fprintf(stderr, "[pools] batch[%zu] %.2f GB\n", i, bb[i].cap / 1e9); }
        fprintf(stderr, "[pools] batch bufs total %.2f GB\n", tot / 1e9);
        const DevBuf* lb = (const DevBuf*)&L; size_t lt = 0; const size_t nl = (size_t)((const char*)&L.la_pool_rows - (const char*)&L) / sizeof(DevBuf);
        for (size_t i = 0; i < nl; ++i) { lt += lb[i].cap; if (lb[i].cap > ((size_t)128 << 20)) fprintf(stderr, "[pools] local[%zu] %.2f GB\n", i, lb[i].cap / 1e9); }
        fprintf(stderr, "[pools] local bufs total %.2f GB\n", lt / 1e9);
        fprintf(stderr, "[pools] process so far: %lld first allocations, %lld re-allocations, %.3f s in hipFree + hipMalloc\n", (long long)vmx::devbuf_stats().first.load(),
                (long long)vmx::devbuf_stats().grows.load(), vmx::devbuf_stats().ns.load() * 1e-9);
    }
    if (stats) *stats = st;
    out_guard.keep = true;
    return VM_OK;
}

// One call = at most max_bases bases and VMX_MAX_BATCH_READS reads per pass over the path: a larger one is processed as consecutive sub-batches of the same
// context and its results are concatenated. run_one(a, b, ...) aligns the reads [a, b). A range the device has no memory for (the grow-only pools of
// this and the other contexts, the index and the reads share the HBM) is cut in two and tried again instead of failing the whole call with VM_ERR_OOM —
// down to single reads.
template <class RunOne>
static int align_in_sub_batches(int64_t n, const int64_t* offsets, int64_t max_bases, RunOne run_one, vm_record** recs, int64_t* n_recs, char** cigar_blob, vm_batch_stats* stats) {
    if (n <= VMX_MAX_BATCH_READS && offsets[n] - offsets[0] <= max_bases) {
        const int rc = run_one(0, n, recs, n_recs, cigar_blob, stats);
        if (rc != VM_ERR_OOM || n <= 1) return rc;
        free(*recs); free(*cigar_blob);                          // fall through: the batch is cut into pieces that fit
    }
    *recs = nullptr; *n_recs = 0; *cigar_blob = nullptr;
    std::vector<vm_record> all; std::string blob;
    vm_batch_stats tot; memset(&tot, 0, sizeof tot);
    std::vector<std::pair<int64_t, int64_t>> work;
    for (int64_t a = 0; a < n;) {
        int64_t b = a + 1;                                       // at least one read per sub-batch, whatever its length
        while (b < n && b - a < VMX_MAX_BATCH_READS && offsets[b + 1] - offsets[a] <= max_bases) ++b;
        work.emplace_back(a, b); a = b;
    }
    for (size_t wi = 0; wi < work.size(); ++wi) {
        const int64_t a = work[wi].first, b = work[wi].second;
        vm_record* r = nullptr; int64_t nr = 0; char* cb = nullptr; vm_batch_stats st;
        const int rc = run_one(a, b, &r, &nr, &cb, &st);
        if (rc == VM_ERR_OOM && b - a > 1) {
            free(r); free(cb);
            vmx::devbuf_retired().flush();                       // parked (outgrown) allocations go back to the device first
            const int64_t mid = a + (b - a) / 2;
            work[wi] = std::make_pair(a, mid); work.insert(work.begin() + (std::ptrdiff_t)wi + 1, std::make_pair(mid, b));
            --wi; continue;
        }
        if (rc < 0) { free(r); free(cb); return rc; }
        int64_t bl = 0; for (int64_t i = 0; i < nr; ++i) bl = std::max(bl, r[i].cigar_off + r[i].cigar_len);
        for (int64_t i = 0; i < nr; ++i) { vm_record x = r[i]; x.read_idx += (int32_t)a; x.cigar_off += (int64_t)blob.size(); all.push_back(x); }
        blob.append(cb, (size_t)bl); blob.push_back('\0');
        free(r); free(cb);
        {   // counters and times add up; sizes are per call
            int64_t* d = (int64_t*)&tot; const int64_t* s2 = (const int64_t*)&st;
            for (size_t i = 0; i < offsetof(vm_batch_stats, ms_total) / 8; ++i) d[i] += s2[i];
            tot.ms_total += st.ms_total; for (int i = 0; i < 16; ++i) tot.ms_stage[i] += st.ms_stage[i];
            tot.ms_gapfill_fill += st.ms_gapfill_fill; tot.ms_gapfill_trace += st.ms_gapfill_trace; tot.n_gapfill_launches += st.n_gapfill_launches;
            tot.n_ed_full += st.n_ed_full; tot.n_ed_tier2 += st.n_ed_tier2; tot.n_ed_tier1 += st.n_ed_tier1;
            tot.n_dp_redo += st.n_dp_redo; tot.dp_redo_tb_bytes += st.dp_redo_tb_bytes; tot.ms_local_seed += st.ms_local_seed; tot.ms_cluster += st.ms_cluster; tot.n_host_syncs += st.n_host_syncs; tot.n_local_general += st.n_local_general; tot.n_ext_retries += st.n_ext_retries; tot.n_batch_retries += st.n_batch_retries;
        }
    }
    *recs = (vm_record*)malloc(sizeof(vm_record) * std::max<size_t>(all.size(), 1)); *cigar_blob = (char*)malloc(std::max<size_t>(blob.size(), 1));
    if (!*recs || !*cigar_blob) { free(*recs); free(*cigar_blob); *recs = nullptr; *cigar_blob = nullptr; set_error("out of host memory"); return VM_ERR_OOM; }
    memcpy(*recs, all.data(), sizeof(vm_record) * all.size()); memcpy(*cigar_blob, blob.data(), blob.size());
    *n_recs = (int64_t)all.size();
    if (stats) *stats = tot;
    return VM_OK;
}
// the bases one pass over the path takes (VMX_MAX_BATCH_BASES). The work pools of a context grow with the pass (~0.85 KB per read base at hg38 size) and never
// shrink, and a scheduler that cuts its batches by read count meets a batch of its longest reads now and then — 4096 reads of 40-100 kb are 160 Mbases
// where the average batch holds 60: every context then keeps pools for 160 Mbases and uses them for 60. With the budget near the AVERAGE batch the
// long-read batch runs as two or three passes of its context and the pools of all contexts are ~2.5 times smaller: more batches in flight fit the HBM.
static int64_t vmx_pass_bases() {
    static const int64_t v = [] { const char* e = getenv("VMX_MAX_BATCH_BASES"); const long long x = e ? atoll(e) : 0; return x > 0 ? (int64_t)x : (int64_t)VMX_MAX_BATCH_BASES; }();
    return v;
}

extern "C" {

int vm_reads_upload(vm_ctx* c, int64_t n, const char* seqs, const int64_t* offsets, vm_reads** out) {
    *out = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    vm_reads* R = new vm_reads(); R->ctx = c; R->n = n; R->h_off.assign(offsets, offsets + n + 1);
    const int64_t tot = offsets[n];
    int rc = 0;
    if ((rc = upload(R->raw, seqs, (size_t)tot, c->stream)) < 0 || (rc = R->codes.reserve((size_t)tot + 64)) < 0 || (rc = upload(R->off, offsets, (size_t)n + 1, c->stream)) < 0) { delete R; return rc; }
    if (tot) LAUNCH1D(k_encode, tot, R->raw.as<char>(), R->codes.as<uint8_t>(), tot);
    VMX_HIP(vmx_stream_sync(c));
    R->raw.release();
    *out = R;
    return VM_OK;
}
// the same into an existing object (its device buffers are kept and only grow): an uploader that streams batches ahead of the aligning contexts
// cycles a few of these instead of paying a hipMalloc / hipFree pair per batch (hipFree waits for the whole device). No alignment of R may be in flight.
int vm_reads_reupload(vm_ctx* c, vm_reads* R, int64_t n, const char* seqs, const int64_t* offsets) {
    if (!c || !R) { set_error("no context / reads object"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    R->n = n; R->h_off.assign(offsets, offsets + n + 1);
    const int64_t tot = offsets[n];
    VMX_TRY(upload(R->raw, seqs, (size_t)tot, c->stream)); VMX_TRY(R->codes.reserve((size_t)tot + 64)); VMX_TRY(upload(R->off, offsets, (size_t)n + 1, c->stream));
    if (tot) LAUNCH1D(k_encode, tot, R->raw.as<char>(), R->codes.as<uint8_t>(), tot);
    VMX_HIP(vmx_stream_sync(c));
    return VM_OK;
}
void vm_reads_free(vm_reads* R) { if (!R) return; R->raw.release(); R->codes.release(); R->off.release(); delete R; }

int vm_align_resident(vm_ctx* c, const vm_index* mi, const vm_params* prm, const vm_reads* R, vm_record** recs, int64_t* n_recs, char** cigar_blob,
                      int32_t* status_per_read, vm_batch_stats* stats) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    if (R->n <= VMX_MAX_BATCH_READS && R->h_off[(size_t)R->n] <= vmx_pass_bases())
        return align_device(c, mi, prm, R->n, R->codes.as<uint8_t>(), R->off.as<int64_t>(), R->h_off, recs, n_recs, cigar_blob, status_per_read, stats);
    // more bases than one pass takes: consecutive ranges of the resident reads (their codes stay where they are; a range's offsets are rebased and uploaded)
    vmx_batch_bufs& B = *batch_bufs(c);
    auto run_one = [&](int64_t a, int64_t b, vm_record** r, int64_t* nr, char** cb, vm_batch_stats* st) -> int {
        std::vector<int64_t> h((size_t)(b - a) + 1);
        for (int64_t i = a; i <= b; ++i) h[(size_t)(i - a)] = R->h_off[(size_t)i] - R->h_off[(size_t)a];
        VMX_TRY(vmx_push(c, B.off, h.data(), h.size()));
        return align_device(c, mi, prm, b - a, R->codes.as<uint8_t>() + R->h_off[(size_t)a], B.off.as<int64_t>(), h, r, nr, cb, status_per_read ? status_per_read + a : nullptr, st);
    };
    return align_in_sub_batches(R->n, R->h_off.data(), vmx_pass_bases(), run_one, recs, n_recs, cigar_blob, stats);
}

int vm_align_trace(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets, int stage, int64_t** rows, int64_t** row_off) {
    *rows = nullptr; *row_off = nullptr;
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (stage != 0 && stage != 3 && stage != 5) { set_error("vm_align_trace: stage must be 0, 3 or 5"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(c->device));
    vmx_batch_bufs& B = *batch_bufs(c);
    const int64_t tot = offsets[n];
    VMX_TRY(upload(B.raw, seqs, (size_t)tot, c->stream)); VMX_TRY(B.codes.reserve((size_t)tot + 64)); VMX_TRY(upload(B.off, offsets, (size_t)n + 1, c->stream));
    if (tot) LAUNCH1D(k_encode, tot, B.raw.as<char>(), B.codes.as<uint8_t>(), tot);
    std::vector<int64_t> h_off(offsets, offsets + n + 1);
    vmx_seg_trace tr; tr.stage = stage;
    vm_record* recs = nullptr; int64_t nrec = 0; char* blob = nullptr;
    std::vector<int32_t> status((size_t)n + 1);
    const int rc = align_device(c, mi, prm, n, B.codes.as<uint8_t>(), B.off.as<int64_t>(), h_off, &recs, &nrec, &blob, status.data(), nullptr, &tr);
    free(recs); free(blob);
    if (rc < 0) return rc;
    if (tr.off.empty()) tr.off.assign((size_t)n + 1, 0);
    *rows = (int64_t*)malloc(8 * std::max<size_t>(tr.rows.size(), 1)); *row_off = (int64_t*)malloc(8 * tr.off.size());
    memcpy(*rows, tr.rows.data(), 8 * tr.rows.size()); memcpy(*row_off, tr.off.data(), 8 * tr.off.size());
    return VM_OK;
}

static int align_batch_one(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets, vm_record** recs, int64_t* n_recs,
                           char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats) {
    vmx_batch_bufs& B = *batch_bufs(c);
    const int64_t base = offsets[0], tot = offsets[n] - base;
    std::vector<int64_t> h_off((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) h_off[(size_t)i] = offsets[i] - base;
    VMX_TRY(vmx_push(c, B.raw, seqs + base, (size_t)tot)); VMX_TRY(B.codes.reserve((size_t)tot + 64)); VMX_TRY(vmx_push(c, B.off, h_off.data(), (size_t)n + 1));
    if (tot) LAUNCH1D(k_encode, tot, B.raw.as<char>(), B.codes.as<uint8_t>(), tot);
    return align_device(c, mi, prm, n, B.codes.as<uint8_t>(), B.off.as<int64_t>(), h_off, recs, n_recs, cigar_blob, status_per_read, stats);
}

// The work pools of a context grow with the batch (≈ 0.85 KB per read base at hg38 size) and never shrink, so one oversized call —
// 100 k long reads in one batch — could exhaust HBM next to the index and the other contexts. A call above VMX_MAX_BATCH_BASES bases or
// VMX_MAX_BATCH_READS reads is therefore processed as consecutive sub-batches of the same context and its results are concatenated;
// callers that size their batches like vacmap_amd.pipeline (4096 reads, ~60 Mbases) never notice.
int vm_align_batch(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets, vm_record** recs, int64_t* n_recs,
                   char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    VMX_HIP(hipSetDevice(c->device));
    if (prm->mode == VM_MODE_ASM)        // contigs of 500 kb and more take the linked path (vmx_asm.hip), one by one
        for (int64_t r = 0; r < n; ++r)
            if (offsets[r + 1] - offsets[r] >= 500000) return vmx_align_batch_asm_mixed(c, mi, prm, n, seqs, offsets, recs, n_recs, cigar_blob, status_per_read, stats);
    // test hook: pretend the device is out of memory for any sub-batch above this many bases (exercises the degradation)
    const char* oom_env = getenv("VMX_TEST_OOM_ABOVE_BASES");
    const int64_t fake_oom = oom_env ? atoll(oom_env) : 0;
    auto run_one = [&](int64_t a, int64_t b, vm_record** r, int64_t* nr, char** cb, vm_batch_stats* st) -> int {
        if (fake_oom > 0 && offsets[b] - offsets[a] > fake_oom) { *r = nullptr; *nr = 0; *cb = nullptr; set_error("out of device memory (test hook)"); return VM_ERR_OOM; }
        return align_batch_one(c, mi, prm, b - a, seqs, offsets + a, r, nr, cb, status_per_read ? status_per_read + a : nullptr, st);
    };
    return align_in_sub_batches(n, offsets, vmx_pass_bases(), run_one, recs, n_recs, cigar_blob, stats);
}

}  // extern "C"
