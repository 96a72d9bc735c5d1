// vmx_asm.hip — host side of -mode asm's batch-linked chain DPs (k_chain_linked.hip): the stage entry vm_chain_linked.
// The per-read function of the fork (contigs below 500 kb) runs through vm_align_batch with VM_MODE_ASM (vmx_align.hip).
#include "vmx_host.h"
#include "vmx_link.h"
#include "vmx_index_prim.h"
#include <cstring>

using namespace vmx;

__global__ void k_chain_linked(vmx_link_job* jobs, int n_jobs, vmx_tables tab, const double* gapcost_list, double skipcost, int maxdiff, int maxgap, int lc, double margin_base, double max_factor);
// max_factor of the fork's GC-exact (mammap_asm.py:21757). Test hook VMX_TEST_ASM_MAX_FACTOR: a lower value drives batches into the linked GC-fast
__global__ void k_chain_linked_win(vmx_link_job* jobs, int n_jobs, vmx_tables tab, const double* gapcost_list, double skipcost, int maxdiff, int maxgap, int lc, double max_factor);
// VMX_LINK_PLAIN=1: the plain form of the linked DP (k_chain_linked.hip) instead of the register-window form (k_chain_linked_win, k_chain.hip)
static bool asm_link_plain() { static const bool v = getenv("VMX_LINK_PLAIN") != nullptr; return v; }
static double asm_max_factor() { const char* e = getenv("VMX_TEST_ASM_MAX_FACTOR"); return e ? atof(e) : 1000.0; }
__global__ void k_link_carry(vmx_link_job* jobs, int n_jobs, double skipcost);
__global__ void k_link_place(vmx_link_job* jobs, int n_jobs);
__global__ void k_link_sort_keys(const vmx_anchor* rows, int64_t n, uint64_t* key, uint64_t* idx);
__global__ void k_link_sort_gather(const vmx_anchor* rows, const uint64_t* idx, int64_t n, vmx_anchor* out);
__global__ void k_chain_linked_fast(vmx_link_job* jobs, int n_jobs, vmx_tables tab, const double* gapcost_list, double skipcost, int maxdiff, int maxgap);

namespace {
struct Bufs { Bufs() = default; Bufs(const Bufs&) = delete; DevBuf st, preS, preP, preR, rows, S, P, SA, job, gap; ~Bufs() { st.release(); preS.release(); preP.release(); preR.release(); rows.release(); S.release(); P.release(); SA.release(); job.release(); gap.release(); } };
}

extern "C" void vm_linked_out_free(vm_linked_out* o) {
    if (!o) return;
    free(o->S); free(o->P); free(o->S_arg_hot); free(o->carry_S); free(o->carry_P); free(o->carry_rows);
    memset(o, 0, sizeof(*o));
}

extern "C" int vm_chain_linked(vm_ctx* c, int which, int kmersize, double skipcost, int maxdiff, int maxgap, int64_t n, const int64_t* rows, int64_t n_pre,
                               const double* pre_S, const int64_t* pre_P, double g_max_scores, int64_t g_max_index, int64_t prereadloc, vm_linked_out* out) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (!out || n <= 0 || n_pre < 0 || n_pre >= n || which < 0 || which > 2 || maxdiff > 62) { set_error("vm_chain_linked: bad arguments"); return VM_ERR_ARG; }
    memset(out, 0, sizeof(*out));
    VMX_HIP(hipSetDevice(c->device));
    const int lc = which == 2;
    const int cap_pre = (int)std::max<int64_t>(4096, n_pre);
    const int64_t n_new = n - n_pre;
    Bufs B;
    const HostTables& T = host_tables();
    std::vector<double> gap(64, 0.0); double gapmax = 0.0;
    for (int g = 1; g <= maxdiff; ++g) { gap[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]); gapmax = std::max(gapmax, gap[g]); }   // mammap_asm.py:21700 / :21509
    double rgcmax = 0.0; if (lc) for (int r = 0; r < 100; ++r) rgcmax = std::max(rgcmax, (double)T.readgap_r[r]);
    const double margin_base = std::max(skipcost + 36.0, gapmax + rgcmax);      // `pen` of k_chain_linked.hip's header (36 = extra's last and largest entry)
    VMX_TRY(upload(B.gap, gap.data(), 64, c->stream));
    // staging + state
    VMX_TRY(B.preS.reserve(8 * (size_t)cap_pre)); VMX_TRY(B.preP.reserve(4 * (size_t)cap_pre)); VMX_TRY(B.preR.reserve(sizeof(vmx_anchor) * (size_t)cap_pre));
    std::vector<vmx_anchor> hr((size_t)n);
    for (int64_t i = 0; i < n; ++i) { vmx_anchor a; a.q = (int32_t)rows[4 * i]; a.r = rows[4 * i + 1]; a.s = (int16_t)rows[4 * i + 2]; a.l = (int16_t)rows[4 * i + 3]; hr[(size_t)i] = a; }
    std::vector<int32_t> hp((size_t)n_pre);
    for (int64_t i = 0; i < n_pre; ++i) hp[(size_t)i] = (int32_t)pre_P[i];
    if (n_pre) {
        VMX_HIP(hipMemcpyAsync(B.preS.p, pre_S, 8 * (size_t)n_pre, hipMemcpyHostToDevice, c->stream));
        VMX_HIP(hipMemcpyAsync(B.preP.p, hp.data(), 4 * (size_t)n_pre, hipMemcpyHostToDevice, c->stream));
        VMX_HIP(hipMemcpyAsync(B.preR.p, hr.data(), sizeof(vmx_anchor) * (size_t)n_pre, hipMemcpyHostToDevice, c->stream));
    }
    vmx_link_state hs; memset(&hs, 0, sizeof hs);
    hs.g_max_scores = g_max_scores; hs.g_max_index = (int32_t)g_max_index; hs.n_pre = (int32_t)n_pre; hs.prereadloc = prereadloc; hs.cap_pre = cap_pre;
    hs.pre_S = B.preS.as<double>(); hs.pre_P = B.preP.as<int32_t>(); hs.pre_rows = B.preR.as<vmx_anchor>();
    VMX_TRY(upload(B.st, &hs, 1, c->stream));
    const size_t tot = (size_t)cap_pre + (size_t)n_new;
    VMX_TRY(B.rows.reserve(sizeof(vmx_anchor) * tot)); VMX_TRY(B.S.reserve(8 * tot)); VMX_TRY(B.P.reserve(4 * tot)); VMX_TRY(B.SA.reserve(4 * tot));
    VMX_HIP(hipMemcpyAsync(B.rows.as<vmx_anchor>() + cap_pre, hr.data() + n_pre, sizeof(vmx_anchor) * (size_t)n_new, hipMemcpyHostToDevice, c->stream));
    vmx_link_job hj; memset(&hj, 0, sizeof hj);
    hj.state = B.st.as<vmx_link_state>(); hj.rows = B.rows.as<vmx_anchor>(); hj.S = B.S.as<double>(); hj.P = B.P.as<int32_t>(); hj.SA = B.SA.as<int32_t>();
    hj.cap_pre = cap_pre; hj.n_new = (int32_t)n_new;
    DevBuf fSi, fT, fCNT; struct RelF { DevBuf *a, *b, *c3; ~RelF() { a->release(); b->release(); c3->release(); } } relf{&fSi, &fT, &fCNT};
    if (which == 1) {                                            // the fork's linked GC-fast on its own (:21871; with n_pre = 0 its plain GC-fast, :20738)
        VMX_TRY(fSi.reserve(4 * tot)); VMX_TRY(fT.reserve(8 * tot)); VMX_TRY(fCNT.reserve(4 * ((size_t)hr.back().q + 50 + 64)));
        hj.Si = fSi.as<int32_t>(); hj.T = fT.as<int64_t>(); hj.CNT = fCNT.as<int32_t>(); hj.ran = 1; hj.gmax = -1;
    }
    VMX_TRY(upload(B.job, &hj, 1, c->stream));
    vmx_link_job* d_job = B.job.as<vmx_link_job>(); const double* d_gap = B.gap.as<double>(); const vmx_tables tabs = c->tables; hipStream_t stq = c->stream;
    hipLaunchKernelGGL(k_link_place, dim3(1), dim3(256), 0, stq, d_job, 1);
    if (which == 1) hipLaunchKernelGGL(k_chain_linked_fast, dim3(1), dim3(64), 0, stq, d_job, 1, tabs, d_gap, skipcost, maxdiff, maxgap);
    else if (asm_link_plain()) hipLaunchKernelGGL(k_chain_linked, dim3(1), dim3(64), 0, stq, d_job, 1, tabs, d_gap, skipcost, maxdiff, maxgap, lc, margin_base, asm_max_factor());
    else hipLaunchKernelGGL(k_chain_linked_win, dim3(1), dim3(64), 0, stq, d_job, 1, tabs, d_gap, skipcost, maxdiff, maxgap, lc, asm_max_factor());
    hipLaunchKernelGGL(k_link_carry, dim3(1), dim3(64), 0, stq, d_job, 1, skipcost);
    VMX_TRY(download(&hj, B.job.p, 1, c->stream)); VMX_TRY(download(&hs, B.st.p, 1, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    if (!hj.ran) { set_error("vm_chain_linked: the batch did not run"); return VM_ERR_HIP; }
    out->gmax = hj.gmax; out->n_hot = hj.hot; out->n_cold = hj.n_cold; out->cold_max = hj.cold_max; out->opcount = hj.opcount;
    const int base = cap_pre - (int)n_pre;
    out->S = (double*)malloc(8 * (size_t)n); out->P = (int64_t*)malloc(8 * (size_t)n); out->S_arg_hot = (int64_t*)malloc(8 * (size_t)std::max<int64_t>(1, hj.hot));
    struct LinkedOutGuard { vm_linked_out* o; bool keep = false; ~LinkedOutGuard() { if (!keep) vm_linked_out_free(o); } } out_guard{out};      // (an error below frees what was allocated: ADVICE r3)
    std::vector<int32_t> p32((size_t)n), sa32((size_t)std::max<int64_t>(1, hj.hot));
    VMX_TRY(download(out->S, B.S.as<double>() + base, (size_t)n, c->stream)); VMX_TRY(download(p32.data(), B.P.as<int32_t>() + base, (size_t)n, c->stream));
    VMX_TRY(download(sa32.data(), B.SA.p, (size_t)hj.hot, c->stream));
    out->carry_status = hs.status; out->saved = hj.saved;
    std::vector<int32_t> cp; std::vector<vmx_anchor> cr;
    if (hs.status == 0 && hj.saved) {
        out->n_carry = hs.n_pre; out->carry_g_max_scores = hs.g_max_scores; out->carry_prereadloc = hs.prereadloc;
        out->carry_S = (double*)malloc(8 * (size_t)hs.n_pre); out->carry_P = (int64_t*)malloc(8 * (size_t)hs.n_pre); out->carry_rows = (int64_t*)malloc(32 * (size_t)hs.n_pre);
        cp.resize((size_t)hs.n_pre); cr.resize((size_t)hs.n_pre);
        VMX_TRY(download(out->carry_S, B.preS.p, (size_t)hs.n_pre, c->stream)); VMX_TRY(download(cp.data(), B.preP.p, (size_t)hs.n_pre, c->stream));
        VMX_TRY(download(cr.data(), B.preR.p, (size_t)hs.n_pre, c->stream));
    }
    VMX_HIP(vmx_stream_sync(c));
    for (int64_t i = 0; i < n; ++i) out->P[i] = p32[(size_t)i];
    for (int64_t i = 0; i < hj.hot; ++i) out->S_arg_hot[i] = sa32[(size_t)i];
    for (size_t i = 0; i < cp.size(); ++i) {
        out->carry_P[i] = cp[i];
        out->carry_rows[4 * i] = cr[i].q; out->carry_rows[4 * i + 1] = cr[i].r; out->carry_rows[4 * i + 2] = cr[i].s; out->carry_rows[4 * i + 3] = (int)cr[i].l & 0xffff;
    }
    out_guard.keep = true;
    return VM_OK;
}

// ================================================================================================ contigs of 500 kb and more
// assembly_get_readmap_DP_test (mammap_asm.py:23204-23422). The host runs the reference's LOOP — which windows form a batch (yield_mapinfo
// :22411), the traceback over the saved batches (:23277-23292), where the second round is cut (yield_second_mapinfo :22444), the overlap trim
// (:23400-23412) — and the device does the work inside it: seeding of the 100 kb windows (vm_map_batch, check_num = -1), the linked chain DPs
// and the state carried between batches (k_chain_linked / k_link_carry), the second-round 9-mer re-seeding (k_local_seed over a given read
// range = collect_second_round_anchors :22477), and ass_extend_func through the extend stage of vm_align_batch on the resulting chain.
// The reference spills every batch's (anchors, P) to <workdir>/<n>.npz; here they are kept in host memory.
#include "vmx_stage.h"
#include "vmx_local.h"
#include <memory>

__global__ void k_local_seed(vmx_lseed_args A);
struct vmx_seg_trace;
int align_device(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, const std::vector<int64_t>& h_roff,
                 vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats, vmx_seg_trace* trace, const vmx_preset* preset);

namespace {

struct LinkRound {
    // cap_pre: anchors of the slice carried between batches that the staging area holds (the reference keeps them all, mammap_asm.py:23250-23272):
    // 2^18 — a repeat-rich window under check_num = -1 carried more than the 4096 of round 3 — 7 MB per contig in flight
    vm_ctx* c = nullptr; int lc = 0, kmersize = 15, maxdiff = 50, maxgap = 1000, cap_pre = 1 << 18; double skipcost = 30., margin_base = 0.;
    DevBuf st, preS, preP, preR, rows, S, P, SA, job, gap, fSi, fT, fCNT;
    std::vector<std::vector<vmx_anchor>> saved_rows; std::vector<std::vector<int32_t>> saved_P;
    int64_t pre_g_max_index = 0; bool have = false; int cur_n_pre = 0;
    LinkRound() = default; LinkRound(const LinkRound&) = delete;
    ~LinkRound() { for (DevBuf* b : {&st, &preS, &preP, &preR, &rows, &S, &P, &SA, &job, &gap, &fSi, &fT, &fCNT}) b->release(); }
    int init(vm_ctx* ctx, int lc_, int k, double skip, int md, int mg) {
        c = ctx; lc = lc_; kmersize = k; skipcost = skip; maxdiff = md; maxgap = mg;
        if (md > 62) { set_error("maxdiff > 62 unsupported"); return VM_ERR_UNSUPPORTED; }
        const HostTables& T = host_tables();
        std::vector<double> g(64, 0.0); double gapmax = 0.0;
        for (int x = 1; x <= md; ++x) { g[x] = (0.01 * k * x + 0.5 * T.log2int[x]); gapmax = std::max(gapmax, g[x]); }
        double rgcmax = 0.0; if (lc) for (int r = 0; r < 100; ++r) rgcmax = std::max(rgcmax, (double)T.readgap_r[r]);
        margin_base = std::max(skip + 36.0, gapmax + rgcmax);
        VMX_TRY(upload(gap, g.data(), 64, c->stream));
        VMX_TRY(preS.reserve(8 * (size_t)cap_pre)); VMX_TRY(preP.reserve(4 * (size_t)cap_pre)); VMX_TRY(preR.reserve(sizeof(vmx_anchor) * (size_t)cap_pre));
        vmx_link_state hs; memset(&hs, 0, sizeof hs);
        hs.cap_pre = cap_pre; hs.pre_S = preS.as<double>(); hs.pre_P = preP.as<int32_t>(); hs.pre_rows = preR.as<vmx_anchor>();
        VMX_TRY(upload(st, &hs, 1, c->stream));
        VMX_HIP(vmx_stream_sync(c));
        return 0;
    }
    // one batch (:23230-23275 / :23329-23373), in two halves so that the batches of several contigs run in ONE launch (run_jobs below).
    // rows_new: sorted by read position (stable)
    int stage(const std::vector<vmx_anchor>& rows_new, vmx_link_job& hj) {
        const int64_t n_new = (int64_t)rows_new.size();
        const size_t tot = (size_t)cap_pre + (size_t)n_new;
        VMX_TRY(rows.reserve(sizeof(vmx_anchor) * tot)); VMX_TRY(S.reserve(8 * tot)); VMX_TRY(P.reserve(4 * tot)); VMX_TRY(SA.reserve(4 * tot));
        VMX_HIP(hipMemcpyAsync(rows.as<vmx_anchor>() + cap_pre, rows_new.data(), sizeof(vmx_anchor) * (size_t)n_new, hipMemcpyHostToDevice, c->stream));
        memset(&hj, 0, sizeof hj);
        hj.state = st.as<vmx_link_state>(); hj.rows = rows.as<vmx_anchor>(); hj.S = S.as<double>(); hj.P = P.as<int32_t>(); hj.SA = SA.as<int32_t>();
        hj.cap_pre = cap_pre; hj.n_new = (int32_t)n_new;
        return 0;
    }
    int finish(const vmx_link_job& hj, const vmx_link_state& hs, const std::vector<vmx_anchor>& rows_new) {
        if (hs.status == VM_LINK_RAISED) { set_error("asm: the reference raises on this contig (linked chain)"); return VM_READ_RAISED; }
        if (hs.status != 0) { set_error(hs.status == VM_LINK_BAILED ? "asm: GC-exact bailed out and the batch was not re-run by the linked GC-fast" : "asm: carried slice outside the stored index / staging area"); return VM_READ_UNSUPPORTED; }
        if (!hj.ran) { set_error("asm: a linked batch did not run"); return VM_ERR_HIP; }
        have = true; pre_g_max_index = hs.pre_g_max_index;
        if (hj.saved) {
            const int n_pre = cur_n_pre, base = cap_pre - n_pre, n = hj.n;
            std::vector<vmx_anchor> lr((size_t)n); std::vector<int32_t> lp((size_t)n);
            VMX_TRY(download(lr.data(), rows.as<vmx_anchor>() + base, (size_t)n_pre, c->stream));
            VMX_TRY(download(lp.data(), P.as<int32_t>() + base, (size_t)n, c->stream));
            VMX_HIP(vmx_stream_sync(c));
            std::copy(rows_new.begin(), rows_new.end(), lr.begin() + n_pre);
            saved_rows.push_back(std::move(lr)); saved_P.push_back(std::move(lp));
            cur_n_pre = hs.n_pre;
        }
        return 0;
    }
    // :23277-23292 / :23379-23396
    int traceback(int64_t start, std::vector<vmx_anchor>& path) const {
        path.clear();
        int64_t gi = start;
        for (int64_t d = (int64_t)saved_rows.size() - 1; d >= 0; --d) {
            const std::vector<vmx_anchor>& R = saved_rows[(size_t)d]; const std::vector<int32_t>& Pp = saved_P[(size_t)d];
            const int64_t n = (int64_t)R.size();
            int64_t take = gi;
            if (take < 0 || take >= n) return VM_READ_RAISED;     // IndexError
            path.push_back(R[(size_t)take]);
            while (true) {
                if (Pp[(size_t)take] < 0) break;
                take = Pp[(size_t)take];
                if (take >= n) return VM_READ_RAISED;
                path.push_back(R[(size_t)take]);
            }
            gi = std::llabs((long long)Pp[(size_t)take]);
        }
        return 0;
    }
};

// the current batch of every listed contig in one launch (one wavefront per contig): rc[i] = 0 or that contig's negative status; a negative
// return value is a failed call. All rounds share the DP's parameters (same round of the same run).
int run_jobs(vm_ctx* c, DevBuf& d_jobs, const std::vector<LinkRound*>& rounds, const std::vector<const std::vector<vmx_anchor>*>& news, std::vector<int>& rc) {
    const size_t nj = rounds.size();
    rc.assign(nj, 0);
    if (!nj) return 0;
    std::vector<vmx_link_job> hj(nj); std::vector<vmx_link_state> hs(nj);
    for (size_t i = 0; i < nj; ++i) VMX_TRY(rounds[i]->stage(*news[i], hj[i]));
    static DevBuf d_dbg; static const bool dbg_on = getenv("VMX_ASM_TIME") != nullptr;
    if (dbg_on) { if (!d_dbg.p) { VMX_TRY(d_dbg.reserve(64)); VMX_HIP(hipMemsetAsync(d_dbg.p, 0, 64, c->stream)); } for (size_t i = 0; i < nj; ++i) hj[i].dbg = d_dbg.as<long long>(); }
    VMX_TRY(upload(d_jobs, hj.data(), nj, c->stream));
    const LinkRound& R0 = *rounds[0];
    vmx_link_job* dj = d_jobs.as<vmx_link_job>(); const double* d_gap = R0.gap.as<double>(); const vmx_tables tabs = c->tables; hipStream_t stq = c->stream;
    const double sk = R0.skipcost, mb = R0.margin_base, mf = asm_max_factor(); const int md = R0.maxdiff, mg = R0.maxgap, l = R0.lc, n_jobs = (int)nj;
    hipLaunchKernelGGL(k_link_place, dim3((unsigned)nj), dim3(256), 0, stq, dj, n_jobs);
    if (asm_link_plain()) hipLaunchKernelGGL(k_chain_linked, dim3((unsigned)nj), dim3(64), 0, stq, dj, n_jobs, tabs, d_gap, sk, md, mg, l, mb, mf);
    else hipLaunchKernelGGL(k_chain_linked_win, dim3((unsigned)nj), dim3(64), 0, stq, dj, n_jobs, tabs, d_gap, sk, md, mg, l, mf);
    hipLaunchKernelGGL(k_link_carry, dim3((unsigned)nj), dim3(64), 0, stq, dj, n_jobs, sk);
    VMX_TRY(download(hj.data(), d_jobs.p, nj, c->stream));
    for (size_t i = 0; i < nj; ++i) VMX_TRY(download(&hs[i], rounds[i]->st.p, 1, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    // :23246-23247: a batch whose linked GC-exact bailed out (opcount / i > 1000: repeat arrays) is chained again with the fork's linked GC-fast
    if (!R0.lc) {
        std::vector<size_t> bailed;
        for (size_t i = 0; i < nj; ++i) if (hs[i].status == VM_LINK_BAILED && hj[i].ran && hj[i].gmax == -1) bailed.push_back(i);
        if (!bailed.empty()) {
            std::vector<vmx_link_job> fj(bailed.size());
            for (size_t b = 0; b < bailed.size(); ++b) {
                const size_t i = bailed[b]; LinkRound& R = *rounds[i];
                const size_t tot = (size_t)R.cap_pre + news[i]->size();
                const size_t cnt_n = (size_t)news[i]->back().q + 50 + 64;
                VMX_TRY(R.fSi.reserve(4 * tot)); VMX_TRY(R.fT.reserve(8 * tot)); VMX_TRY(R.fCNT.reserve(4 * cnt_n));
                hs[i].status = 0;
                VMX_HIP(hipMemcpyAsync(R.st.p, &hs[i], sizeof(vmx_link_state), hipMemcpyHostToDevice, c->stream));
                fj[b] = hj[i]; fj[b].Si = R.fSi.as<int32_t>(); fj[b].T = R.fT.as<int64_t>(); fj[b].CNT = R.fCNT.as<int32_t>();
            }
            VMX_TRY(upload(d_jobs, fj.data(), fj.size(), c->stream));
            const int nb = (int)fj.size();
            hipLaunchKernelGGL(k_chain_linked_fast, dim3((unsigned)nb), dim3(64), 0, stq, dj, nb, tabs, d_gap, sk, md, mg);
            hipLaunchKernelGGL(k_link_carry, dim3((unsigned)nb), dim3(64), 0, stq, dj, nb, sk);
            VMX_TRY(download(fj.data(), d_jobs.p, fj.size(), c->stream));
            for (size_t b = 0; b < bailed.size(); ++b) VMX_TRY(download(&hs[bailed[b]], rounds[bailed[b]]->st.p, 1, c->stream));
            VMX_HIP(vmx_stream_sync(c));
            VMX_HIP(hipGetLastError());
            for (size_t b = 0; b < bailed.size(); ++b) hj[bailed[b]] = fj[b];
        }
    }
    if (dbg_on) { long long h[8]; VMX_TRY(download(h, d_dbg.p, 8, c->stream)); VMX_HIP(vmx_stream_sync(c));
                  fprintf(stderr, "[asm] linked DP so far: %lld anchors, %lld insertions through HBM, %lld scan blocks past the window, %lld position advances, %lld candidates\n", h[0], h[1], h[2], h[3], h[4]); }
    for (size_t i = 0; i < nj; ++i) {
        const int r = rounds[i]->finish(hj[i], hs[i], *news[i]);
        if (r == VM_READ_RAISED || r == VM_READ_UNSUPPORTED || r == VM_READ_CAPACITY) rc[i] = r; else if (r < 0) return r;
    }
    return 0;
}

struct SeedBufs {
    DevBuf guide, glen, ngu, aoff, order, head, epoch, next, sq, dst, hkey, hkey2, hval, hq, goff, pcnt, pc2, stg, gkey, gq, gr, la_rows, la_ekey, la_sorted, la_off, la_cnt, status, rdoff, rdlen, rst, ren;
    SeedBufs() = default; SeedBufs(const SeedBufs&) = delete;
    ~SeedBufs() { for (DevBuf* b : {&guide, &glen, &ngu, &aoff, &order, &head, &epoch, &next, &sq, &dst, &hkey, &hkey2, &hval, &hq, &goff, &pcnt, &pc2, &stg, &gkey, &gq, &gr, &la_rows, &la_ekey, &la_sorted, &la_off, &la_cnt, &status, &rdoff, &rdlen, &rst, &ren}) b->release(); }
};

struct Tuple { int64_t st_read, en_read, lo, hi; };

// development aid (VMX_ASM_DUMP=<dir>): the stages of the long-contig loop as int64 rows (q, r, s, l), for comparison with the oracle's vmo_asm_trace
void dump_rows(const char* name, const std::vector<vmx_anchor>& v, bool append = false) {
    const char* d = getenv("VMX_ASM_DUMP"); if (!d) return;
    const std::string fn = std::string(d) + "/" + name;
    FILE* f = fopen(fn.c_str(), append ? "ab" : "wb"); if (!f) return;
    for (const vmx_anchor& a : v) { const int64_t row[4] = {a.q, a.r, a.s, (int)a.l & 0xffff}; fwrite(row, 8, 4, f); }
    fclose(f);
}

// collect_second_round_anchors (:22477-22756) for every second-round batch at once: one k_local_seed launch, one unit per batch
#define VMX_RETRY_RESEED (-9002)          // second_round_seed_once -> second_round_seed: a pool of the launch was too small for one of its batches
static int second_round_seed_once(vm_ctx* c, const vm_index_view& ix, int k, const uint8_t* d_codes, int64_t L, const std::vector<vmx_anchor>& raw_asc, const std::vector<Tuple>& tuples,
                                  std::vector<std::vector<vmx_anchor>>& out, int64_t mult, int64_t div, std::vector<int64_t>& slot_mul);
// the anchor slots and hit pools of the launch are estimates (8 x the per-read slot of the span, 4 hits per position): a batch that overflows them is not given up — the
// launch is repeated, up to five times, before the contig is reported as VM_READ_CAPACITY (round 5; VMX_TEST_ASM_RESEED_DIV=<d> divides the pools in the tests).
// Round 6: only the batches (tuples) that reported an overflow get their anchor slot x4 (a chromosome-scale contig has thousands of tuples: scaling all
// of them asked for the HBM long before x256); the per-workgroup hit / position pools are shared by all batches and grow x4 up to their clamps; a repeat that
// finds no device memory for its larger pools ends as VM_READ_CAPACITY for THIS contig, not as a failed call that takes the other contigs with it.
int second_round_seed(vm_ctx* c, const vm_index_view& ix, int k, const uint8_t* d_codes, int64_t L, const std::vector<vmx_anchor>& raw_asc, const std::vector<Tuple>& tuples,
                      std::vector<std::vector<vmx_anchor>>& out) {
    int64_t div = 1; if (const char* e = getenv("VMX_TEST_ASM_RESEED_DIV")) { const long long v = atoll(e); if (v >= 1) div = v; }
    std::vector<int64_t> slot_mul(tuples.size(), 1);          // per batch; second_round_seed_once multiplies the entries of the batches that overflowed by 4
    for (int64_t mult = 1;; mult *= 4) {
        const int rc = second_round_seed_once(c, ix, k, d_codes, L, raw_asc, tuples, out, mult, div, slot_mul);
        if (rc == VM_ERR_OOM && mult > 1) { set_error("asm: no device memory for the larger second-round re-seeding pools of this contig"); return VM_READ_CAPACITY; }
        if (rc != VMX_RETRY_RESEED) return rc;
        if (mult >= 256 * div) { set_error("asm: a second-round re-seeding batch overflowed its device pools"); return VM_READ_CAPACITY; }
        if (getenv("VMX_DBG_POOLS")) { int64_t nb = 0; for (int64_t m : slot_mul) nb += m > mult; fprintf(stderr, "[pools] asm second-round re-seeding: %lld of %zu batches overflowed, once more with their slots and the hit pools x%lld\n", (long long)nb, slot_mul.size(), (long long)(mult * 4)); }
    }
}
static int second_round_seed_once(vm_ctx* c, const vm_index_view& ix, int k, const uint8_t* d_codes, int64_t L, const std::vector<vmx_anchor>& raw_asc, const std::vector<Tuple>& tuples,
                                  std::vector<std::vector<vmx_anchor>>& out, int64_t mult, int64_t div, std::vector<int64_t>& slot_mul) {
    const int64_t nt = (int64_t)tuples.size();
    out.assign((size_t)nt, {});
    if (!nt) return 0;
    if (k < 5 || k > 11) { set_error("local k-mer size must be in [5,11]"); return VM_ERR_UNSUPPORTED; }
    SeedBufs B;
    std::vector<vmx_anchor> guide; std::vector<int32_t> glen; std::vector<int64_t> aoff((size_t)nt + 1, 0), la_off((size_t)nt + 1, 0), rdoff((size_t)nt, 0), rdlen((size_t)nt, L);
    std::vector<int32_t> ngu((size_t)nt, 1), rst((size_t)nt), ren((size_t)nt), order((size_t)nt + 1, 0);
    int64_t span_max = 1, glen_max = 1;
    for (int64_t t = 0; t < nt; ++t) {
        const Tuple& u = tuples[(size_t)t];
        if (u.hi <= u.lo) { set_error("asm: the reference raises on this contig (empty second-round guide)"); return VM_READ_RAISED; }
        aoff[(size_t)t] = (int64_t)guide.size();
        for (int64_t i = u.hi - 1; i >= u.lo; --i) guide.push_back(raw_asc[(size_t)i]);          // descending read order, like a stored path
        rst[(size_t)t] = (int32_t)u.st_read; ren[(size_t)t] = (int32_t)u.en_read;
        const int64_t span = u.en_read - u.st_read;
        span_max = std::max(span_max, span); glen_max = std::max(glen_max, u.hi - u.lo);
        la_off[(size_t)t + 1] = la_off[(size_t)t] + std::max<int64_t>(16, 8 * slot_mul[(size_t)t] * VMX_LA_SLOT(span) / div);
        order[(size_t)t + 1] = (int32_t)t;
    }
    aoff[(size_t)nt] = (int64_t)guide.size();
    glen.assign(guide.size() + 1, 0);
    for (int64_t t = 0; t < nt; ++t) glen[(size_t)aoff[(size_t)t]] = (int32_t)(tuples[(size_t)t].hi - tuples[(size_t)t].lo);
    std::stable_sort(order.begin() + 1, order.end(), [&](int32_t a, int32_t b) { return tuples[(size_t)a].en_read - tuples[(size_t)a].st_read > tuples[(size_t)b].en_read - tuples[(size_t)b].st_read; });
    const int TPB = 512;
    int occ = 1;
#ifndef VMX_EMU
    VMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_local_seed, TPB, 0));
    if (occ < 1) occ = 1;
    if (occ > 2) occ = 2;
#endif
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(nt, std::min<int64_t>(64, (int64_t)c->num_cu * occ)));
    const int64_t nkey = (int64_t)1 << (2 * k);
    const int64_t head_stride = (2 * k > 14 && nkey <= (int64_t)VMX_SORT_LDS * 64) ? ((int64_t)1 << 14) : nkey;
    // the guide of a batch spans about its read range on the reference (+ 2000 on either side of every window, :22526-22530)
    const int64_t tpos_cap = std::min<int64_t>(mult * (4 * span_max + 64 * 4000 + 65536), ((int64_t)1 << 23) - 2);
    int64_t hit_cap = 1; while (hit_cap < std::max<int64_t>(64, 4 * mult * (span_max + 14000) / div)) hit_cap <<= 1;
    if (hit_cap > ((int64_t)1 << 26)) hit_cap = (int64_t)1 << 26;
    const int64_t pcnt_cap = span_max + 16;
    int64_t gkey_cap = 1; while (gkey_cap < glen_max) gkey_cap <<= 1;
    const int64_t la_tot = la_off[(size_t)nt];
    VMX_TRY(upload(B.guide, guide.data(), guide.size(), c->stream)); VMX_TRY(upload(B.glen, glen.data(), glen.size(), c->stream)); VMX_TRY(upload(B.ngu, ngu.data(), ngu.size(), c->stream));
    VMX_TRY(upload(B.aoff, aoff.data(), aoff.size(), c->stream)); VMX_TRY(upload(B.order, order.data(), order.size(), c->stream));
    VMX_TRY(upload(B.la_off, la_off.data(), la_off.size(), c->stream)); VMX_TRY(upload(B.rdoff, rdoff.data(), rdoff.size(), c->stream)); VMX_TRY(upload(B.rdlen, rdlen.data(), rdlen.size(), c->stream));
    VMX_TRY(upload(B.rst, rst.data(), rst.size(), c->stream)); VMX_TRY(upload(B.ren, ren.data(), ren.size(), c->stream));
    VMX_TRY(B.head.reserve(4 * (size_t)G * (size_t)head_stride)); VMX_TRY(B.epoch.reserve(4 * (size_t)G + 64));
    VMX_HIP(hipMemsetAsync(B.head.p, 0xff, B.head.cap, c->stream)); VMX_HIP(hipMemsetAsync(B.epoch.p, 0, B.epoch.cap, c->stream));
    VMX_TRY(B.next.reserve(4 * (size_t)G * (size_t)tpos_cap));
    VMX_TRY(B.sq.reserve(4 * (size_t)G * (size_t)hit_cap)); VMX_TRY(B.dst.reserve(4 * (size_t)G * (size_t)hit_cap));
    VMX_TRY(B.hkey.reserve(8 * (size_t)G * (size_t)hit_cap)); VMX_TRY(B.hkey2.reserve(8 * (size_t)G * (size_t)hit_cap)); VMX_TRY(B.hval.reserve(8 * (size_t)G * (size_t)hit_cap));
    VMX_TRY(B.goff.reserve(4 * (size_t)G * (size_t)hit_cap));
    VMX_TRY(B.pcnt.reserve(4 * (size_t)G * (size_t)pcnt_cap)); VMX_TRY(B.pc2.reserve(4 * (size_t)G * (size_t)pcnt_cap)); VMX_TRY(B.stg.reserve(16 * (size_t)G * (size_t)pcnt_cap));
    VMX_TRY(B.gkey.reserve(8 * (size_t)G * (size_t)gkey_cap)); VMX_TRY(B.gq.reserve(4 * (size_t)G * (size_t)gkey_cap)); VMX_TRY(B.gr.reserve(8 * (size_t)G * (size_t)gkey_cap));
    VMX_TRY(B.la_rows.reserve(sizeof(vmx_anchor) * (size_t)(la_tot + 1))); VMX_TRY(B.la_ekey.reserve(8 * (size_t)(la_tot + 1))); VMX_TRY(B.la_sorted.reserve(sizeof(vmx_anchor) * (size_t)(la_tot + 1)));
    VMX_TRY(B.la_cnt.reserve(4 * (size_t)(nt + 1))); VMX_TRY(B.status.reserve(4 * (size_t)(nt + 1)));
    vmx_lseed_args A; memset(&A, 0, sizeof A);
    A.ocodes = d_codes; A.roff = nullptr; A.ref = ix.codes; A.coff = ix.coff; A.nseq = ix.nseq;
    A.guide_rows = B.guide.as<vmx_anchor>(); A.guide_len = B.glen.as<int32_t>(); A.n_guides_used = B.ngu.as<int32_t>(); A.aoff = B.aoff.as<int64_t>();
    A.n_reads = (int)nt; A.k = k; A.look_span = 2000; A.read_span = 500; A.sort_by_start = 1;          // :22526-22530, :22755
    A.queue = B.order.as<int32_t>(); A.order = B.order.as<int32_t>() + 1; A.la_slot_len = 1;
    A.head_pool = B.head.as<int32_t>(); A.next_pool = B.next.as<int32_t>(); A.head_stride = head_stride; A.epoch_pool = B.epoch.as<int32_t>();
    A.sq_pool = B.sq.as<int32_t>(); A.dst_pool = B.dst.as<int32_t>(); A.tpos_pool = nullptr; A.tpos_cap = tpos_cap; A.dbg = nullptr;
    A.hkey2_pool = B.hkey2.as<uint64_t>(); A.hkey_pool = B.hkey.as<uint64_t>(); A.hval_pool = B.hval.as<int64_t>(); A.goff_pool = B.goff.as<int32_t>(); A.hit_cap = hit_cap;
    A.pcnt_pool = B.pcnt.as<int32_t>(); A.pcnt_cap = pcnt_cap; A.pc2_pool = B.pc2.as<int32_t>(); A.stg_pool = B.stg.as<int64_t>();
    A.gkey_pool = B.gkey.as<uint64_t>(); A.gq_pool = B.gq.as<int32_t>(); A.gr_pool = B.gr.as<int64_t>(); A.gkey_cap = gkey_cap;
    A.la_rows = B.la_rows.as<vmx_anchor>(); A.la_ekey = B.la_ekey.as<uint64_t>(); A.la_sorted = B.la_sorted.as<vmx_anchor>(); A.la_off = B.la_off.as<int64_t>();
    A.la_cnt = B.la_cnt.as<int32_t>(); A.status = B.status.as<int32_t>();
    A.rd_off = B.rdoff.as<int64_t>(); A.rd_len = B.rdlen.as<int64_t>(); A.r_st = B.rst.as<int32_t>(); A.r_en = B.ren.as<int32_t>();
    hipStream_t stq = c->stream;
    hipLaunchKernelGGL(k_local_seed, dim3((unsigned)G), dim3(TPB), 0, stq, A);
    std::vector<int32_t> cnt((size_t)nt), stt((size_t)nt);
    VMX_TRY(download(cnt.data(), B.la_cnt.p, (size_t)nt, c->stream)); VMX_TRY(download(stt.data(), B.status.p, (size_t)nt, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    bool again = false;
    for (int64_t t = 0; t < nt; ++t) if (stt[(size_t)t] != 0) { slot_mul[(size_t)t] *= 4; again = true; }
    if (again) return VMX_RETRY_RESEED;
    for (int64_t t = 0; t < nt; ++t) {
        out[(size_t)t].resize((size_t)cnt[(size_t)t]);
        VMX_TRY(download(out[(size_t)t].data(), B.la_sorted.as<vmx_anchor>() + la_off[(size_t)t], (size_t)cnt[(size_t)t], c->stream));
    }
    VMX_HIP(vmx_stream_sync(c));
    return 0;
}

}  // namespace

namespace {

// one_mapinfo[np.argsort(one_mapinfo[:, 0])] (:22431, stable) on the device: radix sort of (read position, index) pairs, gather, back to the host copy
struct SortBufs { DevBuf in, out, k0, k1, v0, v1, tmp; SortBufs() = default; SortBufs(const SortBufs&) = delete; ~SortBufs() { for (DevBuf* b : {&in, &out, &k0, &k1, &v0, &v1, &tmp}) b->release(); } };
int device_sort_by_q(vm_ctx* c, SortBufs& B, std::vector<vmx_anchor>& rows) {
    const size_t n = rows.size();
    if (n < 2) return 0;
    VMX_TRY(upload(B.in, rows.data(), n, c->stream)); VMX_TRY(B.out.reserve(sizeof(vmx_anchor) * n));
    VMX_TRY(B.k0.reserve(8 * n)); VMX_TRY(B.k1.reserve(8 * n)); VMX_TRY(B.v0.reserve(8 * n)); VMX_TRY(B.v1.reserve(8 * n));
    const vmx_anchor* d_in = B.in.as<vmx_anchor>(); vmx_anchor* d_out = B.out.as<vmx_anchor>();
    uint64_t *k0 = B.k0.as<uint64_t>(), *k1 = B.k1.as<uint64_t>(), *v0 = B.v0.as<uint64_t>(), *v1 = B.v1.as<uint64_t>();
    const int64_t nn = (int64_t)n; hipStream_t stq = c->stream;
    const unsigned grid = (unsigned)std::min<int64_t>((nn + 255) / 256, 4096);
    hipLaunchKernelGGL(k_link_sort_keys, dim3(grid), dim3(256), 0, stq, d_in, nn, k0, v0);
    size_t tb = 0;
    VMX_HIP((hipError_t)vmx_prim_sort_pairs_u64(nullptr, &tb, k0, k1, v0, v1, n, 32, c->stream));
    VMX_TRY(B.tmp.reserve(tb + 64));
    VMX_HIP((hipError_t)vmx_prim_sort_pairs_u64(B.tmp.p, &tb, k0, k1, v0, v1, n, 32, c->stream));
    hipLaunchKernelGGL(k_link_sort_gather, dim3(grid), dim3(256), 0, stq, d_in, (const uint64_t*)v1, nn, d_out);
    VMX_TRY(download(rows.data(), B.out.p, n, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    return 0;
}

struct LongContig {
    const char* src = nullptr; int64_t len = 0;
    std::string seq; int status = 0; bool done = false;           // done: finished (records or a status), nothing more to run
    LinkRound r1, r2;
    std::vector<std::vector<vmx_anchor>> batches;                 // first-round batches (yield_mapinfo), then the second round's
    std::vector<vmx_anchor> path, raw, path2;
    std::vector<Tuple> tuples;
    DevBuf d_codes;
    vm_record* recs = nullptr; int64_t n_recs = 0; char* blob = nullptr;
    LongContig() = default; LongContig(const LongContig&) = delete;
    ~LongContig() { d_codes.release(); }
    void fail(int st) { status = st; done = true; }
};

// assembly_get_readmap_DP_test (:23208-23422) for a group of contigs side by side: whatever is serial inside a contig — the linked chain DPs,
// batch after batch — runs for all of them in one launch per batch index
int asm_long_group(vm_ctx* c, const vm_index* mi, const vm_params* prm, std::vector<LongContig*>& G, int64_t batch_anchors, int64_t window) {
    vm_index_view ix; vmx_index_view(mi, &ix);
    static const bool timing = getenv("VMX_ASM_TIME") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now_s(), t_seed = 0, t_dp1 = 0, t_seed2 = 0, t_dp2 = 0, t_ext = 0; int64_t n_a1 = 0, n_a2 = 0, bases = 0;
    auto lap = [&](double& acc) { const double t = now_s(); acc += t - t_mark; t_mark = t; };
    DevBuf d_jobs; struct RelJ { DevBuf* a; ~RelJ() { a->release(); } } relj{&d_jobs};
    SortBufs sortb;
    // ---- first round :23214-23292: the batches of every contig (yield_mapinfo :22411-22443)
    for (LongContig* Cn : G) {
        LongContig& C = *Cn;
        C.seq.assign(C.src, (size_t)C.len);
        for (char& ch : C.seq) if (ch >= 'a' && ch <= 'z') ch -= 32;
        bases += C.len;
        VMX_TRY(C.r1.init(c, 0, ix.k, prm->global_skipcost, prm->global_maxdiff, 1000));
        std::vector<std::vector<vmx_anchor>> cache; int64_t cache_size = 0;
        std::vector<vmx_anchor> one;
        auto emit = [&](std::vector<vmx_anchor>& batch) -> int {
            VMX_TRY(device_sort_by_q(c, sortb, batch));
            n_a1 += (int64_t)batch.size();
            if (!batch.empty()) C.batches.push_back(batch);        // :23231 an empty pack is skipped
            return 0;
        };
        const int64_t n_win = (C.len + window - 1) / window;
        const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(256, ((int64_t)32 << 20) / window));
        for (int64_t w0 = 0; w0 < n_win; w0 += chunk) {
            const int64_t w1 = std::min(n_win, w0 + chunk);
            std::vector<int64_t> woff((size_t)(w1 - w0) + 1);
            for (int64_t w = w0; w <= w1; ++w) woff[(size_t)(w - w0)] = std::min(w * window, C.len) - w0 * window;
            int64_t* anchors = nullptr; int64_t* aoff = nullptr;
            VMX_TRY(vm_map_batch(c, mi, -1, -1, w1 - w0, C.seq.data() + w0 * window, woff.data(), &anchors, &aoff));      // :22419
            for (int64_t w = w0; w < w1; ++w) {
                const int64_t st = w * window;
                one.clear();
                for (int64_t i = aoff[w - w0]; i < aoff[w - w0 + 1]; ++i) { vmx_anchor a; a.q = (int32_t)(anchors[4 * i] + st); a.r = anchors[4 * i + 1]; a.s = (int16_t)anchors[4 * i + 2]; a.l = (int16_t)anchors[4 * i + 3]; one.push_back(a); }
                if ((int64_t)one.size() + cache_size > batch_anchors) {          // :22423-22438
                    if (cache_size > 0) {
                        if (!one.empty()) cache.push_back(one);
                        std::vector<vmx_anchor> all; for (auto& cc : cache) all.insert(all.end(), cc.begin(), cc.end());
                        one.swap(all); cache_size = 0; cache.clear();
                    }
                    std::vector<vmx_anchor> batch = one;
                    { const int rcs = emit(batch); if (rcs < 0) { free(anchors); free(aoff); return rcs; } }
                } else if (!one.empty()) { cache.push_back(one); cache_size += (int64_t)one.size(); }
            }
            free(anchors); free(aoff);
        }
        if (cache_size > 0) {                                    // :22439-22443, including the second copy of the last window's anchors
            if (!one.empty()) cache.push_back(one);
            std::vector<vmx_anchor> all; for (auto& cc : cache) all.insert(all.end(), cc.begin(), cc.end());
            VMX_TRY(emit(all));
        }
    }
    lap(t_seed);
    auto run_round = [&](bool second) -> int {                   // batch b of every contig that has one, b = 0, 1, ...
        for (size_t b = 0;; ++b) {
            std::vector<LinkRound*> rounds; std::vector<const std::vector<vmx_anchor>*> news; std::vector<LongContig*> who;
            for (LongContig* Cn : G) if (!Cn->done && b < Cn->batches.size()) { rounds.push_back(second ? &Cn->r2 : &Cn->r1); news.push_back(&Cn->batches[b]); who.push_back(Cn); }
            if (rounds.empty()) break;
            std::vector<int> rc;
            VMX_TRY(run_jobs(c, d_jobs, rounds, news, rc));
            for (size_t i = 0; i < who.size(); ++i) if (rc[i] < 0) who[i]->fail(rc[i]);
        }
        return 0;
    };
    VMX_TRY(run_round(false));
    for (LongContig* Cn : G) {
        LongContig& C = *Cn;
        if (C.done) continue;
        C.batches.clear(); C.batches.shrink_to_fit();
        if (!C.r1.have) { C.fail(VM_READ_RAISED); continue; }     // NameError: pre_g_max_index (:23278)
        const int rc = C.r1.traceback(C.r1.pre_g_max_index, C.path);
        if (rc < 0) { C.fail(rc); continue; }
        dump_rows("path1.bin", C.path);
        if (C.path.size() <= 1) { C.done = true; continue; }
        C.r1.saved_rows.clear(); C.r1.saved_rows.shrink_to_fit(); C.r1.saved_P.clear(); C.r1.saved_P.shrink_to_fit();
    }
    lap(t_dp1);
    // ---- second round :23309-23396
    const int k2 = prm->local_kmersize;
    for (LongContig* Cn : G) {
        LongContig& C = *Cn;
        if (C.done) continue;
        VMX_TRY(C.r2.init(c, 1, k2, prm->local_skipcost, prm->local_maxdiff, 99));
        {
            DevBuf d_raw; struct Rel { DevBuf* a; ~Rel() { a->release(); } } rel{&d_raw};
            VMX_TRY(upload(d_raw, C.seq.data(), (size_t)C.len, c->stream)); VMX_TRY(C.d_codes.reserve((size_t)C.len + 64));
            const char* dr = d_raw.as<char>(); uint8_t* dc = C.d_codes.as<uint8_t>(); const int64_t nn = C.len; hipStream_t stq = c->stream;
            hipLaunchKernelGGL(k_encode, dim3((unsigned)std::min<int64_t>((nn + 255) / 256, 65535)), dim3(256), 0, stq, dr, dc, nn);
            VMX_HIP(vmx_stream_sync(c));
        }
        C.raw.assign(C.path.rbegin(), C.path.rend());
        const std::vector<vmx_anchor>& raw = C.raw;
        const int64_t np_ = (int64_t)raw.size();
        int64_t st_read = 0, st_path = 0, iloc_path = 0;         // yield_second_mapinfo :22444-22476
        for (int64_t x = 1; x < np_; ++x) {
            const vmx_anchor& now = raw[(size_t)x];
            iloc_path += 1;
            if (iloc_path == np_ - 1 || (iloc_path < np_ - 1 && raw[(size_t)iloc_path + 1].q > raw[(size_t)iloc_path].q)) {
                if (((int64_t)now.q + ((int)now.l & 0xffff)) > (st_read + window) && (iloc_path - st_path) > 300) {
                    const int64_t en_read = raw[(size_t)iloc_path].q;
                    C.tuples.push_back(Tuple{st_read, en_read, std::max<int64_t>(0, st_path - 20), std::min<int64_t>(iloc_path + 20, np_)});
                    st_path = iloc_path + 1;
                    st_read = en_read;
                }
            }
        }
        if (st_read < C.len) C.tuples.push_back(Tuple{st_read, C.len, std::max<int64_t>(0, st_path - 20), std::min<int64_t>(iloc_path + 20, np_)});
        const int rc = second_round_seed(c, ix, k2, C.d_codes.as<uint8_t>(), C.len, raw, C.tuples, C.batches);
        if (rc == VM_READ_RAISED || rc == VM_READ_UNSUPPORTED || rc == VM_READ_CAPACITY) { C.fail(rc); continue; }
        if (rc < 0) return rc;
        for (size_t t = 0; t < C.batches.size(); ++t) {
            dump_rows("second.bin", C.batches[t], t > 0);
            n_a2 += (int64_t)C.batches[t].size();
            if (C.batches[t].empty()) { C.fail(VM_READ_RAISED); break; }        // np.array([])[:, 0] (:22755)
        }
    }
    lap(t_seed2);
    VMX_TRY(run_round(true));
    lap(t_dp2);
    // ---- traceback, overlap trim (:23400-23412), ass_extend_func (:23414) for all contigs of the group in one pass of the extend stage
    std::vector<LongContig*> X;
    for (LongContig* Cn : G) {
        LongContig& C = *Cn;
        if (C.done) continue;
        const int rc = C.r2.traceback(C.r2.have ? C.r2.pre_g_max_index : C.r1.pre_g_max_index, C.path2);
        if (rc < 0) { C.fail(rc); continue; }
        if (C.path2.size() <= 1) { C.done = true; continue; }
        vmx_anchor pre = C.path2[0];
        for (size_t x = 1; x < C.path2.size(); ++x) {            // against the UNtrimmed neighbour
            const vmx_anchor now = C.path2[x];
            const int nl = (int)now.l & 0xffff;
            if (!(pre.q >= now.q + nl)) {
                vmx_anchor t = now; t.l = (int16_t)(pre.q - now.q);
                if (now.s != 1) t.r = now.r + nl - pre.q + now.q;
                C.path2[x] = t;
            }
            pre = now;
        }
        dump_rows("path2.bin", C.path2);
        X.push_back(Cn);
    }
    if (!X.empty()) {
        const int64_t nx = (int64_t)X.size();
        std::vector<int64_t> h_off((size_t)nx + 1, 0);
        for (int64_t i = 0; i < nx; ++i) h_off[(size_t)i + 1] = h_off[(size_t)i] + X[(size_t)i]->len;
        DevBuf d_off, d_all; struct Rel1 { DevBuf *a, *b; ~Rel1() { a->release(); b->release(); } } rel1{&d_off, &d_all};
        VMX_TRY(upload(d_off, h_off.data(), (size_t)nx + 1, c->stream));
        const uint8_t* codes = X[0]->d_codes.as<uint8_t>();
        if (nx > 1) {
            VMX_TRY(d_all.reserve((size_t)h_off[(size_t)nx] + 64));
            for (int64_t i = 0; i < nx; ++i) VMX_HIP(hipMemcpyAsync(d_all.as<uint8_t>() + h_off[(size_t)i], X[(size_t)i]->d_codes.p, (size_t)X[(size_t)i]->len, hipMemcpyDeviceToDevice, c->stream));
            VMX_HIP(vmx_stream_sync(c));
            for (int64_t i = 0; i < nx; ++i) X[(size_t)i]->d_codes.release();
            codes = d_all.as<uint8_t>();
        }
        std::vector<vmx_preset> ps((size_t)nx);
        for (int64_t i = 0; i < nx; ++i) { ps[(size_t)i].chain_desc = X[(size_t)i]->path2.data(); ps[(size_t)i].len = (int64_t)X[(size_t)i]->path2.size(); }     // descending read order, as the extend stage takes a local chain
        vm_record* r0 = nullptr; int64_t nr = 0; char* cb = nullptr; std::vector<int32_t> st((size_t)nx, 0);
        const int rce = align_device(c, mi, prm, nx, codes, d_off.as<int64_t>(), h_off, &r0, &nr, &cb, st.data(), nullptr, nullptr, ps.data());
        if (rce < 0) { free(r0); free(cb); return rce; }
        for (int64_t i = 0; i < nx; ++i) {
            LongContig& C = *X[(size_t)i];
            int64_t cnt = 0; size_t bytes = 0;
            for (int64_t x = 0; x < nr; ++x) if (r0[x].read_idx == (int32_t)i) { ++cnt; bytes += (size_t)r0[x].cigar_len + 1; }
            C.recs = (vm_record*)malloc(sizeof(vm_record) * (size_t)std::max<int64_t>(cnt, 1)); C.blob = (char*)malloc(std::max<size_t>(bytes, 1));
            int64_t w = 0; size_t bo = 0;
            for (int64_t x = 0; x < nr; ++x) if (r0[x].read_idx == (int32_t)i) {
                vm_record y = r0[x]; y.read_idx = 0; memcpy(C.blob + bo, cb + r0[x].cigar_off, (size_t)r0[x].cigar_len); C.blob[bo + (size_t)r0[x].cigar_len] = 0;
                y.cigar_off = (int64_t)bo; bo += (size_t)r0[x].cigar_len + 1; C.recs[w++] = y;
            }
            C.n_recs = cnt; C.status = st[(size_t)i]; C.done = true;
            C.d_codes.release();
        }
        free(r0); free(cb);
    }
    lap(t_ext);
    if (timing) fprintf(stderr, "[asm] %zu contigs, %lld bases: windows+seed %.2f s, linked GC %.2f s (%lld anchors), re-seed %.2f s, linked LC %.2f s (%lld anchors), traceback+extend %.2f s\n",
                        G.size(), (long long)bases, t_seed, t_dp1, (long long)n_a1, t_seed2, t_dp2, (long long)n_a2, t_ext);
    return 0;
}

}  // namespace

extern "C" int vm_align_asm(vm_ctx* c, const vm_index* mi, const vm_params* prm_in, const char* contig, int64_t len, int64_t split_len, int64_t batch_anchors, int64_t window,
                            vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (!prm_in || prm_in->mode != VM_MODE_ASM || !contig || len <= 0 || !recs || !n_recs || !cigar_blob || !status) { set_error("vm_align_asm: bad arguments"); return VM_ERR_ARG; }
    if (len >= ((int64_t)1 << 31) - 64) { set_error("vm_align_asm: contig of 2^31 bases or more"); return VM_ERR_UNSUPPORTED; }
    if (split_len <= 0) split_len = 500000;
    if (batch_anchors <= 0) batch_anchors = 500000;
    if (window <= 0) window = 100000;
    *recs = nullptr; *n_recs = 0; *cigar_blob = nullptr; *status = 0;
    VMX_HIP(hipSetDevice(c->device));
    if (len < split_len) {                                       // :23205: the fork's per-read function
        if (len >= 500000) { set_error("vm_align_asm: split_len above the reference's 500000"); return VM_ERR_ARG; }
        const int64_t off1[2] = {0, len};
        return vm_align_batch(c, mi, prm_in, 1, contig, off1, recs, n_recs, cigar_blob, status, nullptr);
    }
    LongContig C; C.src = contig; C.len = len;
    std::vector<LongContig*> G(1, &C);
    VMX_TRY(asm_long_group(c, mi, prm_in, G, batch_anchors, window));
    if (C.recs) { *recs = C.recs; *n_recs = C.n_recs; *cigar_blob = C.blob; }
    else { *recs = (vm_record*)malloc(sizeof(vm_record)); *cigar_blob = (char*)malloc(1); *n_recs = 0; }
    *status = C.status;
    return VM_OK;
}

// vm_align_batch in VM_MODE_ASM with contigs of 500 kb and more in the batch: the shorter ones go through the batched path together, every long
// one through vm_align_asm; records come back in contig order like any other batch
int vmx_align_batch_asm_mixed(vm_ctx* c, const vm_index* mi, const vm_params* prm, int64_t n, const char* seqs, const int64_t* offsets, vm_record** recs, int64_t* n_recs,
                              char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats) {
    *recs = nullptr; *n_recs = 0; *cigar_blob = nullptr;
    std::vector<int64_t> shorts, longs;
    for (int64_t r = 0; r < n; ++r) ((offsets[r + 1] - offsets[r] >= 500000) ? longs : shorts).push_back(r);
    std::vector<std::vector<vm_record>> per((size_t)n); std::vector<std::string> pblob((size_t)n);
    std::vector<int32_t> st((size_t)n, 0);
    vm_batch_stats tot; memset(&tot, 0, sizeof tot);
    if (!shorts.empty()) {
        std::string cat; std::vector<int64_t> off(1, 0);
        for (int64_t r : shorts) { cat.append(seqs + offsets[r], (size_t)(offsets[r + 1] - offsets[r])); off.push_back((int64_t)cat.size()); }
        vm_record* r0 = nullptr; int64_t nr = 0; char* cb = nullptr; std::vector<int32_t> s0(shorts.size());
        const int rc = vm_align_batch(c, mi, prm, (int64_t)shorts.size(), cat.data(), off.data(), &r0, &nr, &cb, s0.data(), &tot);
        if (rc < 0) { free(r0); free(cb); return rc; }
        for (int64_t i = 0; i < nr; ++i) { const int64_t r = shorts[(size_t)r0[i].read_idx]; vm_record x = r0[i]; x.cigar_off = (int64_t)pblob[(size_t)r].size(); pblob[(size_t)r].append(cb + r0[i].cigar_off, (size_t)r0[i].cigar_len); pblob[(size_t)r].push_back('\0'); per[(size_t)r].push_back(x); }
        for (size_t i = 0; i < shorts.size(); ++i) st[(size_t)shorts[i]] = s0[i];
        free(r0); free(cb);
    }
    // the long contigs side by side, in groups of at most 400 Mbases (their anchors wait in host memory between the rounds)
    for (size_t g0 = 0; g0 < longs.size();) {
        size_t g1 = g0; int64_t gb = 0;
        while (g1 < longs.size() && (g1 == g0 || gb + (offsets[longs[g1] + 1] - offsets[longs[g1]]) <= (int64_t)400000000)) { gb += offsets[longs[g1] + 1] - offsets[longs[g1]]; ++g1; }
        std::vector<std::unique_ptr<LongContig>> own; std::vector<LongContig*> G;
        for (size_t i = g0; i < g1; ++i) {
            const int64_t r = longs[i];
            if (offsets[r + 1] - offsets[r] >= ((int64_t)1 << 31) - 64) { set_error("asm: contig of 2^31 bases or more"); return VM_ERR_UNSUPPORTED; }
            own.emplace_back(new LongContig()); own.back()->src = seqs + offsets[r]; own.back()->len = offsets[r + 1] - offsets[r]; G.push_back(own.back().get());
        }
        const int rc = asm_long_group(c, mi, prm, G, 500000, 100000);
        for (size_t i = g0; i < g1; ++i) {
            const int64_t r = longs[i]; LongContig& C = *G[i - g0];
            if (rc >= 0) {
                for (int64_t x = 0; x < C.n_recs; ++x) { vm_record y = C.recs[x]; y.cigar_off = (int64_t)pblob[(size_t)r].size(); pblob[(size_t)r].append(C.blob + C.recs[x].cigar_off, (size_t)C.recs[x].cigar_len); pblob[(size_t)r].push_back('\0'); per[(size_t)r].push_back(y); }
                st[(size_t)r] = C.status; tot.n_reads += 1; tot.read_bases += C.len; tot.n_records += C.n_recs; if (C.status != 0) tot.n_failed += 1;
            }
            free(C.recs); free(C.blob);
        }
        if (rc < 0) return rc;
        g0 = g1;
    }
    std::vector<vm_record> all; std::string blob;
    for (int64_t r = 0; r < n; ++r) for (vm_record x : per[(size_t)r]) { x.read_idx = (int32_t)r; const int64_t o = x.cigar_off; x.cigar_off = (int64_t)blob.size(); blob.append(pblob[(size_t)r].data() + o, (size_t)x.cigar_len); blob.push_back('\0'); all.push_back(x); }
    *recs = (vm_record*)malloc(sizeof(vm_record) * std::max<size_t>(all.size(), 1)); *cigar_blob = (char*)malloc(std::max<size_t>(blob.size(), 1));
    if (!*recs || !*cigar_blob) { free(*recs); free(*cigar_blob); *recs = nullptr; *cigar_blob = nullptr; set_error("out of host memory"); return VM_ERR_OOM; }
    memcpy(*recs, all.data(), sizeof(vm_record) * all.size()); memcpy(*cigar_blob, blob.data(), blob.size());
    *n_recs = (int64_t)all.size();
    if (status_per_read) for (int64_t r = 0; r < n; ++r) status_per_read[r] = st[(size_t)r];
    if (stats) *stats = tot;
    return VM_OK;
}

// ================================================================================================ decode_hit's tie-break (mammap_asm.py:21302-21326)
// When MAPQ is 0 and the primary's group holds chains within 0.1 % of its score, the fork picks, among them, the chain whose longest co-linear
// stretch (return_main_alignment_size :21244) is least divergent (edlib). k_chain_select marks such a contig (n_paths = -7) and leaves the choice
// to this host step: it repeats the peel and the grouping on the contig's S / P / S_arg (vmx_select.h, the code the kernel runs), computes the
// candidates' edit distances on the device (vm_edit_distance_batch) and writes the chosen path back. Rare: equal-score placements of a contig.
#include "vmx_select.h"

namespace {
std::string py_slice(const std::string& s, int64_t a, int64_t b) {
    const int64_t n = (int64_t)s.size();
    if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
    if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
    return b <= a ? std::string() : s.substr((size_t)a, (size_t)(b - a));
}
int p2c_host(const std::vector<int64_t>& coff, int64_t pos) { int pre = 0; for (size_t ci = 0; ci + 1 < coff.size(); ++ci) { if (pos < coff[ci]) break; pre = (int)ci; } return pre; }
}

// status_override[r]: 0 = resolved (or not a tie), VM_READ_RAISED where the reference raises (division by zero, :21318)
int vmx_asm_resolve_ties(vm_ctx* c, const vm_index* mi, int64_t n, const std::vector<int64_t>& h_roff, const std::vector<int64_t>& h_aoff, const uint8_t* d_codes,
                         const vmx_anchor* d_sorted, const double* d_S, const int32_t* d_P, const int32_t* d_SA, const int64_t* d_gmax, const int32_t* d_flip,
                         int32_t* d_mapq, double* d_gscore, int32_t* d_np, int32_t* d_plen, vmx_anchor* d_prow, std::vector<int32_t>& status_override) {
    status_override.assign((size_t)n, 0);
    std::vector<int32_t> h_np((size_t)n);
    VMX_TRY(download(h_np.data(), d_np, (size_t)n, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    std::vector<int64_t> coff; { const int ns = vm_index_nseq(mi); coff.resize((size_t)ns + 1); for (int i = 0; i < ns; ++i) { const char* nm; int64_t ln, of; vm_index_seq_info(mi, i, &nm, &ln, &of); coff[(size_t)i] = of; coff[(size_t)i + 1] = of + ln; } }
    for (int64_t r = 0; r < n; ++r) {
        if (h_np[(size_t)r] != -7) continue;
        const int64_t a0 = h_aoff[(size_t)r]; const int m = (int)(h_aoff[(size_t)r + 1] - a0); const int64_t L = h_roff[(size_t)r + 1] - h_roff[(size_t)r];
        std::vector<vmx_anchor> A((size_t)m); std::vector<double> S((size_t)m); std::vector<int32_t> P((size_t)m), SA((size_t)m); int64_t gmax = 0; int32_t flip = 0;
        std::vector<uint8_t> codes((size_t)L);
        VMX_TRY(download(A.data(), d_sorted + a0, (size_t)m, c->stream)); VMX_TRY(download(S.data(), d_S + a0, (size_t)m, c->stream));
        VMX_TRY(download(P.data(), d_P + a0, (size_t)m, c->stream)); VMX_TRY(download(SA.data(), d_SA + a0, (size_t)m, c->stream));
        VMX_TRY(download(&gmax, d_gmax + r, 1, c->stream)); VMX_TRY(download(&flip, d_flip + r, 1, c->stream));
        VMX_TRY(download(codes.data(), d_codes + h_roff[(size_t)r], (size_t)L, c->stream));
        VMX_HIP(vmx_stream_sync(c));
        // the peel and the grouping, as k_chain_select ran them
        std::vector<char> scratch((size_t)vmx_select_scratch_bytes(m) + 64);
        vmx_select_scr W = vmx_select_scratch(scratch.data(), m);
        std::vector<unsigned char> used((size_t)m, 0);
        const int nch = vmx_select_peel(m, S.data(), P.data(), SA.data(), (int)gmax, 4, used.data(), W.cscore, W.coff, W.cidx);
        if (nch <= 0) continue;                                  // (cannot happen for a marked contig)
        std::vector<int> order((size_t)nch);
        for (int ci = 0; ci < nch; ++ci) { int pos = 0; while (pos < ci && W.cscore[order[(size_t)pos]] > W.cscore[ci]) ++pos; for (int t = ci; t > pos; --t) order[(size_t)t] = order[(size_t)t - 1]; order[(size_t)pos] = ci; }
        auto bins_of = [&](int ch) { std::vector<int> b; int last = -1; for (int t = W.coff[ch]; t < W.coff[ch + 1]; ++t) { const int v = A[(size_t)W.cidx[t]].q / 100; if (v != last) { b.push_back(v); last = v; } } return b; };
        std::vector<std::vector<int>> prim_bins; std::vector<int> group0;
        prim_bins.push_back(bins_of(order[0])); group0.push_back(order[0]);
        for (int oi = 1; oi < nch; ++oi) {
            const int ch = order[(size_t)oi]; const std::vector<int> b = bins_of(ch);
            double maxov = 0.0; size_t prefer = 0;
            for (size_t p = 0; p < prim_bins.size(); ++p) {
                const int inter = vmx_desc_intersect(b.data(), (int)b.size(), prim_bins[p].data(), (int)prim_bins[p].size());
                const double ov = (double)inter / (double)std::min(b.size(), prim_bins[p].size());
                if (ov > maxov) { maxov = ov; prefer = p; }
            }
            if (maxov < 0.5) prim_bins.push_back(b); else if (prefer == 0) group0.push_back(ch);
        }
        // :21305-21326
        std::string fwd((size_t)L, 'N'); for (int64_t i = 0; i < L; ++i) fwd[(size_t)i] = "ACGTN"[codes[(size_t)i] > 4 ? 4 : codes[(size_t)i]];
        std::string rcs((size_t)L, 'N'); for (int64_t i = 0; i < L; ++i) { const uint8_t cc = codes[(size_t)(L - 1 - i)]; rcs[(size_t)i] = cc < 4 ? "TGCA"[cc] : 'N'; }
        const std::string& rd = flip ? rcs : fwd; const std::string& rc = flip ? fwd : rcs;
        const double base_score = W.cscore[group0[0]];
        std::vector<int> cand; std::vector<std::string> qs, ts;
        bool raised = false;
        for (size_t t = 0; t < group0.size(); ++t) {
            const int ch = group0[t];
            if (W.cscore[ch] / base_score < 0.999) break;
            // return_main_alignment_size on the path in ascending read order
            std::vector<vmx_anchor> asc; for (int x = W.coff[ch + 1] - 1; x >= W.coff[ch]; --x) asc.push_back(A[(size_t)W.cidx[x]]);
            vmx_anchor pre = asc[0], st_item = pre, best_st = pre, best_en = pre; int64_t size = 0;
            for (size_t x = 1; x < asc.size(); ++x) {
                const vmx_anchor now = asc[x];
                if (pre.s == now.s) {
                    const int64_t readgap = (int64_t)now.q - pre.q - ((int)pre.l & 0xffff);
                    if (readgap < 0) continue;
                    const int64_t refgap = pre.s == 1 ? now.r - pre.r - ((int)pre.l & 0xffff) : pre.r - now.r - ((int)now.l & 0xffff);
                    if (std::llabs(readgap - refgap) <= 30 && refgap >= 0 && p2c_host(coff, pre.r) == p2c_host(coff, now.r)) { pre = now; continue; }
                }
                if (pre.q - st_item.q > size) { size = pre.q - st_item.q; best_st = st_item; best_en = pre; }
                pre = now; st_item = pre;
            }
            if (pre.q - st_item.q > size) { best_st = st_item; best_en = pre; }
            const vmx_anchor p0 = best_st, p1 = best_en;
            if (p0.s != p1.s || p0.q == p1.q) continue;
            std::string query, target;
            if (p0.s == 1) {
                const int ci = p2c_host(coff, p0.r); const int64_t bias = coff[(size_t)ci];
                query = py_slice(rd, p0.q, p1.q);
                const int64_t ta = std::max<int64_t>(0, p0.r - bias), tb = std::min<int64_t>(coff[(size_t)ci + 1] - bias, p1.r - bias);
                if (tb > ta) { target.resize((size_t)(tb - ta)); vm_index_seq(mi, ci, ta, tb, &target[0]); }
            } else {
                const int ci = p2c_host(coff, p1.r); const int64_t bias = coff[(size_t)ci];
                query = py_slice(rc, L - p1.q, L - p0.q);
                const int64_t ta = std::max<int64_t>(0, p1.r + ((int)p1.l & 0xffff) - bias), tb = std::min<int64_t>(coff[(size_t)ci + 1] - bias, p0.r + ((int)p0.l & 0xffff) - bias);
                if (tb > ta) { target.resize((size_t)(tb - ta)); vm_index_seq(mi, ci, ta, tb, &target[0]); }
            }
            if (std::min(query.size(), target.size()) == 0) { raised = true; break; }           // ZeroDivisionError (:21318)
            cand.push_back(ch); qs.push_back(query); ts.push_back(target);
        }
        if (raised) { status_override[(size_t)r] = VM_READ_RAISED; const int32_t zero = 0; VMX_HIP(hipMemcpyAsync(d_np + r, &zero, 4, hipMemcpyHostToDevice, c->stream)); VMX_HIP(vmx_stream_sync(c)); continue; }
        int base = group0[0];
        if (!cand.empty()) {
            std::string qcat, tcat; std::vector<int64_t> qo(1, 0), to(1, 0);
            for (size_t i = 0; i < cand.size(); ++i) { qcat += qs[i]; tcat += ts[i]; qo.push_back((int64_t)qcat.size()); to.push_back((int64_t)tcat.size()); }
            int64_t* dist = nullptr;
            VMX_TRY(vm_edit_distance_batch(c, (int64_t)cand.size(), qcat.data(), qo.data(), tcat.data(), to.data(), &dist));
            double min_diff = 10;
            for (size_t i = 0; i < cand.size(); ++i) {
                const double diff = (double)dist[i] / (double)std::min(qs[i].size(), ts[i].size());
                if (diff <= min_diff) { min_diff = diff; base = cand[i]; }
            }
            free(dist);
        }
        // write the choice back: MAPQ 0, signed score, one path (descending read order)
        std::vector<vmx_anchor> path; for (int x = W.coff[base]; x < W.coff[base + 1]; ++x) path.push_back(A[(size_t)W.cidx[x]]);
        const int32_t mq = 0, one = 1, pl = (int32_t)path.size(); const double sc = flip ? -W.cscore[base] : W.cscore[base];
        VMX_HIP(hipMemcpyAsync(d_mapq + r, &mq, 4, hipMemcpyHostToDevice, c->stream)); VMX_HIP(hipMemcpyAsync(d_gscore + r, &sc, 8, hipMemcpyHostToDevice, c->stream));
        VMX_HIP(hipMemcpyAsync(d_np + r, &one, 4, hipMemcpyHostToDevice, c->stream)); VMX_HIP(hipMemcpyAsync(d_plen + a0, &pl, 4, hipMemcpyHostToDevice, c->stream));
        VMX_HIP(hipMemcpyAsync(d_prow + a0, path.data(), sizeof(vmx_anchor) * path.size(), hipMemcpyHostToDevice, c->stream));
        VMX_HIP(vmx_stream_sync(c));
    }
    return 0;
}
