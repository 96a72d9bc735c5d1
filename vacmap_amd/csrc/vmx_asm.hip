// vmx_asm.hip — host side of -mode asm's batch-linked chain DPs (k_chain_linked.hip): the stage entry vm_chain_linked.
// The per-read function of the fork (contigs below 500 kb) runs through vm_align_batch with VM_MODE_ASM (vmx_align.hip).
#include "vmx_host.h"
#include "vmx_link.h"
#include <cstring>

using namespace vmx;

__global__ void k_chain_linked(vmx_link_job* jobs, int n_jobs, vmx_tables tab, const double* gapcost_list, double skipcost, int maxdiff, int maxgap, int lc, double margin_base);
__global__ void k_link_carry(vmx_link_job* jobs, int n_jobs, double skipcost);
__global__ void k_link_place(vmx_link_job* jobs, int n_jobs);

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
    if (!out || n <= 0 || n_pre < 0 || n_pre >= n || (which != 0 && which != 2) || maxdiff > 62) { set_error("vm_chain_linked: bad arguments"); return VM_ERR_ARG; }
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
    VMX_TRY(upload(B.job, &hj, 1, c->stream));
    vmx_link_job* d_job = B.job.as<vmx_link_job>(); const double* d_gap = B.gap.as<double>(); const vmx_tables tabs = c->tables; hipStream_t stq = c->stream;
    hipLaunchKernelGGL(k_link_place, dim3(1), dim3(256), 0, stq, d_job, 1);
    hipLaunchKernelGGL(k_chain_linked, dim3(1), dim3(64), 0, stq, d_job, 1, tabs, d_gap, skipcost, maxdiff, maxgap, lc, margin_base);
    hipLaunchKernelGGL(k_link_carry, dim3(1), dim3(64), 0, stq, d_job, 1, skipcost);
    VMX_TRY(download(&hj, B.job.p, 1, c->stream)); VMX_TRY(download(&hs, B.st.p, 1, c->stream));
    VMX_HIP(vmx_stream_sync(c));
    VMX_HIP(hipGetLastError());
    if (!hj.ran) { set_error("vm_chain_linked: the batch did not run"); return VM_ERR_HIP; }
    out->gmax = hj.gmax; out->n_hot = hj.hot; out->n_cold = hj.n_cold; out->cold_max = hj.cold_max; out->opcount = hj.opcount;
    const int base = cap_pre - (int)n_pre;
    out->S = (double*)malloc(8 * (size_t)n); out->P = (int64_t*)malloc(8 * (size_t)n); out->S_arg_hot = (int64_t*)malloc(8 * (size_t)std::max<int64_t>(1, hj.hot));
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
    return VM_OK;
}
