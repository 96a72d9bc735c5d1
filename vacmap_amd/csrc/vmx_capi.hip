// vmx_capi.hip — C-ABI entry points of libvacmapx.so (include/vacmapx.h): context, tables, DP and chain stage entries.
// Everything that computes runs the HIP kernels of this directory on the context's stream; there is no CPU path.
#include "vmx_host.h"
#include <time.h>
#include "vmx_select.h"
#include "vmx_stage.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace vmx {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " at " + what + " (" + file + ":" + std::to_string(line) + ")";
    return e == hipErrorOutOfMemory ? VM_ERR_OOM : VM_ERR_HIP;
}
}  // namespace vmx
using namespace vmx;

void vmx_ctx_free_local_bufs(vm_ctx* c);    // vmx_stage_local.hip
void vmx_ctx_free_batch_bufs(vm_ctx* c);    // vmx_align.hip

static inline int grid_for(const vm_ctx* c, int64_t n_items, int per_cu = 8) {
    int64_t g = std::min<int64_t>(n_items, (int64_t)c->num_cu * per_cu);
    return (int)std::max<int64_t>(g, 1);
}
template <class T> static T* host_alloc(size_t n) { return (T*)malloc(sizeof(T) * (n ? n : 1)); }

extern "C" {

const char* vm_last_error(void) { return g_err.c_str(); }
const char* vm_version(void) { return "vacmapx 0.1 (gfx950)"; }
void vm_free(void* p) { free(p); }

void vm_params_default(vm_params* p, int mode) {
    memset(p, 0, sizeof(*p));
    p->mode = mode; p->check_num = 100; p->mid_occ = -1;
    p->global_maxdiff = 50; p->local_maxdiff = 30; p->local_kmersize = 9;
    if (mode == VM_MODE_L) { p->local_skipcost = 59.; p->global_skipcost = 40.; p->maxdivergence = 0.1; }
    else if (mode == VM_MODE_H) { p->local_skipcost = 40.; p->global_skipcost = 40.; p->maxdivergence = 0.2; }
    else { p->local_skipcost = 30.; p->global_skipcost = 30.; p->maxdivergence = 0.5; }
    p->nodiscard = !(mode == VM_MODE_L || mode == VM_MODE_H);
    // -mode asm: --eqx forced (src/vacmap/vacmap:246), maxdivergence forced to 1 by the worker (mammap_asm.py:23483), check_num = -1 (:23206)
    if (mode == VM_MODE_ASM) { p->eqx = 1; p->maxdivergence = 1.0; p->check_num = -1; }
}

}  // extern "C"
// ---- the mailbox (vmx_host.h)
__global__ void k_signal(unsigned long long* word, unsigned long long v) {
#ifdef VMX_EMU
    *word = v;
#else
    __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
// every piece is cut into 16 KB slices dealt round-robin to the workgroups; the last workgroup to finish publishes the sequence word (system scope: the host polls it)
__global__ void __launch_bounds__(256) k_export(vmx_export_args A) {
    __shared__ int s_last;
    const unsigned G = gridDim.x, b = blockIdx.x;
    unsigned long long slice0 = 0;
    for (int p = 0; p < A.n; ++p) {
        const unsigned long long nb = A.bytes[p], ns = (nb + 16383ULL) >> 14;
        for (unsigned long long sl = (b + G - (unsigned)(slice0 % G)) % G; sl < ns; sl += G) {
            const unsigned long long lo = sl << 14, hi = lo + 16384ULL < nb ? lo + 16384ULL : nb;
            const char* s = A.src[p] + lo; char* d = A.dst[p] + lo;
            const unsigned long long n = hi - lo;
            if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
                const unsigned long long nv = n >> 4;
                struct b16 { unsigned long long x, y; };
                for (unsigned long long i = threadIdx.x; i < nv; i += blockDim.x) ((b16*)d)[i] = ((const b16*)s)[i];
                for (unsigned long long i = (nv << 4) + threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
            } else for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
        }
        slice0 += ns;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#ifdef VMX_EMU
        const unsigned long long old = __atomic_fetch_add(A.done, 1ULL, __ATOMIC_ACQ_REL);
#else
        __threadfence_system();
        const unsigned long long old = __hip_atomic_fetch_add(A.done, 1ULL, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
        s_last = old + 1ULL == (unsigned long long)G;
        if (s_last) {
#ifdef VMX_EMU
            *A.done = 0ULL; *A.word = A.seq;
#else
            __hip_atomic_store(A.done, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(A.word, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        }
    }
}
// the pieces of an upload group out of the ring (device-mapped host memory) into their places, 16 KB slices dealt round-robin to the workgroups
__global__ void __launch_bounds__(256) k_import(vmx_import_args A) {
    const unsigned G = gridDim.x, b = blockIdx.x;
    unsigned long long slice0 = 0;
    for (int p = 0; p < A.n; ++p) {
        const unsigned long long nb = A.bytes[p], ns = (nb + 16383ULL) >> 14;
        for (unsigned long long sl = (b + G - (unsigned)(slice0 % G)) % G; sl < ns; sl += G) {
            const unsigned long long lo = sl << 14, hi = lo + 16384ULL < nb ? lo + 16384ULL : nb, n = hi - lo;
            const char* s = A.src[p] + lo; char* d = A.dst[p] + lo;
            if ((((uintptr_t)s | (uintptr_t)d) & 7) == 0) {
                const unsigned long long nv = n >> 3;
                for (unsigned long long i = threadIdx.x; i < nv; i += blockDim.x) ((unsigned long long*)d)[i] = ((const unsigned long long*)s)[i];
                for (unsigned long long i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
            } else for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
        }
        slice0 += ns;
    }
}
static int mailbox_create(vm_ctx* c) {
    vmx_mailbox& m = c->mb;
    const char* mode = getenv("VMX_WAIT_MODE");
    if (mode && !strcmp(mode, "spin")) { m.on = false; return 0; }
    if (const char* e = getenv("VMX_POLL_US")) { const long long v = atoll(e); if (v >= 1 && v <= 100000) m.poll_ns = v * 1000; }
    const size_t up = (size_t)8 << 20, dn = (size_t)8 << 20, tot = 4096 + up + dn;
    void* hp = nullptr;
    if (hipHostMalloc(&hp, tot, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); m.on = false; return 0; }       // no page-locked memory: the legacy waits
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(hp); m.on = false; return 0; }
    memset(hp, 0, 4096);
    m.h = (char*)hp; m.d = (char*)dp; m.up_off = 4096; m.up_cap = up; m.dn_off = 4096 + up; m.dn_cap = dn; m.on = true;
    void* dc = nullptr;
    if (hipMalloc(&dc, 64) == hipSuccess) { (void)hipMemset(dc, 0, 64); m.done_ctr = (unsigned long long*)dc; } else (void)hipGetLastError();      // (without it: copies + k_signal)
    return 0;
}
static void mailbox_destroy(vm_ctx* c) {
    vmx_mailbox& m = c->mb;
    if (m.h) (void)hipHostFree(m.h);
    if (m.big) (void)hipHostFree(m.big);
    if (m.done_ctr) (void)hipFree(m.done_ctr);
    m = vmx_mailbox();
}
int vmx_mailbox_wait(vm_ctx* c) {
    vmx_mailbox& m = c->mb;
    const unsigned long long want = ++m.seq;
    {   // the noted downloads: one kernel per VMX_EXPORT_MAX pieces, the last of them carries the sequence word (no piece: the bare signal)
        vmx_export_args E; memset(&E, 0, sizeof E);
        unsigned long long tot = 0; bool signalled = false;
        auto flush = [&](bool last) {
            if (!E.n && !last) return;
            if (!E.n) { hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, c->stream, (unsigned long long*)m.d, want); signalled = true; return; }
            // the word of a kernel that is not the last one of this wait is a scratch word of the mailbox (offset 64): only the last launch moves the real one
            E.word = (unsigned long long*)(m.d + (last ? 0 : 64)); E.seq = want; E.done = m.done_ctr;
            const unsigned G = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>((tot + 16383ULL) >> 14, 64ULL));
            hipLaunchKernelGGL(k_export, dim3(G), dim3(256), 0, c->stream, E);
            if (last) signalled = true;
            memset(&E, 0, sizeof E); tot = 0;
        };
        size_t left = 0; for (const vmx_mailbox::Pending& p : m.pend) left += p.dev ? 1 : 0;
        for (const vmx_mailbox::Pending& p : m.pend) {
            if (!p.dev) continue;
            const char* land_d = (p.src >= m.h && p.src < m.h + m.dn_off + m.dn_cap) ? m.d + (p.src - m.h) : m.big_d + (p.src - m.big);
            E.src[E.n] = (const char*)p.dev; E.dst[E.n] = (char*)land_d; E.bytes[E.n] = p.bytes; ++E.n; tot += p.bytes; --left;
            if (E.n == VMX_EXPORT_MAX) flush(left == 0);
        }
        if (!signalled) flush(true);
    }
    volatile unsigned long long* w = (volatile unsigned long long*)m.h;
    struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)m.poll_ns;
    long long slept = 0, next_query = 500000000LL;
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) < want) {
        nanosleep(&ts, nullptr);
        slept += m.poll_ns;
        if (slept >= next_query) {                                    // a stream that died (a faulting kernel) never delivers the word
            next_query += 500000000LL;
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipSuccess && q != hipErrorNotReady) { m.pend.clear(); m.dn_used = 0; m.big_used = 0; m.up_used = 0; return vmx::hip_fail(q, "hipStreamQuery (mailbox wait)", __FILE__, __LINE__); }
        }
    }
    for (const vmx_mailbox::Pending& p : m.pend) memcpy(p.dst, p.src, p.bytes);
    m.pend.clear(); m.dn_used = 0; m.big_used = 0; m.up_used = 0;
    return 0;
}
int vmx_fetch_bytes(vm_ctx* c, void* host, const void* dev, size_t bytes) {
    vmx_mailbox& m = c->mb;
    char* land = nullptr;
    const size_t at = (m.dn_used + 63) & ~(size_t)63;
    if (at + bytes <= m.dn_cap) { land = m.h + m.dn_off + at; m.dn_used = at + bytes; }
    else {
        const size_t bat = (m.big_used + 63) & ~(size_t)63;
        if (bat + bytes > m.big_cap && m.big_used == 0) {             // grow the landing block while nothing is on its way into it
            if (m.big) { (void)hipHostFree(m.big); m.big = nullptr; m.big_cap = 0; m.big_d = nullptr; }
            const size_t want = bytes + bytes / 2 + ((size_t)4 << 20);
            void* hp = nullptr; void* dp = nullptr;
            if (hipHostMalloc(&hp, want, hipHostMallocMapped) == hipSuccess) {
                if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) { m.big = (char*)hp; m.big_d = (char*)dp; m.big_cap = want; }
                else { (void)hipGetLastError(); (void)hipHostFree(hp); }
            } else (void)hipGetLastError();
        }
        if (bat + bytes <= m.big_cap) { land = m.big + bat; m.big_used = bat + bytes; }
    }
    if (!land) {                                                      // no page-locked room: the plain copy (waits inside the call)
        VMX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        return 0;
    }
    static const bool by_copy = getenv("VMX_FETCH_COPIES") != nullptr;      // A/B knob: a runtime copy per piece + k_signal (the first form of the mailbox)
    if (by_copy || !m.done_ctr) { VMX_HIP(hipMemcpyAsync(land, dev, bytes, hipMemcpyDeviceToHost, c->stream)); m.pend.push_back(vmx_mailbox::Pending{host, land, bytes, nullptr}); }
    else m.pend.push_back(vmx_mailbox::Pending{host, land, bytes, dev});
    return 0;
}
// ---- counters of the row chain kernels (vmx_host.h)
static unsigned long long* g_chain_dbg = nullptr;
static std::mutex g_chain_dbg_m;
static int chain_dbg_enable() {
    std::lock_guard<std::mutex> g(g_chain_dbg_m);
    if (g_chain_dbg) return 0;
    unsigned long long* q = nullptr;
    if (hipMalloc((void**)&q, 64) != hipSuccess) { (void)hipGetLastError(); return -1; }
    (void)hipMemset(q, 0, 64);
    g_chain_dbg = q;
    return 0;
}
unsigned long long* vmx_chain_dbg() {
    static const bool env = [] { if (getenv("VMX_DBG_CHAIN")) (void)chain_dbg_enable(); return true; }(); (void)env;
    return g_chain_dbg;
}
void vmx_chain_dbg_report(hipStream_t st) {
    static const bool print = getenv("VMX_DBG_CHAIN") != nullptr;
    unsigned long long* p = vmx_chain_dbg(); if (!p || !print) return;
    unsigned long long h[8]; (void)hipStreamSynchronize(st); (void)hipMemcpy(h, p, 64, hipMemcpyDeviceToHost);
    fprintf(stderr, "[chain rows] global: anchors %llu scans past the window %llu insertions through HBM %llu opcount %llu | local: %llu %llu %llu %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}
extern "C" {
int vm_debug_chain_counters(int enable, unsigned long long* out8) {
    if (enable > 0 && chain_dbg_enable() < 0) { vmx::set_error("vm_debug_chain_counters: no device memory"); return VM_ERR_OOM; }
    unsigned long long* p = g_chain_dbg;
    if (out8) { memset(out8, 0, 64); if (p) { VMX_HIP(hipDeviceSynchronize()); VMX_HIP(hipMemcpy(out8, p, 64, hipMemcpyDeviceToHost)); } }
    if (enable < 0 && p) { VMX_HIP(hipDeviceSynchronize()); VMX_HIP(hipMemset(p, 0, 64)); }           // -1: read and reset
    return VM_OK;
}

int vm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vm_ctx_create(int device_id, vm_ctx** out) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device: libvacmapx has no CPU fallback"); return VM_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= n) { set_error("bad device id"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(device_id));
    vm_ctx* c = new vm_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
    int prio_least = 0, prio_greatest = 0;
    const bool prio = vmx_stream_prio_on() && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
    if ((prio ? hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_greatest) : hipStreamCreate(&c->stream)) != hipSuccess) { delete c; set_error("hipStreamCreate failed"); return VM_ERR_HIP; }
    if (prio && hipStreamCreateWithPriority(&c->low, hipStreamDefault, prio_least) == hipSuccess) { (void)hipEventCreateWithFlags(&c->low_ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&c->low_ev[1], hipEventDisableTiming); }
    else c->low = nullptr;
    for (int i = 0; i < 24; ++i) (void)hipEventCreate(&c->ev[i]);
    for (int i = 0; i < 48; ++i) (void)hipEventCreate(&c->gev[i]);
    for (int i = 0; i < 4; ++i) (void)hipEventCreate(&c->kev[i]);
    for (int i = 0; i < 4; ++i) { if (prio) (void)hipStreamCreateWithPriority(&c->aux[i], hipStreamDefault, prio_greatest); else (void)hipStreamCreate(&c->aux[i]); (void)hipEventCreate(&c->join_ev[i]); }
    (void)hipEventCreate(&c->fork_ev);
    (void)mailbox_create(c);
    // cost tables -> one device blob
    const HostTables& T = host_tables();
    size_t o_extra = 0, o_rh = o_extra + T.extra.size() * 4, o_rr = o_rh + 400, o_lr = o_rr + 400;
    size_t o_l2c = (o_lr + 400 + 15) & ~(size_t)15, o_l2i = o_l2c + T.log2cache.size() * 8, tot = o_l2i + T.log2int.size() * 8;
    std::vector<char> blob(tot);
    memcpy(&blob[o_extra], T.extra.data(), T.extra.size() * 4);
    memcpy(&blob[o_rh], T.readgap_h.data(), 400); memcpy(&blob[o_rr], T.readgap_r.data(), 400); memcpy(&blob[o_lr], T.large_readgap.data(), 400);
    memcpy(&blob[o_l2c], T.log2cache.data(), T.log2cache.size() * 8); memcpy(&blob[o_l2i], T.log2int.data(), T.log2int.size() * 8);
    if (c->tab_buf.reserve(tot) < 0) { delete c; return VM_ERR_OOM; }
    if (hipMemcpy(c->tab_buf.p, blob.data(), tot, hipMemcpyHostToDevice) != hipSuccess) { delete c; set_error("table upload failed"); return VM_ERR_HIP; }
    char* base = (char*)c->tab_buf.p;
    c->tables.extra = (const float*)(base + o_extra); c->tables.extra_n = (int)T.extra.size();
    c->tables.readgap_h = (const float*)(base + o_rh); c->tables.readgap_r = (const float*)(base + o_rr);
    c->tables.large_readgap = (const float*)(base + o_lr);
    c->tables.log2cache = (const double*)(base + o_l2c); c->tables.log2cache_n = (int)T.log2cache.size();
    c->tables.log2int = (const double*)(base + o_l2i);
    // closed-form prefix of `extra` (vmx_extra_cost, vmx_kernels.h): the same two multiplies the kernels do, compared entry by entry
    {
        int an = 0; const int lim = std::min<int>(20000, (int)T.extra.size() - 1);
        for (; an < lim; ++an) {
            const double dg = (double)an; const double a = dg * 0.01;
            const float v = (float)((a < 10.0 ? a : 10.0) + dg * 0.001);
            if (memcmp(&v, &T.extra[an], 4) != 0) break;
        }
        c->tables.extra_arith_n = an;
        if (T.extra.back() != 36.0f) { delete c; set_error("cost table `extra` does not end at 36.0"); return VM_ERR_UNSUPPORTED; }
    }
    *out = c;
    return VM_OK;
}

#ifdef VMX_CF_TICKS
extern "C" void vmx_cf_ticks_dump();
#endif
void vm_ctx_destroy(vm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
#ifdef VMX_CF_TICKS
    vmx_cf_ticks_dump();
#endif
    for (int i = 0; i < VMX_NBUF; ++i) c->b[i].release();
    vmx_ctx_free_local_bufs(c);
    vmx_ctx_free_batch_bufs(c);
    c->tab_buf.release();
    vmx::devbuf_retired().flush();
    for (int i = 0; i < 24; ++i) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 48; ++i) (void)hipEventDestroy(c->gev[i]);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(c->kev[i]);
    for (int i = 0; i < 4; ++i) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); (void)hipEventDestroy(c->join_ev[i]); }
    (void)hipEventDestroy(c->fork_ev);
    if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
    if (c->low) { (void)hipStreamSynchronize(c->low); (void)hipStreamDestroy(c->low); (void)hipEventDestroy(c->low_ev[0]); (void)hipEventDestroy(c->low_ev[1]); }
    (void)hipStreamDestroy(c->stream);
    mailbox_destroy(c);
    delete c;
}

int vm_ctx_set_inflight(vm_ctx* c, int n_contexts) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (n_contexts < 1) { set_error("n_contexts must be >= 1"); return VM_ERR_ARG; }
    c->inflight = n_contexts;
    return VM_OK;
}

int vm_ctx_mem_info(vm_ctx* c, int64_t* free_bytes, int64_t* total_bytes) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    size_t f = 0, t = 0;
#ifndef VMX_EMU
    VMX_HIP(hipSetDevice(c->device));
    VMX_HIP(hipMemGetInfo(&f, &t));
#else
    f = (size_t)1 << 34; t = (size_t)1 << 34;
#endif
    *free_bytes = (int64_t)f; *total_bytes = (int64_t)t;
    return VM_OK;
}

int vm_ctx_set_blocking_sync(vm_ctx* c, int on) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    // round 6: sleeping waits are the default (the mailbox of vmx_host.h); on == 0 switches this context to the legacy spinning waits (A/B runs)
    if (on && !c->mb.h) { VMX_HIP(hipSetDevice(c->device)); (void)mailbox_create(c); }
    c->mb.on = on != 0 && c->mb.h != nullptr;
    return VM_OK;
}

int64_t vm_table(vm_ctx* c, int which, void** data) {
    if (!c) return VM_ERR_NO_CTX;
    const vmx_tables& t = c->tables;
    const void* src; size_t n, es;
    switch (which) {
        case 0: src = t.extra; n = t.extra_n; es = 4; break;
        case 1: src = t.readgap_h; n = 100; es = 4; break;
        case 2: src = t.readgap_r; n = 100; es = 4; break;
        case 3: src = t.large_readgap; n = 100; es = 4; break;
        case 4: src = t.log2cache; n = t.log2cache_n; es = 8; break;
        case 5: src = t.log2int; n = 1025; es = 8; break;
        default: return VM_ERR_ARG;
    }
    *data = malloc(n * es);
    VMX_HIP(hipMemcpy(*data, src, n * es, hipMemcpyDeviceToHost));
    return (int64_t)n;
}

}  // extern "C"

// upload concatenated ASCII + encode to codes in buf_codes; offsets uploaded to buf_off
static int upload_encode(vm_ctx* c, const char* s, const int64_t* off, int64_t n, DevBuf& raw, DevBuf& codes, DevBuf& doff) {
    const int64_t tot = off[n];
    VMX_TRY(upload(raw, s, (size_t)tot, c->stream));
    VMX_TRY(codes.reserve((size_t)tot + 64));
    VMX_TRY(upload(doff, off, (size_t)n + 1, c->stream));
    if (tot) hipLaunchKernelGGL(k_encode, dim3(grid_for(c, (tot + 255) / 256)), dim3(256), 0, c->stream, raw.as<char>(), codes.as<uint8_t>(), tot);
    return 0;
}

extern "C" {

int vm_edit_distance_batch(vm_ctx* c, int64_t n, const char* q, const int64_t* q_off, const char* t, const int64_t* t_off, int64_t** dist) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    *dist = nullptr;
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_encode(c, q, q_off, n, c->b[0], c->b[1], c->b[2]));
    VMX_TRY(upload_encode(c, t, t_off, n, c->b[3], c->b[4], c->b[5]));
    VMX_TRY(c->b[6].reserve((size_t)VMX_ED_WAVES * (size_t)t_off[n] + 64));     // carry ring: VMX_ED_WAVES x one int8 per text column
    VMX_TRY(c->b[7].reserve(sizeof(int64_t) * (size_t)(n + 1)));
    if (n) {
        // longest-first device work queue; long patterns (> 4 passes) on 16-wave workgroups, the rest on 4-wave workgroups
        std::vector<int64_t> sz((size_t)n); for (int64_t i = 0; i < n; ++i) sz[i] = q_off[i + 1] - q_off[i];
        int32_t nn = (int32_t)n;
        VMX_TRY(upload(c->b[8], sz.data(), (size_t)n, c->stream)); VMX_TRY(upload(c->b[9], &nn, 1, c->stream));
        VMX_TRY(c->b[10].reserve(4 * (size_t)(n + 1))); VMX_TRY(c->b[11].reserve(64));
        int32_t* d_range = c->b[11].as<int32_t>(); int32_t* d_cnt = d_range + 4;
        int64_t thresh = VMX_ED_LONG;
        if (const char* e = getenv("VMX_ED_LONG")) thresh = atoll(e);      // test knob: exercise the 16-wave launch with short patterns
        hipLaunchKernelGGL(k_size_order, dim3(1), dim3(1024), 0, c->stream, c->b[8].as<int64_t>(), c->b[9].as<int32_t>(), thresh, c->b[10].as<int32_t>(), d_range, d_cnt);
        for (int which = 0; which < 2; ++which)
            hipLaunchKernelGGL(k_edit_distance, dim3(grid_for(c, n, which == 0 ? 2 : 8)), dim3(which == 0 ? 64 * VMX_ED_WAVES : 256), 0, c->stream, c->b[1].as<uint8_t>(),
                               c->b[2].as<int64_t>(), c->b[4].as<uint8_t>(), c->b[5].as<int64_t>(), c->b[6].as<int8_t>(), c->b[10].as<int32_t>(), d_range, d_cnt, which,
                               c->b[7].as<int64_t>(), (int64_t)0, (int32_t*)nullptr);
    }
    *dist = host_alloc<int64_t>((size_t)n);
    VMX_TRY(download(*dist, c->b[7].p, (size_t)n, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    VMX_HIP(hipGetLastError());
    return VM_OK;
}

int vm_edit_distance_bound_batch(vm_ctx* c, int tier, int64_t n, const char* q, const int64_t* q_off, const char* t, const int64_t* t_off, int64_t** bound) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    *bound = nullptr;
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_encode(c, q, q_off, n, c->b[0], c->b[1], c->b[2]));
    VMX_TRY(upload_encode(c, t, t_off, n, c->b[3], c->b[4], c->b[5]));
    VMX_TRY(c->b[7].reserve(sizeof(int64_t) * (size_t)(n + 1)));
    if (n) {
        std::vector<int64_t> sz((size_t)n); for (int64_t i = 0; i < n; ++i) sz[i] = q_off[i + 1] - q_off[i];
        int32_t nn = (int32_t)n;
        VMX_TRY(upload(c->b[8], sz.data(), (size_t)n, c->stream)); VMX_TRY(upload(c->b[9], &nn, 1, c->stream));
        VMX_TRY(c->b[10].reserve(4 * (size_t)(n + 1))); VMX_TRY(c->b[11].reserve(64));
        int32_t* d_range = c->b[11].as<int32_t>(); int32_t* d_cnt = d_range + 4;
        hipLaunchKernelGGL(k_size_order, dim3(1), dim3(1024), 0, c->stream, c->b[8].as<int64_t>(), c->b[9].as<int32_t>(), (int64_t)0, c->b[10].as<int32_t>(), d_range, d_cnt);
        if (tier == 1)
            hipLaunchKernelGGL(k_ed_banded4, dim3(grid_for(c, (n + 3) / 4, 16)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(), c->b[4].as<uint8_t>(),
                               c->b[5].as<int64_t>(), c->b[10].as<int32_t>(), d_range, d_cnt, c->b[7].as<int64_t>());
        else
            hipLaunchKernelGGL(k_ed_banded, dim3(grid_for(c, n, 16)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(), c->b[4].as<uint8_t>(), c->b[5].as<int64_t>(),
                               c->b[10].as<int32_t>(), d_range, d_cnt, c->b[7].as<int64_t>());
    }
    *bound = host_alloc<int64_t>((size_t)n);
    VMX_TRY(download(*bound, c->b[7].p, (size_t)n, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    VMX_HIP(hipGetLastError());
    return VM_OK;
}

int64_t vm_edit_distance(vm_ctx* c, const char* q, int64_t ql, const char* t, int64_t tl) {
    int64_t qo[2] = {0, ql}, to[2] = {0, tl};
    int64_t* d = nullptr;
    int rc = vm_edit_distance_batch(c, 1, q, qo, t, to, &d);
    if (rc < 0) return rc;
    int64_t v = d[0]; free(d);
    return v;
}

int vm_k_extend_batch(vm_ctx* c, int match, int mismatch, int o, int e, int bw, int zdrop, int64_t n, const char* t, const int64_t* t_off,
                      const char* q, const int64_t* q_off, int32_t** t_e, int32_t** q_e, int32_t** score) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (bw > 496) { set_error("vm_k_extend_batch: band wider than 496 unsupported"); return VM_ERR_UNSUPPORTED; }
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_encode(c, t, t_off, n, c->b[0], c->b[1], c->b[2]));
    VMX_TRY(upload_encode(c, q, q_off, n, c->b[3], c->b[4], c->b[5]));
    VMX_TRY(c->b[6].reserve(sizeof(int32_t) * 3 * (size_t)(n + 1)));
    int32_t* d_te = c->b[6].as<int32_t>(); int32_t* d_qe = d_te + n; int32_t* d_sc = d_qe + n;
    if (n) hipLaunchKernelGGL(k_extend, dim3(grid_for(c, n, 16)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[2].as<int64_t>(),
                              c->b[4].as<uint8_t>(), c->b[5].as<int64_t>(), (int)n, match, mismatch, o, e, bw, zdrop, d_te, d_qe, d_sc, (const int32_t*)nullptr);
    *t_e = host_alloc<int32_t>((size_t)n); *q_e = host_alloc<int32_t>((size_t)n); *score = host_alloc<int32_t>((size_t)n);
    VMX_TRY(download(*t_e, d_te, (size_t)n, c->stream)); VMX_TRY(download(*q_e, d_qe, (size_t)n, c->stream));
    VMX_TRY(download(*score, d_sc, (size_t)n, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    VMX_HIP(hipGetLastError());
    return VM_OK;
}

// schedule = 0: k_gapfill_fill (full matrix, scores captured). schedule = 1: exactly what vm_align_batch launches for its gap-fill
// problems (vmx_align.hip): longest-first device queue, k_gapfill_fill_ns with the anti-diagonal BAND form first (eight small problems per
// wave), the problems whose band is not proven (or that are not worth a band) queued and filled in full by the second launch, the
// per-problem layout flag handed to k_gapfill_trace. stats: small problems tried in a band, kept (proven), queued for the second launch
// (incl. the small ones never tried), problems outside the small class.
static int k_cigar_batch_impl(vm_ctx* c, const vm_score* sc, int eqx, int schedule, int64_t n, const char* t, const int64_t* t_off, const char* q,
                              const int64_t* q_off, char** cigars, int64_t** cigar_off, int32_t** scores, int32_t** band_flag, int64_t* stats) {
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (n > 0x7fffffff) { set_error("too many problems"); return VM_ERR_ARG; }
    VMX_HIP(hipSetDevice(c->device));
    VMX_TRY(upload_encode(c, t, t_off, n, c->b[0], c->b[1], c->b[2]));
    VMX_TRY(upload_encode(c, q, q_off, n, c->b[3], c->b[4], c->b[5]));
    std::vector<vmx_dp_prob> probs((size_t)n);
    std::vector<int64_t> tbsz((size_t)n);
    int64_t tb = 0, bnd = 0, run = 0, cig = 0;
    for (int64_t i = 0; i < n; ++i) {
        vmx_dp_prob& p = probs[i];
        p.t_off = t_off[i]; p.q_off = q_off[i]; p.tl = (int32_t)(t_off[i + 1] - t_off[i]); p.ql = (int32_t)(q_off[i + 1] - q_off[i]);
        p.tb_off = tb; p.bnd_off = bnd; p.run_off = run; p.cig_off = cig;
        tbsz[i] = schedule == 0 ? VMX_TB_BYTES((int64_t)p.tl, (int64_t)p.ql) : VMX_TB_BYTES_NS((int64_t)p.tl, (int64_t)p.ql);
        tb += tbsz[i];
        bnd += 3 * (int64_t)(p.ql + 1); run += (int64_t)p.tl + p.ql + 2; cig += 2 * ((int64_t)p.tl + p.ql) + 16;
    }
    VMX_TRY(upload(c->b[6], probs.data(), (size_t)n, c->stream));
    VMX_TRY(c->b[7].reserve((size_t)tb + 64)); VMX_TRY(c->b[8].reserve(sizeof(int32_t) * (size_t)(bnd + 4)));
    VMX_TRY(c->b[9].reserve(sizeof(uint32_t) * (size_t)(run + 4))); VMX_TRY(c->b[10].reserve((size_t)cig + 16));
    VMX_TRY(c->b[11].reserve(sizeof(int32_t) * 2 * (size_t)(n + 1)));
    int32_t* d_score = c->b[11].as<int32_t>(); int32_t* d_len = d_score + n;
    int32_t redo[2] = {0, 0};
    const int ad_pct = vmx_ad_pct_env();
    if (n && schedule == 0) {
        hipLaunchKernelGGL(k_gapfill_fill, dim3(grid_for(c, n, 16)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[4].as<uint8_t>(),
                           c->b[6].as<vmx_dp_prob>(), (int)n, sc->match, sc->mismatch, sc->o1, sc->e1, sc->o2, sc->e2, c->b[7].as<uint8_t>(),
                           c->b[8].as<int32_t>(), d_score, (const int32_t*)nullptr, (int32_t*)nullptr);
        hipLaunchKernelGGL(k_gapfill_trace, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[4].as<uint8_t>(),
                           c->b[6].as<vmx_dp_prob>(), (int)n, eqx, c->b[7].as<uint8_t>(), c->b[9].as<uint32_t>(), c->b[10].as<char>(), d_len, (const int32_t*)nullptr, (const uint8_t*)nullptr, 1, (int32_t*)nullptr);
    } else if (n) {
        const int32_t nn = (int32_t)n;
        VMX_TRY(upload(c->b[12], tbsz.data(), (size_t)n, c->stream)); VMX_TRY(upload(c->b[13], &nn, 1, c->stream));
        VMX_TRY(c->b[14].reserve(4 * (size_t)(2 * n + 64))); VMX_TRY(c->b[15].reserve(128));
        int32_t* d_range = c->b[15].as<int32_t>(); int32_t* d_cnt = d_range + 4; int32_t* d_redo_cnt = d_range + 12;
        unsigned long long* d_redo_bytes = (unsigned long long*)(d_range + 16);
        int32_t* d_order = c->b[14].as<int32_t>(); int32_t* d_redo_list = d_order + n + 32;
        hipLaunchKernelGGL(k_size_order, dim3(1), dim3(1024), 0, c->stream, c->b[12].as<int64_t>(), c->b[13].as<int32_t>(), (int64_t)VMX_HEAD_THRESH, d_order, d_range, d_cnt);
        VMX_HIP(hipMemsetAsync(d_redo_cnt, 0, 16, c->stream)); VMX_HIP(hipMemsetAsync(d_redo_bytes, 0, 8, c->stream));
        hipLaunchKernelGGL(k_gapfill_fill_ns, dim3(grid_for(c, n, 16)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[4].as<uint8_t>(), c->b[6].as<vmx_dp_prob>(), (int)n,
                           sc->match, sc->mismatch, sc->o1, sc->e1, sc->o2, sc->e2, c->b[7].as<uint8_t>(), c->b[8].as<int32_t>(), d_score, d_order, d_range, d_cnt, d_redo_list, d_redo_cnt, 0, ad_pct,
                           (uint8_t*)nullptr, d_redo_bytes);
        unsigned long long redo_bytes = 0;
        VMX_TRY(download(redo, d_redo_cnt, 1, c->stream));          // entries queued by the first launch and the full-matrix traceback space they need
        VMX_TRY(download(&redo_bytes, d_redo_bytes, 1, c->stream));
        VMX_HIP(hipStreamSynchronize(c->stream));
        VMX_TRY(c->b[16].reserve((size_t)redo_bytes + 64));
        hipLaunchKernelGGL(k_gapfill_fill_ns, dim3(grid_for(c, (n + 3) / 4, 4)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[4].as<uint8_t>(), c->b[6].as<vmx_dp_prob>(), (int)n,
                           sc->match, sc->mismatch, sc->o1, sc->e1, sc->o2, sc->e2, c->b[7].as<uint8_t>(), c->b[8].as<int32_t>(), d_score, d_order, d_range, d_cnt, d_redo_list, d_redo_cnt, 1, ad_pct,
                           c->b[16].as<uint8_t>(), d_redo_bytes);
        hipLaunchKernelGGL(k_gapfill_trace, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->b[1].as<uint8_t>(), c->b[4].as<uint8_t>(),
                           c->b[6].as<vmx_dp_prob>(), (int)n, eqx, c->b[7].as<uint8_t>(), c->b[9].as<uint32_t>(), c->b[10].as<char>(), d_len, d_score, c->b[16].as<uint8_t>(), 1, (int32_t*)nullptr);
    }
    std::vector<char> hc((size_t)cig + 16); std::vector<int32_t> hl((size_t)n);
    *scores = host_alloc<int32_t>((size_t)n);
    VMX_TRY(download(hc.data(), c->b[10].p, (size_t)cig, c->stream));
    VMX_TRY(download(hl.data(), d_len, (size_t)n, c->stream));
    VMX_TRY(download(*scores, d_score, (size_t)n, c->stream));
    VMX_HIP(hipStreamSynchronize(c->stream));
    VMX_HIP(hipGetLastError());
    if (schedule != 0) {            // the _ns form leaves the layout flag where the score would be (it never captures scores)
        if (band_flag) { *band_flag = host_alloc<int32_t>((size_t)n); memcpy(*band_flag, *scores, sizeof(int32_t) * (size_t)n); }
        int64_t nb = 0, eligible = 0;
        for (int64_t i = 0; i < n; ++i) {
            const bool small = probs[i].tl > 0 && probs[i].ql > 0 && VMX_DP16X4_OK(probs[i].tl, probs[i].ql);
            if (!small && band_flag) (*band_flag)[i] = 0;         // (the larger forms leave their score there)
            nb += small && (*scores)[i] > VMX_AD_FLAG;
            eligible += probs[i].tl > 0 && probs[i].ql > 0 && VMX_DP16X4_OK(probs[i].tl, probs[i].ql) &&
                        vmx_ad_ns(probs[i].tl, probs[i].ql, sc->match, sc->o1, sc->e1, sc->o2, sc->e2, ad_pct & 0xffff, (ad_pct >> 16) & 0xffff) > 0;
            (*scores)[i] = 0;
        }
        if (stats) {
            int64_t small = 0;
            for (int64_t i = 0; i < n; ++i) small += probs[i].tl > 0 && probs[i].ql > 0 && VMX_DP16X4_OK(probs[i].tl, probs[i].ql);
            stats[0] = eligible; stats[1] = nb; stats[2] = redo[0]; stats[3] = n - small;
        }
    } else if (band_flag) *band_flag = nullptr;
    *cigar_off = host_alloc<int64_t>((size_t)n + 1);
    int64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) { (*cigar_off)[i] = tot; tot += hl[i] + 1; }
    (*cigar_off)[n] = tot;
    *cigars = host_alloc<char>((size_t)tot + 1);
    for (int64_t i = 0; i < n; ++i) { memcpy(*cigars + (*cigar_off)[i], hc.data() + probs[i].cig_off, (size_t)hl[i]); (*cigars)[(*cigar_off)[i] + hl[i]] = 0; }
    return VM_OK;
}

int vm_k_cigar_batch(vm_ctx* c, const vm_score* sc, int eqx, int64_t n, const char* t, const int64_t* t_off, const char* q,
                     const int64_t* q_off, char** cigars, int64_t** cigar_off, int32_t** scores) {
    return k_cigar_batch_impl(c, sc, eqx, 0, n, t, t_off, q, q_off, cigars, cigar_off, scores, nullptr, nullptr);
}

int vm_k_cigar_batch_banded(vm_ctx* c, const vm_score* sc, int eqx, int64_t n, const char* t, const int64_t* t_off, const char* q,
                            const int64_t* q_off, char** cigars, int64_t** cigar_off, int32_t** band_flag, int64_t* stats) {
    int32_t* dummy = nullptr;
    const int rc = k_cigar_batch_impl(c, sc, eqx, 1, n, t, t_off, q, q_off, cigars, cigar_off, &dummy, band_flag, stats);
    free(dummy);
    return rc;
}

int vm_k_cigar(vm_ctx* c, const char* t, int64_t tl, const char* q, int64_t ql, const vm_score* sc, int bw, int zdrop, int eqx, vm_cigar_out* out) {
    memset(out, 0, sizeof(*out));
    int64_t to[2] = {0, tl}, qo[2] = {0, ql};
    if (zdrop < 0) {   // global end-to-end (:21554)
        char* cg = nullptr; int64_t* co = nullptr; int32_t* s = nullptr;
        int rc = vm_k_cigar_batch(c, sc, eqx, 1, t, to, q, qo, &cg, &co, &s);
        if (rc < 0) return rc;
        out->cigar = cg; out->q_e = (int32_t)ql; out->t_e = (int32_t)tl; out->score = s[0];
        free(co); free(s);
        return VM_OK;
    }
    if (sc->o1 != sc->o2 || sc->e1 != sc->e2) { set_error("vm_k_cigar: z-drop extension needs identical gap pieces (the reference calls it with 4,4,4,4)"); return VM_ERR_UNSUPPORTED; }
    int32_t *te = nullptr, *qe = nullptr, *s = nullptr;
    int rc = vm_k_extend_batch(c, sc->match, sc->mismatch, sc->o1, sc->e1, bw, zdrop, 1, t, to, q, qo, &te, &qe, &s);
    if (rc < 0) return rc;
    out->cigar = (char*)calloc(1, 1); out->t_e = te[0]; out->q_e = qe[0]; out->score = s[0];
    free(te); free(qe); free(s);
    return VM_OK;
}

// ------------------------------------------------------------------------------------------------ global chain stage
void vm_chains_out_free(vm_chains_out* o) {
    free(o->need_reverse); free(o->mapq); free(o->score); free(o->fast_used); free(o->read_path_off); free(o->path_off);
    free(o->path_anchors); free(o->S); free(o->P); free(o->S_arg); free(o->gmax); free(o->opcount);
    memset(o, 0, sizeof(*o));
}

int vm_chain_global_batch(vm_ctx* c, const vm_params* prm, int kmersize, int64_t n, const int64_t* anchors, const int64_t* aoff,
                          const int64_t* readlens, int want_raw, vm_chains_out* out) {
    memset(out, 0, sizeof(*out));
    if (!c) { set_error("no context"); return VM_ERR_NO_CTX; }
    if (prm->global_maxdiff > 62) { set_error("global_maxdiff > 62 unsupported"); return VM_ERR_UNSUPPORTED; }
    VMX_HIP(hipSetDevice(c->device));
    const int64_t tot = aoff[n];
    DevBuf &d_rows = c->b[0], &d_aoff = c->b[1], &d_len = c->b[2], &d_keys = c->b[3], &d_koff = c->b[4], &d_sorted = c->b[5], &d_flip = c->b[6];
    DevBuf &d_S = c->b[7], &d_P = c->b[8], &d_SA = c->b[9], &d_cov = c->b[10], &d_gmax = c->b[11], &d_opc = c->b[12], &d_rl = c->b[13];
    DevBuf &d_gap = c->b[14], &d_scr = c->b[15], &d_soff = c->b[16], &d_res = c->b[17], &d_plen = c->b[18], &d_prow = c->b[19];
    VMX_TRY(upload(d_rows, anchors, (size_t)tot * 4, c->stream));
    VMX_TRY(upload(d_aoff, aoff, (size_t)n + 1, c->stream));
    VMX_TRY(upload(d_len, readlens, (size_t)n, c->stream));
    std::vector<int64_t> koff((size_t)n + 1), soff((size_t)n + 1);
    int64_t kt = 0, st = 0;
    for (int64_t r = 0; r < n; ++r) {
        int64_t m = aoff[r + 1] - aoff[r]; int64_t N = 1; while (N < m) N <<= 1;
        koff[r] = kt; kt += N;
        soff[r] = st; st += (vmx_select_scratch_bytes(m) + 15) & ~(int64_t)15;
    }
    koff[n] = kt; soff[n] = st;
    VMX_TRY(d_keys.reserve(sizeof(uint64_t) * (size_t)(kt + 1)));
    VMX_TRY(upload(d_koff, koff.data(), (size_t)n + 1, c->stream));
    VMX_TRY(d_sorted.reserve(sizeof(vmx_anchor) * (size_t)(tot + 1)));
    VMX_TRY(d_flip.reserve(sizeof(int32_t) * (size_t)(n + 1)));
    if (n) hipLaunchKernelGGL(k_flip_sort, dim3(grid_for(c, n, 8)), dim3(256), 0, c->stream, d_rows.as<int64_t>(), d_aoff.as<int64_t>(),
                              d_len.as<int64_t>(), (int)n, d_keys.as<uint64_t>(), d_koff.as<int64_t>(), d_sorted.as<vmx_anchor>(), d_flip.as<int32_t>());
    VMX_TRY(d_S.reserve(sizeof(double) * (size_t)(tot + 1))); VMX_TRY(d_P.reserve(sizeof(int32_t) * (size_t)(tot + 1)));
    VMX_TRY(d_SA.reserve(sizeof(int32_t) * (size_t)(tot + 1))); VMX_TRY(d_cov.reserve((size_t)tot + 16));
    VMX_TRY(d_gmax.reserve(sizeof(int64_t) * (size_t)(n + 1))); VMX_TRY(d_opc.reserve(sizeof(int64_t) * (size_t)(n + 1)));
    // gapcost_list (:24843-24846): 0.01*k*g + 0.5*log2(g), evaluated in double exactly like the reference
    const HostTables& T = host_tables();
    std::vector<double> gap(64, 0.0);
    for (int g = 1; g <= prm->global_maxdiff; ++g) gap[g] = (0.01 * kmersize * g + 0.5 * T.log2int[g]);
    VMX_TRY(upload(d_gap, gap.data(), 64, c->stream));
    const int rmode = prm->mode == VM_MODE_R ? 1 : (prm->mode == VM_MODE_ASM ? 2 : 0);
    if (rmode == 1) { VMX_TRY(c->b[25].reserve(8 * (size_t)(tot + 1))); VMX_TRY(c->b[26].reserve(8 * (size_t)(tot + 1))); }   // mode R: fixed_penatly / pre_penatly
    // bucket the reads by anchor count so that each launch asks for no more LDS than it needs (160 KiB per CU on gfx950)
    const int caps[4] = {768, 1536, 3072, 13056};
    std::vector<int32_t> lists[5];
    std::vector<char> fastflag((size_t)n, 0);
    for (int64_t r = 0; r < n; ++r) {
        int64_t m = aoff[r + 1] - aoff[r];
        if ((double)m / (double)readlens[r] > 5.0) { fastflag[r] = 1; continue; }   // fast_enable (:23570)
        int bk = 4; for (int k = 0; k < 4; ++k) if (m <= caps[k]) { bk = k; break; }
        lists[bk].push_back((int32_t)r);
    }
    std::vector<int32_t> rl; std::vector<int64_t> rl_off(6, 0);
    for (int k = 0; k < 5; ++k) { rl_off[k] = (int64_t)rl.size(); rl.insert(rl.end(), lists[k].begin(), lists[k].end()); }
    rl_off[5] = (int64_t)rl.size();
    VMX_TRY(upload(d_rl, rl.data(), rl.size(), c->stream));
    VMX_HIP(hipMemsetAsync(d_gmax.p, 0xff, sizeof(int64_t) * (size_t)n, c->stream));   // -1 = needs GC-fast
    if (vmx_chain_rows_on() && rmode != 2 && !rl.empty()) {            // four reads per wavefront (k_chain_rows.hip), most anchors first
        std::vector<int32_t> all(rl);
        std::stable_sort(all.begin(), all.end(), [&](int32_t a, int32_t b) { return aoff[a + 1] - aoff[a] > aoff[b + 1] - aoff[b]; });
        VMX_TRY(upload(d_rl, all.data(), all.size(), c->stream));
        const int cnt = (int)all.size();
        hipLaunchKernelGGL((vmx_chain_rows_win3() ? k_chain_global_rows_w3 : k_chain_global_rows), dim3((unsigned)((cnt + 3) / 4)), dim3(64), 0, c->stream, d_sorted.as<vmx_anchor>(), d_aoff.as<int64_t>(),
                           d_rl.as<int32_t>(), cnt, c->tables, d_gap.as<double>(), prm->global_skipcost, prm->global_maxdiff,
                           1000, d_S.as<double>(), d_P.as<int32_t>(), d_SA.as<int32_t>(), d_cov.as<uint8_t>(), d_gmax.as<int64_t>(), d_opc.as<int64_t>(), rmode,
                           c->b[25].as<double>(), c->b[26].as<double>(), vmx_chain_dbg());
        for (auto& l : lists) l.clear();
    }
    for (int k = 0; k < 5; ++k) {
        int cnt = (int)lists[k].size();
        if (!cnt) continue;
        int cap = k < 4 ? caps[k] : 0;
        size_t shmem = (size_t)cap * VMX_GC_BYTES_PER_ANCHOR + 64;
#ifndef VMX_EMU
        if (shmem > 48 * 1024) VMX_HIP(hipFuncSetAttribute((const void*)k_chain_global, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
#endif
        hipLaunchKernelGGL(k_chain_global, dim3(grid_for(c, cnt, 8)), dim3(64), shmem, c->stream, d_sorted.as<vmx_anchor>(), d_aoff.as<int64_t>(),
                           d_rl.as<int32_t>() + rl_off[k], cnt, cap, c->tables, d_gap.as<double>(), prm->global_skipcost, prm->global_maxdiff,
                           1000, d_S.as<double>(), d_P.as<int32_t>(), d_SA.as<int32_t>(), d_cov.as<uint8_t>(), d_gmax.as<int64_t>(), d_opc.as<int64_t>(), rmode,
                           c->b[25].as<double>(), c->b[26].as<double>());
    }
    // G3: reads left at gmax = -1 (fast_enable or the opcount bail-out) go through GC-fast (k_chain_fast.hip)
    if (n) {
        DevBuf &d_si = c->b[20], &d_tg = c->b[21], &d_cnt = c->b[22], &d_roff = c->b[23], &d_ran = c->b[24];
        VMX_TRY(d_ran.reserve(4 * (size_t)(n + 1))); VMX_HIP(hipMemsetAsync(d_ran.p, 0, 4 * (size_t)n, c->stream));
        std::vector<int64_t> roff((size_t)n + 1, 0);
        for (int64_t r = 0; r < n; ++r) roff[(size_t)r + 1] = roff[(size_t)r] + readlens[r];
        VMX_TRY(upload(d_roff, roff.data(), (size_t)n + 1, c->stream));
        VMX_TRY(d_si.reserve(4 * (size_t)(tot + 1))); VMX_TRY(d_tg.reserve(8 * (size_t)(tot + 1))); VMX_TRY(d_cnt.reserve(4 * (size_t)(roff[(size_t)n] + 50 * n + 64)));
        hipLaunchKernelGGL(k_chain_global_fast, dim3((unsigned)n), dim3(64), 0, c->stream, d_sorted.as<vmx_anchor>(), d_aoff.as<int64_t>(), (int)n, d_roff.as<int64_t>(),
                           c->tables, d_gap.as<double>(), prm->global_skipcost, prm->global_maxdiff, 1000, d_S.as<double>(), d_P.as<int32_t>(), d_SA.as<int32_t>(),
                           d_cov.as<uint8_t>(), d_si.as<int32_t>(), d_tg.as<int64_t>(), d_cnt.as<int32_t>(), d_gmax.as<int64_t>(), d_ran.as<int32_t>(), rmode,
                           c->b[25].as<double>(), c->b[26].as<double>());
    }
    VMX_TRY(d_scr.reserve((size_t)st + 64));
    VMX_TRY(upload(d_soff, soff.data(), (size_t)n + 1, c->stream));
    VMX_TRY(d_res.reserve((sizeof(double) + 2 * sizeof(int32_t)) * (size_t)(n + 1) + 64));
    double* d_score = d_res.as<double>(); int32_t* d_mapq = (int32_t*)(d_score + n + 1); int32_t* d_np = d_mapq + n + 1;
    VMX_TRY(d_plen.reserve(sizeof(int32_t) * (size_t)(tot + 1))); VMX_TRY(d_prow.reserve(sizeof(vmx_anchor) * (size_t)(tot + 1)));
    std::vector<int64_t> h_ao(aoff, aoff + n + 1);
    VMX_TRY(vmx_launch_chain_select(c, n, h_ao.data(), c->b[27], d_sorted.as<vmx_anchor>(), d_aoff.as<int64_t>(), d_len.as<int64_t>(), d_S.as<double>(), d_P.as<int32_t>(), d_SA.as<int32_t>(),
                                    d_gmax.as<int64_t>(), d_flip.as<int32_t>(), prm->mode, d_scr.as<char>(), d_soff.as<int64_t>(), d_mapq, d_score, d_np, d_plen.as<int32_t>(), d_prow.as<vmx_anchor>()));
    // download
    std::vector<int32_t> h_np((size_t)n), h_plen((size_t)tot);
    std::vector<vmx_anchor> h_prow((size_t)tot);
    out->need_reverse = host_alloc<int32_t>((size_t)n); out->mapq = host_alloc<int32_t>((size_t)n); out->score = host_alloc<double>((size_t)n);
    out->fast_used = host_alloc<int32_t>((size_t)n); out->gmax = host_alloc<int64_t>((size_t)n); out->opcount = host_alloc<int64_t>((size_t)n);
    VMX_TRY(download(out->need_reverse, d_flip.p, (size_t)n, c->stream)); VMX_TRY(download(out->mapq, d_mapq, (size_t)n, c->stream));
    VMX_TRY(download(out->score, d_score, (size_t)n, c->stream)); VMX_TRY(download(h_np.data(), d_np, (size_t)n, c->stream));
    VMX_TRY(download(h_plen.data(), d_plen.p, (size_t)tot, c->stream)); VMX_TRY(download(h_prow.data(), d_prow.p, (size_t)tot, c->stream));
    VMX_TRY(download(out->gmax, d_gmax.p, (size_t)n, c->stream)); VMX_TRY(download(out->opcount, d_opc.p, (size_t)n, c->stream));
    std::vector<int32_t> hP, hSA;
    if (want_raw) {
        out->S = host_alloc<double>((size_t)tot); out->P = host_alloc<int64_t>((size_t)tot); out->S_arg = host_alloc<int64_t>((size_t)tot);
        hP.resize((size_t)tot); hSA.resize((size_t)tot);
        VMX_TRY(download(out->S, d_S.p, (size_t)tot, c->stream)); VMX_TRY(download(hP.data(), d_P.p, (size_t)tot, c->stream));
        VMX_TRY(download(hSA.data(), d_SA.p, (size_t)tot, c->stream));
    }
    VMX_HIP(hipStreamSynchronize(c->stream));
    VMX_HIP(hipGetLastError());
    if (want_raw) for (int64_t i = 0; i < tot; ++i) { out->P[i] = hP[i]; out->S_arg[i] = hSA[i]; }
    for (int64_t r = 0; r < n; ++r) if (h_np[r] < 0) h_np[r] = 0;      // -mode asm: a contig waiting for decode_hit's edlib tie-break (made inside vm_align_batch) has no path at this stage entry
    int64_t npaths = 0, nrows = 0;
    for (int64_t r = 0; r < n; ++r) { npaths += h_np[r]; for (int p = 0; p < h_np[r]; ++p) nrows += h_plen[aoff[r] + p]; }
    if (n) { VMX_TRY(download(out->fast_used, c->b[24].p, (size_t)n, c->stream)); VMX_HIP(hipStreamSynchronize(c->stream)); }
    out->read_path_off = host_alloc<int64_t>((size_t)n + 1); out->path_off = host_alloc<int64_t>((size_t)npaths + 1);
    out->path_anchors = host_alloc<int64_t>((size_t)nrows * 4);
    int64_t pi = 0, ro = 0;
    for (int64_t r = 0; r < n; ++r) {
        out->read_path_off[r] = pi;
        int64_t src = aoff[r];
        for (int p = 0; p < h_np[r]; ++p) {
            out->path_off[pi++] = ro;
            for (int x = 0; x < h_plen[aoff[r] + p]; ++x) {
                const vmx_anchor& a = h_prow[src++];
                int64_t* o = out->path_anchors + 4 * ro++;
                o[0] = a.q; o[1] = a.r; o[2] = a.s; o[3] = a.l;
            }
        }
    }
    out->read_path_off[n] = pi; out->path_off[pi] = ro;
    return VM_OK;
}

}  // extern "C"
