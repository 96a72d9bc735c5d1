// vmx_host.h — host-side internals of libvacmapx (context, buffers, kernel prototypes).
#ifndef VMX_HOST_H
#define VMX_HOST_H
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "../../include/vacmapx.h"
#include <algorithm>
#include <map>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>
#include <mutex>
#include <execinfo.h>

namespace vmx {

void set_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define VMX_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return vmx::hip_fail(_e, #expr, __FILE__, __LINE__);  \
    } while (0)
#define VMX_TRY(expr) do { int _rc = (expr); if (_rc < 0) return _rc; } while (0)

// grow-only device buffer
// growth statistics of the grow-only pools (tuning aid, printed with VMX_DBG_POOLS): a growth is a hipFree — which waits for the whole
// device, every other context's batch included — plus a hipMalloc
struct DevBufStats { std::atomic<long long> grows{0}, first{0}, ns{0}; };
inline DevBufStats& devbuf_stats() { static DevBufStats s; return s; }

// A buffer that has to grow is not freed on the spot: hipFree waits for the WHOLE device — every other context's batch in flight — and a
// run's first windows grow a few hundred buffers (each new longest read resizes the per-slot pools). The outgrown allocations are parked
// here and freed together once they add up to VMX_RETIRE_BYTES (one device-wide wait instead of hundreds), when an allocation fails, or
// when a context is destroyed.
struct DevBufRetired {
    std::mutex m; std::vector<void*> v; size_t bytes = 0;
    void flush_locked() { for (void* q : v) (void)hipFree(q); v.clear(); bytes = 0; }
    void flush() { std::lock_guard<std::mutex> g(m); flush_locked(); }
    void park(void* q, size_t cap) {
        std::lock_guard<std::mutex> g(m);
        v.push_back(q); bytes += cap;
        if (bytes > ((size_t)6 << 30) || v.size() >= 512) flush_locked();
    }
};
inline DevBufRetired& devbuf_retired() { static DevBufRetired r; return r; }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // grows with head-room so that slightly larger batches do not reallocate: 1/8 on the first allocation, 1/2 (at most 2 GB) when the
    // buffer has to grow again — the input has shown that it varies
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return 0;
        const auto t0 = std::chrono::steady_clock::now();
        const bool regrow = p != nullptr;
        if (p) { devbuf_retired().park(p, cap); p = nullptr; cap = 0; }
        size_t want = bytes + (regrow ? std::min<size_t>(bytes / 2, (size_t)2 << 30) : std::min<size_t>(bytes / 8, (size_t)1 << 30)) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); devbuf_retired().flush(); e = hipMalloc(&p, want); }                // give the parked memory back first
        if (e != hipSuccess && regrow) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&p, want); }          // no room for the head-room: exact size
        if (e != hipSuccess) { p = nullptr; if (getenv("VMX_DBG_POOLS")) { fprintf(stderr, "[pools] hipMalloc of %.3f GB (asked: %.3f GB) failed\n", want / 1e9, bytes / 1e9); void* bt[24]; const int nb = backtrace(bt, 24); backtrace_symbols_fd(bt, nb, 2); } return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
        cap = want;
        DevBufStats& st = devbuf_stats(); (regrow ? st.grows : st.first)++;
        const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        st.ns += ns;
        static const bool verbose = getenv("VMX_DBG_POOLS") && atoi(getenv("VMX_DBG_POOLS")) >= 2;
        if (verbose && ns > 2000000) fprintf(stderr, "[pools] %s of %.3f GB took %.1f ms\n", regrow ? "re-allocation" : "allocation", want / 1e9, ns * 1e-6);
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct HostTables {
    std::vector<float> extra, readgap_h, readgap_r, large_readgap;
    std::vector<double> log2cache, log2int;
};
const HostTables& host_tables();

// VMX_DBG_COPIES=1: the copies of the helpers below by call site (file line), printed at exit — a tuning aid: which of a batch's ~75 small copies to go after
struct CopyCensus {
    std::mutex m; std::map<std::pair<std::string, int>, std::pair<long long, long long>> by_line; bool on = getenv("VMX_DBG_COPIES") != nullptr;
    void add(const char* file, int line, size_t bytes) { if (!on) return; std::lock_guard<std::mutex> g(m); const char* b = strrchr(file, '/'); auto& e = by_line[std::make_pair(std::string(b ? b + 1 : file), line)]; e.first++; e.second += (long long)bytes; }
    ~CopyCensus() { if (!on) return; for (auto& kv : by_line) fprintf(stderr, "[copies] %s:%d (%s): %lld copies, %lld bytes\n", kv.first.first.c_str(), kv.first.second < 0 ? -kv.first.second : kv.first.second, kv.first.second < 0 ? "download" : "upload", kv.second.first, kv.second.second); }
};
inline CopyCensus& copy_census() { static CopyCensus c; return c; }
inline long long& download_wait_ns() { static thread_local long long v = 0; return v; }      // per host thread = per context in flight
template <class T> int upload(DevBuf& b, const T* host, size_t n, hipStream_t st, int line = __builtin_LINE(), const char* file = __builtin_FILE()) {
    VMX_TRY(b.reserve(sizeof(T) * (n ? n : 1)));
    if (n) { VMX_HIP(hipMemcpyAsync(b.p, host, sizeof(T) * n, hipMemcpyHostToDevice, st)); copy_census().add(file, line, sizeof(T) * n); }
    return 0;
}
template <class T> int download(T* host, const void* dev, size_t n, hipStream_t st, int line = __builtin_LINE(), const char* file = __builtin_FILE()) {
    if (n) {
        // (a copy into pageable host memory returns when the data has arrived, i.e. after everything queued on the stream before it: these calls are where a
        // batch's host thread really waits for the GPU — counted with vmx_stream_sync's time in the tuning counters)
        const auto t0 = std::chrono::steady_clock::now();
        VMX_HIP(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, st)); copy_census().add(file, -line, sizeof(T) * n);
        download_wait_ns() += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

}  // namespace vmx

// ---- the context's MAILBOX (round 6): how the host thread of a batch waits and exchanges its small data with the device WITHOUT a HIP wait call.
// Measured in round 5: hipStreamSynchronize, a blocking event's hipEventSynchronize and the runtime's copies to / from pageable memory all SPIN on this
// runtime — one busy core per batch in flight (5.9 cores for six contexts), 47-66 spinning cores for eight ranks. The mailbox is one page-locked,
// device-mapped host block per context:
//   * a sequence word. vmx_stream_sync() launches k_signal(word, ++seq) on the stream and polls the word with nanosleep between looks: the thread is
//     ASLEEP while the GPU works (VMX_POLL_US, default 40 microseconds between looks; hipStreamQuery twice a second to notice a dead stream);
//   * an upload ring: vmx_push() copies a small host vector into the ring and queues an asynchronous copy from there (page-locked source: the call
//     returns at once and the source vector may die); the ring restarts after every completed wait (everything queued before it has been consumed);
//   * a download area: vmx_fetch() queues an asynchronous copy into the area (page-locked destination: no wait inside the call) and notes where the bytes
//     have to go; the next completed wait copies them there. Large results (records, CIGAR text: tens of MB per batch) land in a second block that grows.
// VMX_WAIT_MODE=spin (or vm_ctx_set_blocking_sync(ctx, 0)) keeps hipStreamSynchronize and pageable copies for A/B runs.
struct vmx_mailbox {
    char* h = nullptr; char* d = nullptr;                 // the block as the host / the device address it
    size_t up_off = 0, up_cap = 0, up_used = 0;           // upload ring
    size_t dn_off = 0, dn_cap = 0, dn_used = 0;           // download area
    char* big = nullptr; size_t big_cap = 0, big_used = 0; // page-locked landing block of the large results (grows)
    unsigned long long seq = 0;
    struct Pending { void* dst; const char* src; size_t bytes; const void* dev; };      // dev != null: not copied yet — the export kernel of the next wait moves it
    std::vector<Pending> pend;
    char* big_d = nullptr;                                 // the landing block as the device addresses it
    unsigned long long* done_ctr = nullptr;                // device word: workgroups of the running export kernel that have finished
    bool on = false;                                       // false: legacy waits (spinning)
    long long poll_ns = 40000;
};
__global__ void k_signal(unsigned long long* word, unsigned long long v);
// Round 6, second step: the downloads of a wait are not copies at all. vmx_fetch only NOTES (device source, landing place); the wait launches ONE kernel that
// moves every noted piece into the page-locked landing area over the bus (16 B per lane, all pieces side by side) and — its last workgroup to finish — writes
// the sequence word behind them: one launch instead of a runtime copy per piece plus the signal (a batch made ~40 such copies).
#define VMX_EXPORT_MAX 12
struct vmx_export_args { const char* src[VMX_EXPORT_MAX]; char* dst[VMX_EXPORT_MAX]; unsigned long long bytes[VMX_EXPORT_MAX]; int n; unsigned long long* word; unsigned long long seq; unsigned long long* done; };
__global__ void k_export(vmx_export_args A);

enum { VMX_NBUF = 64 };
struct vm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    vmx_tables tables{};
    vmx::DevBuf tab_buf;
    vmx::DevBuf b[VMX_NBUF];     // scratch buffers reused by the entry points (grow-only)
    int num_cu = 256;
    int inflight = 1;                            // contexts sharing the GPU (vm_ctx_set_inflight)
    hipEvent_t sync_ev = nullptr;                // (legacy, unused since round 6: a blocking event's wait spins on this runtime)
    vmx_mailbox mb;                              // sleeping waits + page-locked exchange of the small data (above)
    hipEvent_t ev[24];
    hipEvent_t gev[48];                          // gap-fill chunk events: [redo][chunk 0..7][before fill, after fill, after trace]
    int n_gev[2] = {0, 0};                        // chunks recorded by the last batch per pass
    hipEvent_t kev[4] = {nullptr, nullptr, nullptr, nullptr};   // around k_local_seed's main launch [0,1] and the clustering kernels [2,3] of the last batch
    int kev_set = 0;                              // bit 0: [0,1] recorded, bit 1: [2,3] recorded
    int64_t n_bandfall = 0;                       // reads k_local_seed_band handed back to k_local_seed (reset per batch)
    int ext_mul = 1;                              // extend-stage pool multiplier of the running call (grow-and-retry in align_device)
    long long redo_need_max = 0;                  // largest full-matrix traceback need of one gap-fill chunk this context has seen (sizes the second launch's pool)
    long long geo_cap[4] = {0, 0, 0, 0}; bool geo_valid = false;      // what the extend-stage pools of this context hold (segment anchors, segments, record text, problem slots): the room of the batches that do not ask
    bool run_pass1 = false;                       // this call runs pass 1 of the extend stage (the nofilter re-run; side batches of align_device)
    int32_t* rc_cur = nullptr; int rc_next = 0;    // the running extend-stage phase's problem counter inside the batch's counter block (64 counters, cleared once per batch)
    bool force_exact = false;                     // this call launches the exact edit-distance tier whatever the banded tiers left (side batches of align_device)
    long long round_epoch = 0;                    // launches of k_round_prep by this context (k_round.hip: its publication flags carry the launch number)
    int64_t n_syncs = 0;                          // host waits on this context's stream (reset per batch)
    int64_t sync_wait_ns = 0, call_t0_ns = 0;     // time inside those waits; wall clock at the start of the batch (tuning: ms_stage[14] / [15])
    double res_rec_per_read = 0.0, res_blob_per_base = 0.0;   // largest records per read / CIGAR bytes per base a batch of this context produced (result copy size guess)
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};   // side streams for independent launches (LDS-bucketed kernels)
    hipEvent_t fork_ev = nullptr, join_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t low = nullptr;                   // low-priority stream for the long VALU-bound launches (the gap fill's first launch): see vmx_lowprio_begin
    hipEvent_t low_ev[2] = {nullptr, nullptr};
    int64_t last_n_minimizers = 0;               // of the last seed stage (stats)
    struct vmx_local_bufs* lbufs = nullptr;      // vmx_stage.h
    struct vmx_extend_bufs* ebufs = nullptr;
    struct vmx_batch_bufs* bbufs = nullptr;
};

// With several batches in flight a batch is made of ~170 small launches (a few to a few hundred microseconds each) around a handful of long VALU-bound
// ones, and what stretches it from 27 ms alone to 77 ms with four others is the small launches queueing behind the other batches' long ones (seed stage
// 2.6 -> 12.8 ms, edge extension 0.8 -> 5.4 ms). The contexts' streams are created at the device's HIGHEST priority and the long launches go to a stream
// of the LOWEST: when a workgroup of a long launch retires, the waiting small launch should be dispatched first. MEASURED (round 5, two alternating
// pairs of the default bench): 16.57 / 16.93 ms per step with the priorities against 16.45 / 15.51 without — the hardware queues' priority does not
// shorten the small launches' wait here. Off unless VMX_STREAM_PRIO=1.
static inline bool vmx_stream_prio_on() { static const bool on = [] { const char* e = getenv("VMX_STREAM_PRIO"); return e && atoi(e) != 0; }(); return on; }
struct vmx_lowprio {
    vm_ctx* c; bool on;
    explicit vmx_lowprio(vm_ctx* ctx) : c(ctx), on(ctx->low != nullptr) { if (on) { (void)hipEventRecord(c->low_ev[0], c->stream); (void)hipStreamWaitEvent(c->low, c->low_ev[0], 0); } }
    hipStream_t stream() const { return on ? c->low : c->stream; }
    void join() { if (on) { (void)hipEventRecord(c->low_ev[1], c->low); (void)hipStreamWaitEvent(c->stream, c->low_ev[1], 0); on = false; } }
};

// wait for the context's main stream (and finish the queued vmx_fetch copies). Default since round 6: the mailbox's sleeping wait (above); legacy:
// hipStreamSynchronize (the runtime spins: one busy core per waiting thread).
int vmx_mailbox_wait(vm_ctx* c);                 // vmx_capi.hip
static inline hipError_t vmx_stream_sync(vm_ctx* c) {
    ++c->n_syncs;
    const auto t0 = std::chrono::steady_clock::now();            // (time spent waiting: ms_stage[14] of the batch; the rest of the call's wall time is host work with the stream empty)
    hipError_t e = hipSuccess;
    if (c->mb.on) { if (vmx_mailbox_wait(c) < 0) e = hipErrorUnknown; }
    else e = hipStreamSynchronize(c->stream);
    c->sync_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}
// small host vector(s) -> device buffer(s) through the upload ring: the data is copied into the ring (page-locked, device-mapped) and ONE kernel (k_import) reads it
// from there over the bus into its place(s) — asynchronous, `host` may be reused at once. (A runtime copy from the ring cost two or three blit launches per piece:
// head, body, tail of an unaligned size. 48 copy launches per batch for 14 pieces.) Falls back to a plain copy when the ring is full / off.
struct vmx_import_args { const char* src[4]; char* dst[4]; unsigned long long bytes[4]; int n; };
__global__ void k_import(vmx_import_args A);
struct vmx_push_piece { void* dev; const void* host; size_t bytes; };
static inline int vmx_push_pieces(vm_ctx* c, const vmx_push_piece* pc, int np) {
    vmx_mailbox& m = c->mb;
    static const bool by_copy = getenv("VMX_PUSH_COPIES") != nullptr;        // A/B knob: runtime copies from the ring
    size_t at = (m.up_used + 63) & ~(size_t)63, need = 0;
    for (int i = 0; i < np; ++i) need += (pc[i].bytes + 63) & ~(size_t)63;
    if (!m.on || at + need > m.up_cap) { for (int i = 0; i < np; ++i) if (pc[i].bytes) VMX_HIP(hipMemcpyAsync(pc[i].dev, pc[i].host, pc[i].bytes, hipMemcpyHostToDevice, c->stream)); return 0; }
    vmx_import_args A; memset(&A, 0, sizeof A);
    unsigned long long tot = 0;
    for (int i = 0; i < np; ++i) {
        if (!pc[i].bytes) continue;
        memcpy(m.h + m.up_off + at, pc[i].host, pc[i].bytes);
        if (by_copy) VMX_HIP(hipMemcpyAsync(pc[i].dev, m.h + m.up_off + at, pc[i].bytes, hipMemcpyHostToDevice, c->stream));
        else { A.src[A.n] = m.d + m.up_off + at; A.dst[A.n] = (char*)pc[i].dev; A.bytes[A.n] = pc[i].bytes; ++A.n; tot += pc[i].bytes; }
        at += (pc[i].bytes + 63) & ~(size_t)63;
    }
    m.up_used = at;
    if (A.n) hipLaunchKernelGGL(k_import, dim3((unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>((tot + 16383ULL) >> 14, 32ULL))), dim3(256), 0, c->stream, A);
    return 0;
}
template <class T> int vmx_push(vm_ctx* c, vmx::DevBuf& b, const T* host, size_t n, int line = __builtin_LINE(), const char* file = __builtin_FILE()) {
    const size_t bytes = sizeof(T) * n;
    if (!c->mb.on || !n) return vmx::upload(b, host, n, c->stream, line, file);
    VMX_TRY(b.reserve(bytes));
    vmx::copy_census().add(file, line, bytes);
    const vmx_push_piece pc{b.p, host, bytes};
    return vmx_push_pieces(c, &pc, 1);
}
// up to four vectors in one launch
struct vmx_push_req { vmx::DevBuf* b; const void* host; size_t bytes; };
static inline int vmx_push_many(vm_ctx* c, const vmx_push_req* rq, int np) {
    vmx_push_piece pc[4];
    for (int i = 0; i < np; ++i) { VMX_TRY(rq[i].b->reserve(rq[i].bytes ? rq[i].bytes : 1)); pc[i] = vmx_push_piece{rq[i].b->p, rq[i].host, rq[i].bytes}; }
    return vmx_push_pieces(c, pc, np);
}
// the same into a raw device address
template <class T> int vmx_push_to(vm_ctx* c, void* dev, const T* host, size_t n) {
    if (!n) return 0;
    const vmx_push_piece pc{dev, host, sizeof(T) * n};
    return vmx_push_pieces(c, &pc, 1);
}
// device -> host, delivered by the NEXT vmx_stream_sync(c) (the caller must not look at `host` before). Falls back to the plain (waiting) copy when off.
int vmx_fetch_bytes(vm_ctx* c, void* host, const void* dev, size_t bytes);      // vmx_capi.hip
template <class T> int vmx_fetch(vm_ctx* c, T* host, const void* dev, size_t n, int line = __builtin_LINE(), const char* file = __builtin_FILE()) {
    if (!n) return 0;
    if (!c->mb.on) return vmx::download(host, dev, n, c->stream, line, file);
    vmx::copy_census().add(file, -line, sizeof(T) * n);
    return vmx_fetch_bytes(c, (void*)host, dev, sizeof(T) * n);
}

// fork/join of independent launches over the context's side streams (all ordered after / before the main stream)
struct vmx_fork {
    vm_ctx* c; int used = 0;
    explicit vmx_fork(vm_ctx* ctx) : c(ctx) { (void)hipEventRecord(c->fork_ev, c->stream); }
    hipStream_t next() {
        hipStream_t s = c->aux[used & 3];
        if (used < 4) (void)hipStreamWaitEvent(s, c->fork_ev, 0);
        ++used;
        return s;
    }
    void join() {
        const int n = used < 4 ? used : 4;
        for (int i = 0; i < n; ++i) { (void)hipEventRecord(c->join_ev[i], c->aux[i]); (void)hipStreamWaitEvent(c->stream, c->join_ev[i], 0); }
        used = 0;
    }
};

// ---- kernels (k_dp.hip, k_chain.hip, ...) ----
__global__ void k_encode(const char* in, uint8_t* out, int64_t n);
__global__ void k_edit_distance(const uint8_t* qcodes, const int64_t* q_off, const uint8_t* tcodes, const int64_t* t_off,
                                int8_t* carry_pool, const int32_t* order, const int32_t* range, int32_t* counters, int which, int64_t* out,
                                int64_t carry_stride, int32_t* oflow);
__global__ void k_size_order(const int64_t* size, const int32_t* n_ptr, int64_t thresh, int32_t* order, int32_t* range, int32_t* counters);
__global__ void k_size_hist(const int64_t* size, const int32_t* n_ptr, int64_t thresh, int32_t* ghist);
__global__ void k_size_scatter(const int64_t* size, const int32_t* n_ptr, const int32_t* ghist, int32_t* gcur, int32_t* order, int32_t* range, int32_t* counters);
// k_size_order's result with the whole device (k_ed.hip): n_upper = a host-side bound on *n_ptr (grid size); scratch: 513 ints
static inline void vmx_size_order_wide(vm_ctx* c, const int64_t* size, const int32_t* n_ptr, int64_t n_upper, int64_t thresh, int32_t* order, int32_t* range, int32_t* counters,
                                       int32_t* scratch) {
    static const bool one_wg = getenv("VMX_SIZE_ORDER_ONE") != nullptr;          // A/B knob: the one-workgroup kernel of round 3
    if (one_wg) { hipLaunchKernelGGL(k_size_order, dim3(1), dim3(1024), 0, c->stream, size, n_ptr, thresh, order, range, counters); return; }
    (void)hipMemsetAsync(scratch, 0, 4 * 513, c->stream);
    const unsigned G = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_upper + 1023) / 1024, (int64_t)c->num_cu * 2));
    hipLaunchKernelGGL(k_size_hist, dim3(G), dim3(256), 0, c->stream, size, n_ptr, thresh, scratch);
    hipLaunchKernelGGL(k_size_scatter, dim3(G), dim3(256), 0, c->stream, size, n_ptr, (const int32_t*)scratch, scratch + 257, order, range, counters);
}
__global__ void k_ed_banded(const uint8_t* qcodes, const int64_t* q_off, const uint8_t* tcodes, const int64_t* t_off, const int32_t* order,
                            const int32_t* range, int32_t* counter, int64_t* ub_out);
struct vmx_ext_args;
__global__ void k_ed_banded4(const uint8_t* qcodes, const int64_t* q_off, const uint8_t* tcodes, const int64_t* t_off, const int32_t* order,
                             const int32_t* range, int32_t* counter, int64_t* ub_out);
__global__ void k_ed_flag(const int64_t* ub, const int64_t* q_off, const int64_t* t_off, const int32_t* n_ptr, double maxdiv, int64_t* sizes,
                          int64_t* ed_out, int32_t* n_flagged, int first, const int32_t* prob_read, struct vmx_ext_read* er);
__global__ void k_extend(const uint8_t* tcodes, const int64_t* t_off, const uint8_t* qcodes, const int64_t* q_off, int n_prob,
                         int match, int mismatch, int o, int e, int bw_in, int zdrop, int32_t* out_te, int32_t* out_qe, int32_t* out_sc, const int32_t* n_ptr);
__global__ void k_gapfill_fill(const uint8_t* tcodes, const uint8_t* qcodes, const vmx_dp_prob* probs, int n_prob, int match,
                               int mismatch, int o1, int e1, int o2, int e2, uint8_t* tb_pool, int32_t* bnd_pool, int32_t* out_score,
                               const int32_t* order, int32_t* counter);
__global__ void k_gapfill_fill_ns(const uint8_t* tcodes, const uint8_t* qcodes, vmx_dp_prob* probs, int n_prob, int match,
                                  int mismatch, int o1, int e1, int o2, int e2, uint8_t* tb_pool, int32_t* bnd_pool, int32_t* out_score,
                                  const int32_t* order, const int32_t* range, int32_t* counter, int32_t* redo_list, int32_t* redo_cnt, int redo_pass, int ad_pct,
                                  uint8_t* redo_pool, unsigned long long* redo_bytes, const int32_t* n_ptr = nullptr, unsigned long long redo_cap = ~0ULL, int tb_by_ns = 0);
// band-width rule of the anti-diagonal gap fill (vmx_ad_ns): pct | pct_min << 16; tuning knobs VMX_AD_PCT / VMX_AD_PCT_MIN
// (round 4: the default follows the read mode — HiFi problems score near the all-match bound, so a band whose margin covers 40 % of the problem
// is proven as often as one that covers 100 %: fill 7.0 -> 6.0 ms per batch; ONT modes 100 -> 90: 8.9 -> 8.6; profiles/r04_x_*)
static inline int vmx_ad_pct_env(int mode = 0) {
    int pct = mode == 1 ? 40 : 90, pmin = mode == 1 ? 40 : VMX_AD_PCT_MIN_DEFAULT;
    if (const char* e = getenv("VMX_AD_PCT")) { const int v = atoi(e); if (v >= 0 && v <= 60000) pct = v; }
    if (const char* e = getenv("VMX_AD_PCT_MIN")) { const int v = atoi(e); if (v >= 0 && v <= 60000) pmin = v; }
    return pct | (pmin << 16);
}
__global__ void k_gapfill_trace(const uint8_t* tcodes, const uint8_t* qcodes, const vmx_dp_prob* probs, int n_prob, int eqx,
                                const uint8_t* tb_pool, uint32_t* run_pool, char* cig_pool, int32_t* cig_len, const int32_t* band_flag, const uint8_t* redo_pool, int spread, int32_t* cig_q, const int32_t* n_ptr = nullptr);
__global__ void k_flip_sort(const int64_t* rows, const int64_t* aoff, const int64_t* readlens, int n_reads, uint64_t* key_pool,
                            const int64_t* key_off, vmx_anchor* sorted, int32_t* need_reverse);
__global__ void k_chain_global(const vmx_anchor* anchors, const int64_t* aoff, const int32_t* rlist, int nlist, int lds_cap,
                               vmx_tables tab, const double* gapcost_list, double oskipcost, int omaxdiff, int maxgap,
                               double* S_out, int32_t* P_out, int32_t* SA_out, uint8_t* cov_pool, int64_t* gmax_out, int64_t* opcount_out, int rmode,
                               double* FP_pool, double* PP_pool);
// k_chain_rows.hip: four reads per wavefront (one per 16-lane row); VMX_CHAIN_ROWS=0 keeps the one-wavefront-per-read kernels for A/B
__global__ void k_chain_global_rows(const vmx_anchor* anchors, const int64_t* aoff, const int32_t* rlist, int nlist, vmx_tables tab, const double* gapcost_list,
                                    double oskipcost, int omaxdiff, int maxgap, double* S_out, int32_t* P_out, int32_t* SA_out, uint8_t* cov_pool,
                                    int64_t* gmax_out, int64_t* opcount_out, int rmode, double* FP_pool, double* PP_pool, unsigned long long* dbg);
__global__ void k_chain_global_rows_w3(const vmx_anchor* anchors, const int64_t* aoff, const int32_t* rlist, int nlist, vmx_tables tab, const double* gapcost_list,
                                       double oskipcost, int omaxdiff, int maxgap, double* S_out, int32_t* P_out, int32_t* SA_out, uint8_t* cov_pool,
                                       int64_t* gmax_out, int64_t* opcount_out, int rmode, double* FP_pool, double* PP_pool, unsigned long long* dbg);
// VMX_RW_WIN=3 (tests, read at every launch): the 3-entry-window TEST kernels instead of the product's 16-entry ones (k_chain_rows.hip)
static inline bool vmx_chain_rows_win3() { const char* e = getenv("VMX_RW_WIN"); return e && atoi(e) == 3; }
// eight device counters of the row kernels (anchors, scans that left the window, insertions through HBM, opcount; global / local): one block for the whole
// library (vmx_capi.hip), switched on by VMX_DBG_CHAIN=1 (printed by vmx_chain_dbg_report(): a tuning aid) or by vm_debug_chain_counters(1, ...) (the GPU
// tests assert them); never on by default
unsigned long long* vmx_chain_dbg();
void vmx_chain_dbg_report(hipStream_t st);
static inline bool vmx_chain_rows_on() { static const bool on = [] { const char* e = getenv("VMX_CHAIN_ROWS"); return !e || atoi(e) != 0; }(); return on; }
__global__ void k_chain_global_fast(const vmx_anchor* anchors, const int64_t* aoff, int n_reads, const int64_t* roff, vmx_tables tab,
                                    const double* gapcost_list, double oskipcost, int omaxdiff, int maxgap, double* S_out, int32_t* P_out, int32_t* SA_out,
                                    uint8_t* cov_pool, int32_t* si_pool, int64_t* t_pool, int32_t* cnt_pool, int64_t* gmax_out, int32_t* ran, int rmode,
                                    double* FP_pool, double* PP_pool);
__global__ void k_chain_select(const vmx_anchor* anchors, const int64_t* aoff, const int64_t* readlens, const int32_t* rlist, int nlist, int lds_cap,
                               const double* S, const int32_t* P, const int32_t* SA, const int64_t* gmax, const int32_t* need_reverse, int mode,
                               char* scratch, const int64_t* scratch_off, int32_t* out_mapq, double* out_score, int32_t* out_npaths,
                               int32_t* out_path_len, vmx_anchor* out_path_anchors);

#endif
