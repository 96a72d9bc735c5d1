// k_round.hip — ONE launch per DP round of the extend stage for everything between "the phase kernel published the round's problem descriptors" and
// "the strings are gathered" (round 6; the counterpart of the Python-side list building around every k_cigar / edlib call of extend_func,
// /root/reference/src/vacmap/mammap_clrnano.py:19238-19330 — the reference sizes nothing: it slices strings per call).
//
// Before: k_desc_lens, k_stat_put, two three-launch scans (string offsets) and — in the gap-fill rounds — k_dp_sizes, four more three-launch scans,
// k_tb_plan and k_dp_table: 9 launches per x-drop / edit-distance round, 24 per gap-fill round, ~110 of a batch's ~170 launches, each a few
// microseconds of work that queued behind the other batches' long kernels (VERDICT r5 Weak 5). Now k_round_prep<DP>: every workgroup takes a
// contiguous slice of the round's problems (count read on the device), computes the lengths / pool sizes from the descriptors, publishes the slice's
// sums, adds up its predecessors' sums (look-back: a workgroup only ever waits for workgroups dispatched BEFORE it, which need nothing from it), scans
// its slice and writes the offsets — and, gap-fill rounds, the problem table. The workgroup that finishes last writes the totals, the count for the
// host's statistics block and the traceback chunk plan.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_extend.h"
#include "vmx_round.h"

#ifdef VMX_EMU
#define VMX_LD_ACQ(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define VMX_ST_REL(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define VMX_LD_RLX(p) __atomic_load_n((p), __ATOMIC_RELAXED)
#define VMX_ADD_ACQREL(p, v) __atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL)
#else
#define VMX_LD_ACQ(p) __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define VMX_ST_REL(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define VMX_LD_RLX(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VMX_ADD_ACQREL(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
#endif

template <bool DP> struct vmx_round_vals { static constexpr int NV = DP ? 6 : 2; };

// slot width of a small-class problem's own band (0: no band is tried, the problem is filled in full by the second launch out of its pool)
__device__ __forceinline__ int vmx_round_w(const vmx_round_args& A, long long tl, long long ql) {
    return VMX_AD_W(vmx_ad_ns((int)tl, (int)ql, A.ad_match, A.ad_o1, A.ad_e1, A.ad_o2, A.ad_e2, A.ad_pct & 0xffff, (A.ad_pct >> 16) & 0xffff));
}
template <bool DP>
__device__ __forceinline__ void vmx_round_values(const vmx_round_args& A, const vmx_pair_desc& d, long long* v) {
    const long long tl = d.t.len, ql = d.q.len;
    v[0] = tl; v[1] = ql;
    if constexpr (DP) {
        const bool small = tl > 0 && ql > 0 && VMX_DP16X4_OK(tl, ql);
        v[2] = (small && A.ad_on) ? VMX_AD_TB_BYTES_W(tl, ql, vmx_round_w(A, tl, ql)) : VMX_TB_BYTES_NS(tl, ql);
        v[3] = 3 * (ql + 1); v[4] = tl + ql + 2; v[5] = 2 * (tl + ql) + 16;      // (k_dp_sizes of round 1-5)
    }
}
// the queue key of a gap-fill problem (vmx_round.h)
__device__ __forceinline__ long long vmx_round_key(const vmx_round_args& A, long long tl, long long ql) {
    if (tl <= 0 || ql <= 0) return 0;
    if (!VMX_DP16X4_OK(tl, ql)) {
        const long long b = VMX_TB_BYTES(tl, ql);
        if (b > VMX_HEAD_THRESH) return (b < (1LL << 38) ? b : (1LL << 38)) << 24;
        return b << 23;
    }
    const long long b64 = VMX_AD_TB_BYTES(tl, ql);                       // (tl + ql) * 64: 2^7 .. 2^16
    if (!A.ad_on) return b64;
    const int w = vmx_round_w(A, tl, ql);
    return w == 4 ? b64 << 16 : (w == 2 ? b64 << 6 : (w == 1 ? b64 >> 4 : (b64 >> 14) + 1));
}

// the traceback chunk plan of a gap-fill round (k_tb_plan of rounds 2-5, unchanged): chunks of at most `limit` traceback bytes, found by bisection on tboff
__device__ void vmx_round_plan(const vmx_round_args& A, int n, const long long* tot) {
    int64_t* out = A.plan_out;
    const int64_t* tboff = A.off[2];
    out[0] = n; out[1] = tot[2]; out[2] = tot[3]; out[3] = tot[4]; out[4] = tot[5]; out[5] = tot[0]; out[6] = tot[1];
    int64_t* cuts = out + 8; int64_t* offs = out + 8 + VMX_MAX_CHUNKS + 1;
    int m = 0, p = 0;
    cuts[0] = 0; offs[0] = 0;
    while (p < n) {
        const int64_t base = VMX_LD_RLX(tboff + p);
        int lo = p + 1, hi = n;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; const int64_t tv = mid == n ? (int64_t)tot[2] : VMX_LD_RLX(tboff + mid); if (tv - base <= A.tb_limit) lo = mid; else hi = mid - 1; }
        p = lo; ++m;
        if (m > VMX_MAX_CHUNKS) { m = -1; break; }
        cuts[m] = p; offs[m] = p == n ? (int64_t)tot[2] : VMX_LD_RLX(tboff + p);
    }
    out[7] = m;
}

template <bool DP>
__global__ void __launch_bounds__(256) k_round_prep(vmx_round_args A) {
    constexpr int NV = vmx_round_vals<DP>::NV;
    __shared__ long long s_w[4][NV];
    __shared__ long long s_base[NV];
    __shared__ long long s_tot[NV];
    __shared__ int s_last;
    const int n = *A.n_prob;
    const int G = (int)gridDim.x, b = (int)blockIdx.x, tid = (int)threadIdx.x, lane = vmx_lane(), wv = tid >> 6;
    const long long chunk = ((long long)n + G - 1) / G;
    long long lo = (long long)b * chunk; if (lo > n) lo = n;
    const long long hi = lo + chunk < n ? lo + chunk : (long long)n;
    long long* part = (long long*)A.part + (size_t)b * 8;
    // ---- slice sums
    long long s[NV];
    for (int v = 0; v < NV; ++v) s[v] = 0;
    for (long long i = lo + tid; i < hi; i += 256) { long long x[NV]; vmx_round_values<DP>(A, A.desc[i], x); for (int v = 0; v < NV; ++v) s[v] += x[v]; }
    for (int v = 0; v < NV; ++v) { s[v] = vmx_wave_sum_i64(s[v]); if (lane == 0) s_w[wv][v] = s[v]; }
    __syncthreads();
    if (tid == 0) {
        for (int v = 0; v < NV; ++v) { const long long t = s_w[0][v] + s_w[1][v] + s_w[2][v] + s_w[3][v]; s_tot[v] = t; part[v] = t; }
        VMX_ST_REL(part + 7, (long long)A.epoch);               // the slice's sums are published (release: they are visible before the flag)
    }
    // ---- look-back: thread t adds up the predecessors t, t + 256, ... (G <= 256 as launched: one each)
    for (int v = 0; v < NV; ++v) s[v] = 0;
    for (int t = tid; t < b; t += 256) {
        long long* pp = (long long*)A.part + (size_t)t * 8;
        while (VMX_LD_ACQ(pp + 7) != (long long)A.epoch) { }
        for (int v = 0; v < NV; ++v) s[v] += VMX_LD_RLX(pp + v);
    }
    __syncthreads();
    for (int v = 0; v < NV; ++v) { s[v] = vmx_wave_sum_i64(s[v]); if (lane == 0) s_w[wv][v] = s[v]; }
    __syncthreads();
    if (tid == 0) for (int v = 0; v < NV; ++v) s_base[v] = s_w[0][v] + s_w[1][v] + s_w[2][v] + s_w[3][v];
    __syncthreads();
    // ---- the slice's exclusive scans, 256 problems per step
    for (long long i0 = lo; i0 < hi; i0 += 256) {
        const long long i = i0 + tid;
        long long x[NV], inc[NV];
        vmx_pair_desc d; d.t.len = 0; d.q.len = 0;
        if (i < hi) d = A.desc[i];
        if (i < hi) vmx_round_values<DP>(A, d, x); else for (int v = 0; v < NV; ++v) x[v] = 0;
        for (int v = 0; v < NV; ++v) {
            long long c = x[v];
            for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(c, o); if (lane >= o) c += y; }
            inc[v] = c;
            if (lane == 63) s_w[wv][v] = c;
        }
        __syncthreads();
        long long off[NV];
        for (int v = 0; v < NV; ++v) { long long wb = 0; for (int w = 0; w < wv; ++w) wb += s_w[w][v]; off[v] = s_base[v] + wb + inc[v] - x[v]; }
        if (i < hi) {
            for (int v = 0; v < NV; ++v) A.off[v][i] = off[v];
            if constexpr (DP) {
                A.tb_size[i] = vmx_round_key(A, x[0], x[1]);
                vmx_dp_prob p; p.t_off = off[0]; p.q_off = off[1]; p.tl = (int32_t)x[0]; p.ql = (int32_t)x[1]; p.tb_off = off[2]; p.bnd_off = off[3]; p.run_off = off[4]; p.cig_off = off[5];
                if (A.cap[0] > 0 && (off[2] + x[2] > A.cap[0] || off[3] + x[3] > A.cap[1] || off[4] + x[4] > A.cap[2] || off[5] + x[5] > A.cap[3])) {
                    p.tl = 0; p.ql = 0; p.t_off = 0; p.q_off = 0; p.tb_off = 0; p.bnd_off = 0; p.run_off = 0; p.cig_off = 0; A.tb_size[i] = 0;      // (run / CIGAR slot 0 of the pools: 2 words / 16 bytes are always there)
                }
                A.probs[i] = p;
            }
        }
        __syncthreads();
        if (tid == 0) for (int v = 0; v < NV; ++v) s_base[v] += s_w[0][v] + s_w[1][v] + s_w[2][v] + s_w[3][v];
        __syncthreads();
    }
    // ---- the workgroup that finishes last: totals (out[n]), the count for the statistics block, the chunk plan
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        unsigned long long* done = (unsigned long long*)A.part + (size_t)256 * 8;
        const unsigned long long old = VMX_ADD_ACQREL(done, 1ULL);
        s_last = old + 1ULL == (unsigned long long)G;
        if (s_last) VMX_ST_REL(done, 0ULL);                         // (every workgroup of this launch has counted itself: ready for the next launch)
    }
    __syncthreads();
    if (s_last && tid == 0) {
        long long tot[NV];
        for (int v = 0; v < NV; ++v) { long long t = 0; for (int g = 0; g < G; ++g) t += VMX_LD_RLX((long long*)A.part + (size_t)g * 8 + v); tot[v] = t; A.off[v][n] = t; }
        if (A.stat_out) *A.stat_out = n;
        if constexpr (DP) vmx_round_plan(A, n, tot);
    }
}
template __global__ void k_round_prep<false>(vmx_round_args A);
template __global__ void k_round_prep<true>(vmx_round_args A);

// ---- the local stage without a host wait (round 6): the local-anchor counts stay on the device
// queue keys of the local chain DP: reads with local anchors, most first (k_size_hist / k_size_scatter); the others are not queued
__global__ void k_la_sizes(const int32_t* __restrict__ la_cnt, int n, int64_t* __restrict__ size, int32_t* __restrict__ n_dev) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r == 0) *n_dev = n;
    if (r < n) size[r] = la_cnt[r] > 0 ? (int64_t)la_cnt[r] : -1;
}
// The per-read pool geometry of the extend stage (segment anchors 3 cl + 8, segments cl + 2, record text 3 len + 64 (cl / 2 + 2) + 56 for a read with cl local anchors) as
// three exclusive scans over the reads, ONE workgroup. The pools themselves are sized from the context's history: a read whose slice would pass the end of a pool
// gets an EMPTY slice (the phase kernels then mark it VMX_EXT_CAPACITY_DEV and it is run again alone). tot[0..5] = segment anchors, segments, text bytes, problem
// slots (cl + 2 per read with anchors), local anchors, reads that were cut.
__global__ void __launch_bounds__(1024) k_ext_geometry(const int32_t* __restrict__ la_cnt, const int64_t* __restrict__ roff, int n, int mul, int tmask, long long tdiv,
                                                       long long capA, long long capS, long long capB, int64_t* __restrict__ coff3, int64_t* __restrict__ soff2,
                                                       int64_t* __restrict__ bloboff, int64_t* __restrict__ tot) {
    __shared__ long long s_w[16][3];
    __shared__ long long s_base[3];
    __shared__ long long s_acc[3];
    const int tid = (int)threadIdx.x, lane = vmx_lane(), wv = tid >> 6;
    auto pool = [&](int bit, long long x) -> long long { const long long y = x * mul; return (tmask & bit) ? (y / tdiv > 1 ? y / tdiv : 1) : y; };
    if (tid < 3) s_base[tid] = 0;
    if (tid == 0) { s_acc[0] = 0; s_acc[1] = 0; s_acc[2] = 0; }
    __syncthreads();
    long long full = 0, la = 0, cut = 0;
    for (int r0 = 0; r0 < n; r0 += 1024) {
        const int r = r0 + tid;
        long long x[3] = {0, 0, 0};
        if (r < n) {
            const long long cl = la_cnt[r], len = roff[r + 1] - roff[r];
            if (cl > 0) { x[0] = pool(1, 3 * cl + 8); x[1] = pool(2, cl + 2); x[2] = (pool(4, 3 * len + 64 * (cl / 2 + 2) + 56) + 7) & ~7LL; full += cl + 2; la += cl; }
        }
        long long inc[3];
        for (int v = 0; v < 3; ++v) {
            long long c = x[v];
            for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(c, o); if (lane >= o) c += y; }
            inc[v] = c;
            if (lane == 63) s_w[wv][v] = c;
        }
        __syncthreads();
        long long off[3];
        for (int v = 0; v < 3; ++v) { long long wb = 0; for (int w = 0; w < wv; ++w) wb += s_w[w][v]; off[v] = s_base[v] + wb + inc[v] - x[v]; }
        // (a cut read keeps its place in the scan — the offsets stay monotonic — but its slice is emptied by giving the NEXT read the same offset: done in the pass below)
        if (r < n) { coff3[r] = off[0]; soff2[r] = off[1]; bloboff[r] = off[2]; }
        __syncthreads();
        if (tid == 0) for (int v = 0; v < 3; ++v) { long long t = 0; for (int w = 0; w < 16; ++w) t += s_w[w][v]; s_base[v] += t; }
        __syncthreads();
    }
    if (tid == 0) { coff3[n] = s_base[0]; soff2[n] = s_base[1]; bloboff[n] = s_base[2]; }
    __syncthreads();
    // the pools end at capA / capS / capB: every offset is clamped there, so a read that starts beyond an end, or reaches over it, owns less than it needs (the
    // phase kernels check a read's slice before they use it)
    for (int r = tid; r <= n; r += 1024) {
        if (r < n && (coff3[r + 1] > capA || soff2[r + 1] > capS || bloboff[r + 1] > capB) && la_cnt[r] > 0) ++cut;
    }
    __syncthreads();
    for (int r = tid; r <= n; r += 1024) {
        if (coff3[r] > capA) coff3[r] = capA;
        if (soff2[r] > capS) soff2[r] = capS;
        if (bloboff[r] > capB) bloboff[r] = capB;
    }
    full = vmx_wave_sum_i64(full); la = vmx_wave_sum_i64(la); cut = vmx_wave_sum_i64(cut);
    if (lane == 0) { atomicAdd((unsigned long long*)&s_acc[0], (unsigned long long)full); atomicAdd((unsigned long long*)&s_acc[1], (unsigned long long)la); atomicAdd((unsigned long long*)&s_acc[2], (unsigned long long)cut); }
    __syncthreads();
    if (tid == 0) { tot[0] = s_base[0]; tot[1] = s_base[1]; tot[2] = s_base[2]; tot[3] = s_acc[0]; tot[4] = s_acc[1]; tot[5] = s_acc[2]; }
}
