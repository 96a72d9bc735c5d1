// k_ed.hip — E2 divergence filter: global edit distance (edlib.align(task='distance'), /root/reference/src/vacmap/mammap_clrnano.py:19251),
// plus k_size_order, the longest-first work ordering shared by the DP kernels.
//
// Myers/Hyyro bit-vector recurrence (spec VMX-ED; oracle/vmo_dp.cc). The pattern (query) is cut into 64-row blocks; a lane owns
// one block and a step is one anti-diagonal (lane b works on text column step-b), so a wavefront sweeps 64 blocks x n columns
// per PASS. A segment of a 15 kb read has ~235 blocks = 4 passes, a 100 kb read 25. Passes are PIPELINED over the waves of one
// workgroup: wave w runs passes w, w+W, ...; pass p consumes the horizontal deltas leaving block 63 of pass p-1 ("carry", one
// int8 per text column, ring of W arrays in HBM) 64 columns at a time, gated by a progress word in LDS. The critical path of a
// problem is therefore ~n + 128*passes steps instead of passes*n. Text and carry are fetched 64 columns at a time (one
// coalesced 64-byte load per wave) and handed to lane 0 with v_readlane; lane-to-lane hand-off is a DPP wave_shr:1 move.
// Nothing on the per-step path touches memory. Problems are taken longest-first from a device-side queue (k_size_order);
// long patterns (> 4 passes) run on 16-wave workgroups, the rest on 4-wave workgroups.
#include "vmx_device.h"
#include "vmx_kernels.h"

// order[] = problem indices sorted by size class descending (quarter octaves: floor(log2(size)) and the two bits below the leading one);
// range[0] = number of problems with size > thresh (the head of the order when thresh + 1 is a power of two),
// range[1] = number of problems queued (those with size >= 0); counters[0..3] = 0 (work-queue heads of the consumers)
__device__ __forceinline__ int vmx_size_class(long long s) {
    if (s < 0) return -1;
    if (s == 0) return 0;
    const int e = 63 - __clzll(s);
    const int m = (int)((e >= 2 ? s >> (e - 2) : s << (2 - e)) & 3);
    return (e << 2) | m;
}
__global__ void __launch_bounds__(1024) k_size_order(const int64_t* __restrict__ size, const int32_t* __restrict__ n_ptr, int64_t thresh,
                                                     int32_t* __restrict__ order, int32_t* __restrict__ range, int32_t* __restrict__ counters) {
    VMX_SETPRIO(3);
    __shared__ int s_hist[256];
    __shared__ int s_cur[256];
    __shared__ int s_long;
    const int n = *n_ptr;
    const int lane = vmx_lane();
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    // nearly all problems of a launch share two or three size classes: the lanes of a wave that hit the same class are counted with
    // one ballot and ONE LDS atomic (a plain per-lane atomicAdd serialises 10^5 updates on the same word)
    int nl = 0;
    for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
        const int i = i0 + (int)threadIdx.x;
        const long long s = i < n ? size[i] : -1;
        int b = vmx_size_class(s);                                  // negative size: problem already settled, not queued
        if (s > thresh) ++nl;
        unsigned long long todo = __ballot(b >= 0);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b0 = vmx_readlane(b, leader);
            const unsigned long long same = __ballot(b == b0);
            if (lane == leader) atomicAdd(&s_hist[b0], __popcll(same));
            todo &= ~same;
        }
    }
    if (nl) atomicAdd(&s_long, nl);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 255; b >= 0; --b) { s_cur[b] = acc; acc += s_hist[b]; }
        range[0] = s_long; range[1] = acc;
        counters[0] = 0; counters[1] = 0; counters[2] = 0; counters[3] = 0;
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += (int)blockDim.x) {
        const int i = i0 + (int)threadIdx.x;
        const long long s = i < n ? size[i] : -1;
        int b = vmx_size_class(s);
        unsigned long long todo = __ballot(b >= 0);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b0 = vmx_readlane(b, leader);
            const unsigned long long same = __ballot(b == b0);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_cur[b0], __popcll(same));
            base = vmx_readlane(base, leader);
            if (b == b0) order[base + __popcll(same & ((1ULL << lane) - 1ULL))] = i;
            todo &= ~same;
        }
    }
}

// The same ordering on the whole device (round 4): k_size_order is ONE workgroup walking all ~230 k problems of a batch twice (0.55 ms during which its
// stream's batch does nothing else). k_size_hist counts the classes with any number of workgroups (per-workgroup LDS histogram, then one global
// atomic per class), k_size_scatter turns the counts into class bases (largest class first) and places every problem with one global atomic
// per wave and class. ghist[257] (class counts, [256] = problems above `thresh`) and gcur[256] must be zero at launch. The order INSIDE a class
// is whatever the atomics give — it only ever decided which of two equally large problems a wave takes first.
__global__ void __launch_bounds__(256) k_size_hist(const int64_t* __restrict__ size, const int32_t* __restrict__ n_ptr, int64_t thresh, int32_t* __restrict__ ghist) {
    __shared__ int s_hist[257];
    const int n = *n_ptr;
    const int lane = vmx_lane();
    for (int t = (int)threadIdx.x; t < 257; t += (int)blockDim.x) s_hist[t] = 0;
    __syncthreads();
    int nl = 0;
    for (int i0 = (int)(blockIdx.x * blockDim.x); i0 < n; i0 += (int)(gridDim.x * blockDim.x)) {
        const int i = i0 + (int)threadIdx.x;
        const long long s = i < n ? size[i] : -1;
        const int b = vmx_size_class(s);
        if (s > thresh) ++nl;
        unsigned long long todo = __ballot(b >= 0);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b0 = vmx_readlane(b, leader);
            const unsigned long long same = __ballot(b == b0);
            if (lane == leader) atomicAdd(&s_hist[b0], __popcll(same));
            todo &= ~same;
        }
    }
    if (nl) atomicAdd(&s_hist[256], nl);
    __syncthreads();
    for (int t = (int)threadIdx.x; t < 257; t += (int)blockDim.x) if (s_hist[t]) atomicAdd(&ghist[t], s_hist[t]);
}
__global__ void __launch_bounds__(256) k_size_scatter(const int64_t* __restrict__ size, const int32_t* __restrict__ n_ptr, const int32_t* __restrict__ ghist,
                                                      int32_t* __restrict__ gcur, int32_t* __restrict__ order, int32_t* __restrict__ range, int32_t* __restrict__ counters) {
    VMX_SETPRIO(3);
    __shared__ int s_base[256];
    const int n = *n_ptr;
    const int lane = vmx_lane();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 255; b >= 0; --b) { s_base[b] = acc; acc += ghist[b]; }
        if (blockIdx.x == 0) { range[0] = ghist[256]; range[1] = acc; counters[0] = 0; counters[1] = 0; counters[2] = 0; counters[3] = 0; }
    }
    __syncthreads();
    for (int i0 = (int)(blockIdx.x * blockDim.x); i0 < n; i0 += (int)(gridDim.x * blockDim.x)) {
        const int i = i0 + (int)threadIdx.x;
        const long long s = i < n ? size[i] : -1;
        const int b = vmx_size_class(s);
        unsigned long long todo = __ballot(b >= 0);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int b0 = vmx_readlane(b, leader);
            const unsigned long long same = __ballot(b == b0);
            int base = 0;
            if (lane == leader) base = atomicAdd(&gcur[b0], __popcll(same));
            base = vmx_readlane(base, leader);
            if (b == b0) order[s_base[b0] + base + __popcll(same & ((1ULL << lane) - 1ULL))] = i;
            todo &= ~same;
        }
    }
}

// which = 0: problems order[0 .. range[0]) (long), which = 1: order[range[0] .. range[1]). blockDim.x = 64 * W, W <= VMX_ED_WAVES.
__global__ void __launch_bounds__(64 * VMX_ED_WAVES) k_edit_distance(const uint8_t* __restrict__ qcodes, const int64_t* __restrict__ q_off,
                                                                      const uint8_t* __restrict__ tcodes, const int64_t* __restrict__ t_off,
                                                                      int8_t* __restrict__ carry_pool, const int32_t* __restrict__ order,
                                                                      const int32_t* __restrict__ range, int32_t* __restrict__ counters, int which,
                                                                      int64_t* __restrict__ out, int64_t carry_stride, int32_t* __restrict__ oflow) {
    __shared__ volatile unsigned long long s_prog[VMX_ED_WAVES];   // (pass << 32) | columns whose carry is published
    __shared__ int s_next;
    const int lane = vmx_lane();
    const int w = (int)(threadIdx.x >> 6);
    const int W = (int)(blockDim.x >> 6);
    const int lo = which == 0 ? 0 : range[0];
    const int hi = which == 0 ? range[0] : range[1];
    while (true) {
        if (threadIdx.x == 0) s_next = lo + atomicAdd(&counters[which], 1);
        __syncthreads();
        const int qi = vmx_uniform_i32(s_next);       // tell the compiler the queue index (and everything derived from it) is wave-uniform
        __syncthreads();
        if (qi >= hi) break;
        const int p = vmx_uniform_i32(order[qi]);
        const uint8_t* pat = qcodes + q_off[p];
        const uint8_t* txt = tcodes + t_off[p];
        const int m = vmx_uniform_i32((int)(q_off[p + 1] - q_off[p]));
        const int n = vmx_uniform_i32((int)(t_off[p + 1] - t_off[p]));
        // ring of up to VMX_ED_WAVES arrays of n entries: per problem (carry_stride == 0: the pool holds VMX_ED_WAVES bytes per text base) or
        // per workgroup (carry_stride = longest text the pool was sized for; a longer text raises the batch's overflow flag)
        int8_t* carry = carry_stride > 0 ? carry_pool + (size_t)blockIdx.x * (size_t)VMX_ED_WAVES * (size_t)carry_stride
                                         : carry_pool + (size_t)VMX_ED_WAVES * (size_t)t_off[p];
        const bool too_long = carry_stride > 0 && n > carry_stride;
        if (too_long && threadIdx.x == 0) { atomicExch(oflow, 1); out[p] = 0x7fffffffLL; }
        // an empty side makes the distance trivial; such problems still walk through every barrier below (P = 0 passes)
        const bool trivial = m == 0 || n == 0 || too_long;
        if (lane == 0) s_prog[w] = 0ULL;
        __syncthreads();
        const int B = (m + 63) >> 6;
        const int P = trivial ? 0 : (B + 63) >> 6;
        long long score = m;
        for (int ps = w; ps < P; ps += W) {
            const int b = ps * 64 + lane;
            const bool active_b = b < B;
            int nact = B - ps * 64; if (nact > 64) nact = 64;
            unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
            if (active_b) {
                const int base = b << 6;
                int lim = m - base; if (lim > 64) lim = 64;
                for (int x = 0; x < lim; ++x) {
                    const uint8_t c = pat[base + x];
                    const unsigned long long bit = 1ULL << x;
                    if (c == 0) p0 |= bit; else if (c == 1) p1 |= bit; else if (c == 2) p2 |= bit; else if (c == 3) p3 |= bit; else p4 |= bit;
                }
            }
            unsigned long long Pv = ~0ULL, Mv = 0ULL;
            const bool is_last_block = b == B - 1;
            const int hbit = is_last_block ? ((m - 1) & 63) : 63;     // bit whose horizontal delta leaves the block
            const bool last_pass = ps == P - 1;
            const int8_t* cin_arr = carry + (size_t)((ps + W - 1) % W) * (size_t)n;
            int8_t* cout_arr = carry + (size_t)(ps % W) * (size_t)n;
            const unsigned long long need_hi = (unsigned long long)(ps - 1) << 32;
            int hout_cur = 0, c_cur = 4;
            int outc = 0;            // delta of column j (leaving block 63) parked in lane j & 63 until 64 of them are published
            const int steps = n + nact - 1;
            for (int t0 = 0; t0 < steps; t0 += 64) {
                const int tch = (t0 + lane < n) ? (int)txt[t0 + lane] : 4;
                int cin = 0;
                if (ps > 0 && t0 < n) {
                    int need = t0 + 64; if (need > n) need = n;
                    while (s_prog[(ps - 1) % W] < (need_hi | (unsigned long long)need)) VMX_SPIN_PAUSE();
                    __threadfence_block();
                    cin = (t0 + lane < n) ? (int)cin_arr[t0 + lane] : 0;
                }
                int tend = steps - t0; if (tend > 64) tend = 64;
                for (int tt = 0; tt < tend; ++tt) {
                    const int t = t0 + tt;
                    const int c_up = vmx_shr1(c_cur);
                    const int h_up = vmx_shr1(hout_cur);
                    const int c0 = vmx_readlane(tch, tt);
                    const int h0 = vmx_readlane(cin, tt);
                    int hin;
                    if (lane == 0) { c_cur = t < n ? c0 : 4; hin = ps == 0 ? 1 : (t < n ? h0 : 0); }
                    else { c_cur = c_up; hin = h_up; }
                    const int j = t - lane;
                    if (active_b && j >= 0 && j < n) {
                        // branch-free body: 2-level select of the match vector, sign tricks for the carries
                        const unsigned long long s01 = (c_cur & 1) ? p1 : p0, s23 = (c_cur & 1) ? p3 : p2;
                        const unsigned long long s03 = (c_cur & 2) ? s23 : s01;
                        unsigned long long Eq = (c_cur & 4) ? p4 : s03;
                        const unsigned long long neg = (unsigned long long)((unsigned)hin >> 31);      // hin < 0
                        const unsigned long long pos = (unsigned long long)((unsigned)(-hin) >> 31);   // hin > 0
                        const unsigned long long Xv = Eq | Mv;
                        Eq |= neg;
                        const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                        unsigned long long Ph = Mv | ~(Xh | Pv);
                        unsigned long long Mh = Pv & Xh;
                        const int hout = (int)((Ph >> hbit) & 1ULL) - (int)((Mh >> hbit) & 1ULL);
                        Ph = (Ph << 1) | pos; Mh = (Mh << 1) | neg;
                        Pv = Mh | ~(Xv | Ph);
                        Mv = Ph & Xv;
                        hout_cur = hout;
                        score += is_last_block ? hout : 0;
                    }
                    if (!last_pass) {
                        // park the delta leaving block 63 (column t-63) in lane (column & 63); publish 64 columns at a time
                        const int h63 = vmx_readlane(hout_cur, 63);
                        const int j63 = t - 63;
                        if (j63 >= 0 && j63 < n) {
                            if (lane == (j63 & 63)) outc = h63;
                            if ((j63 & 63) == 63 || j63 == n - 1) {
                                const int base = j63 & ~63;
                                if (base + lane <= j63) cout_arr[base + lane] = (int8_t)outc;
                                __threadfence_block();
                                if (lane == 0) s_prog[w] = ((unsigned long long)ps << 32) | (unsigned long long)(j63 + 1);
                            }
                        }
                    }
                }
            }
        }
        // the score lives in the lane that owns block B-1 (wave (P-1) % W)
        if (trivial) { if (threadIdx.x == 0 && !too_long) out[p] = m == 0 ? n : m; }
        else if (w == (P - 1) % W) {
            const long long s = __shfl(score, (B - 1) & 63);
            if (lane == 0) out[p] = s;
        }
        __syncthreads();
    }
}
