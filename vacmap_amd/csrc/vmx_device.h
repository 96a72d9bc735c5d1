// vmx_device.h — device-side helpers (wave64 idioms for gfx950). Under -DVMX_EMU the same kernels are compiled
// against tests/emu/hip_emu.h (TEST-ONLY fiber emulator, never part of the product library).
#ifndef VMX_DEVICE_H
#define VMX_DEVICE_H
#ifdef VMX_EMU
#include "hip_emu.h"
#define VMX_DYN_SHARED(type, name) type* name = (type*)hipemu::cur()->dynshared
#else
#include <hip/hip_runtime.h>
#define VMX_DYN_SHARED(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif
#include <stdint.h>

#define VMX_WAVE 64
#define VMX_NEG (-(1 << 28))
#define VMX_NOPRE (-9999999)

__device__ __forceinline__ int vmx_lane() { return (int)(threadIdx.x & 63); }

// lane i receives the value of lane i-1 (lane 0 keeps its own): one v_mov_b32_dpp wave_shr:1 on gfx950 instead of the
// ds_bpermute that __shfl_up lowers to. vmx_readlane: value of a wave-uniform lane (v_readlane_b32).
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_shr1(int v) { return __shfl_up(v, 1); }
__device__ __forceinline__ int vmx_readlane(int v, int l) { return __shfl(v, l); }
#define VMX_SPIN_PAUSE() hipemu::yield()
#define VMX_CLOCK() 0LL
#else
__device__ __forceinline__ int vmx_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int vmx_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
#define VMX_SPIN_PAUSE() __builtin_amdgcn_s_sleep(1)
#define VMX_CLOCK() ((long long)wall_clock64())
#endif
// systolic hand-off used by the DP kernels: vmx_shr1_in(v, in): lane i <- lane i-1 of v, lane 0 <- its own `in` (wave_shr:1 leaves
// lane 0's destination untouched, so `in` rides in as the old value: one v_mov_dpp, no select);
// vmx_rol1: lane i <- lane i+1, lane 63 <- lane 0; vmx_ror1: lane i <- lane i-1, lane 0 <- lane 63
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_shr1_in(int v, int in) { int e = __shfl_up(v, 1); return vmx_lane() == 0 ? in : e; }
__device__ __forceinline__ int vmx_rol1(int v) { return __shfl(v, (vmx_lane() + 1) & 63); }
__device__ __forceinline__ int vmx_ror1(int v) { return __shfl(v, (vmx_lane() + 63) & 63); }
#else
__device__ __forceinline__ int vmx_shr1_in(int v, int in) { return __builtin_amdgcn_update_dpp(in, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int vmx_rol1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x134, 0xf, 0xf, false); }
__device__ __forceinline__ int vmx_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x13C, 0xf, 0xf, false); }
#endif
// packed int16 pairs in one 32-bit register (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16 / v_pk_ashrrev_i16 on gfx950): the gap-fill
// DP keeps two rows per lane. vmx_pk_neg(a) = 0xffff in every half whose int16 is negative.
#ifdef VMX_EMU
__device__ __forceinline__ unsigned vmx_pk(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int vmx_pk_lo(unsigned a) { return (int)(short)(a & 0xffffu); }
__device__ __forceinline__ int vmx_pk_hi(unsigned a) { return (int)(short)(a >> 16); }
__device__ __forceinline__ unsigned vmx_pk_add(unsigned a, unsigned b) { return vmx_pk(vmx_pk_lo(a) + vmx_pk_lo(b), vmx_pk_hi(a) + vmx_pk_hi(b)); }
__device__ __forceinline__ unsigned vmx_pk_sub(unsigned a, unsigned b) { return vmx_pk(vmx_pk_lo(a) - vmx_pk_lo(b), vmx_pk_hi(a) - vmx_pk_hi(b)); }
__device__ __forceinline__ unsigned vmx_pk_max(unsigned a, unsigned b) {
    const int l = vmx_pk_lo(a) > vmx_pk_lo(b) ? vmx_pk_lo(a) : vmx_pk_lo(b), h = vmx_pk_hi(a) > vmx_pk_hi(b) ? vmx_pk_hi(a) : vmx_pk_hi(b);
    return vmx_pk(l, h);
}
__device__ __forceinline__ unsigned vmx_pk_neg(unsigned a) { return (vmx_pk_lo(a) < 0 ? 0xffffu : 0u) | (vmx_pk_hi(a) < 0 ? 0xffff0000u : 0u); }
__device__ __forceinline__ unsigned vmx_pk_min_u16(unsigned a, unsigned b) { const unsigned al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16); }
__device__ __forceinline__ unsigned vmx_pk_mad(unsigned a, unsigned b, unsigned c) { return vmx_pk(vmx_pk_lo(a) * vmx_pk_lo(b) + vmx_pk_lo(c), vmx_pk_hi(a) * vmx_pk_hi(b) + vmx_pk_hi(c)); }
__device__ __forceinline__ unsigned vmx_alignbit16(unsigned hi, unsigned lo) { return (hi << 16) | (lo >> 16); }
#else
typedef short vmx_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vmx_pk(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int vmx_pk_lo(unsigned a) { return (int)(short)(a & 0xffffu); }
__device__ __forceinline__ int vmx_pk_hi(unsigned a) { return (int)(short)(a >> 16); }
__device__ __forceinline__ unsigned vmx_pk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (vmx_v2s)(__builtin_bit_cast(vmx_v2s, a) + __builtin_bit_cast(vmx_v2s, b))); }
__device__ __forceinline__ unsigned vmx_pk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (vmx_v2s)(__builtin_bit_cast(vmx_v2s, a) - __builtin_bit_cast(vmx_v2s, b))); }
__device__ __forceinline__ unsigned vmx_pk_max(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(vmx_v2s, a), __builtin_bit_cast(vmx_v2s, b))); }
// opaque on purpose: given the plain shift the compiler turns every use of the mask back into per-half compares and selects
__device__ __forceinline__ unsigned vmx_pk_neg(unsigned a) { unsigned r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ unsigned vmx_pk_min_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned vmx_pk_mad(unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ unsigned vmx_alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#endif
// low bytes of the two halves of a packed register, side by side in the low 16 bits (one v_perm_b32)
#ifdef VMX_EMU
__device__ __forceinline__ unsigned vmx_pk_bytes(unsigned b) { return (b & 0xffu) | ((b >> 8) & 0xff00u); }
#else
__device__ __forceinline__ unsigned vmx_pk_bytes(unsigned b) { return __builtin_amdgcn_perm(b, b, 0x0c0c0200u); }
#endif
// issue priority of the calling wave (s_setprio, 0 .. 3). The serial, latency-bound kernels (one lane or one wave per read walking
// dependent loads) raise it: they issue a handful of instructions per microsecond, and when another batch's VALU-bound kernel shares the
// SIMD every one of those waits behind a queue of packed-int16 arithmetic; getting them through first costs the arithmetic nothing.
#if defined(VMX_EMU) || defined(VMX_NOPRIO)
#define VMX_SETPRIO(p) ((void)0)
#else
#define VMX_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
// single-wavefront workgroups: LDS operations of one wave execute in order, so a value one lane stores is visible to the other lanes' later
// loads without a barrier; what is needed is only that the compiler keeps the order. (A __syncthreads() here also drains the wave's
// outstanding global stores — an HBM write acknowledgement per anchor in the chain kernels.) The emulator's lanes are fibers: keep the rendezvous.
#ifdef VMX_EMU
__device__ __forceinline__ void vmx_wave_lds_fence() { __syncthreads(); }
#else
__device__ __forceinline__ void vmx_wave_lds_fence() { asm volatile("" ::: "memory"); }
#endif
__device__ __forceinline__ unsigned vmx_bfi(unsigned m, unsigned a, unsigned b) { return (m & a) | (~m & b); }     // v_bfi_b32
// value known to be identical in every lane: hand it to the compiler as a scalar (v_readfirstlane_b32)
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_uniform_i32(int v) { return v; }
#else
__device__ __forceinline__ int vmx_uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
__device__ __forceinline__ long long vmx_uniform_i64(long long v) {
    union { long long d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_uniform_i32(u.i[0]); u.i[1] = vmx_uniform_i32(u.i[1]);
    return u.d;
}
__device__ __forceinline__ double vmx_uniform_f64(double v) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_uniform_i32(u.i[0]); u.i[1] = vmx_uniform_i32(u.i[1]);
    return u.d;
}
// a pointer read from a structure in memory is a generic (flat) pointer to the compiler: every access through it is a flat_load that counts
// on both wait counters and cannot take a scalar base. VMX_GLOBAL_PTR says "this is device memory" (address space 1) and "the same in every
// lane" (scalar register pair): accesses become global_load v, v_off, s[base].
#ifdef VMX_EMU
#define VMX_GLOBAL_PTR(T, p) (p)
#else
#define VMX_GLOBAL_PTR(T, p) ((T*)(__attribute__((address_space(1))) T*)(unsigned long long)vmx_uniform_i64((long long)(p)))
#endif
// broadcast lane 0's value to the wave
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_bcast0(int v) { return __shfl(v, 0); }
#else
__device__ __forceinline__ int vmx_bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
__device__ __forceinline__ double vmx_shr1_f64(double v) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_shr1(u.i[0]); u.i[1] = vmx_shr1(u.i[1]);
    return u.d;
}
__device__ __forceinline__ long long vmx_shr1_i64(long long v) {
    union { long long d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_shr1(u.i[0]); u.i[1] = vmx_shr1(u.i[1]);
    return u.d;
}
// The chain kernels' candidate window: lane t holds entry S_arg[m-1-t] of the score-sorted index (m entries so far), i.e. the 64 best
// predecessors in scan order, with everything a candidate evaluation needs (anchor index, score, read position, length|strand<<16,
// reference position). A new anchor whose score lands inside the window is inserted with one ballot and seven v_mov_dpp; no LDS read
// sits on the per-anchor dependency chain.
struct vmx_cwin { int j, q, ls; double S; long long r; };
__device__ __forceinline__ void vmx_cwin_insert(vmx_cwin& w, int at, int k, double Sk, int qk, int lsk, long long rk, int lane) {
    const int sj = vmx_shr1(w.j), sq = vmx_shr1(w.q), sls = vmx_shr1(w.ls);
    const double sS = vmx_shr1_f64(w.S); const long long sr = vmx_shr1_i64(w.r);
    if (lane > at) { w.j = sj; w.q = sq; w.ls = sls; w.S = sS; w.r = sr; }
    else if (lane == at) { w.j = k; w.q = qk; w.ls = lsk; w.S = Sk; w.r = rk; }
}
__device__ __forceinline__ double vmx_readlane_f64(double v, int l) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_readlane(u.i[0], l); u.i[1] = vmx_readlane(u.i[1], l);
    return u.d;
}

// wave-wide reductions / scans; every lane of the wave must call them (wave-uniform control flow)
__device__ __forceinline__ int vmx_wave_max_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) { int x = __shfl_xor(v, o); v = x > v ? x : v; }
    return v;
}
__device__ __forceinline__ int vmx_wave_sum_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ long long vmx_wave_sum_i64(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ int vmx_wave_incl_scan_i32(int v) {
    int lane = vmx_lane();
    for (int o = 1; o < 64; o <<= 1) { int x = __shfl_up(v, o); if (lane >= o) v += x; }
    return v;
}
// exclusive prefix MAX of doubles across the wave; lane 0 gets `init`
__device__ __forceinline__ double vmx_wave_excl_max_f64(double v, double init) {
    int lane = vmx_lane();
    // inclusive scan of max
    for (int o = 1; o < 64; o <<= 1) { double x = __shfl_up(v, o); if (lane >= o) v = x > v ? x : v; }
    double e = __shfl_up(v, 1);
    if (lane == 0) e = init; else e = e > init ? e : init;
    return e;
}

// DPP forms of the prefix max used by the chain kernels' candidate scan: Kogge-Stone inside the 16-lane rows (row_shr:1/2/4/8),
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 — 12 v_mov_dpp + 6 max instead of 14 ds_bpermute round trips.
#define VMX_F64_NEG (-__builtin_inf())
#ifdef VMX_EMU
__device__ __forceinline__ double vmx_wave_incl_max_f64(double v) {
    int lane = vmx_lane();
    for (int o = 1; o < 64; o <<= 1) { double x = __shfl_up(v, o); if (lane >= o) v = x > v ? x : v; }
    return v;
}
__device__ __forceinline__ double vmx_wave_shr1_f64_fill(double v, double fill) {     // lane i <- lane i-1, lane 0 <- fill
    double e = __shfl_up(v, 1);
    return vmx_lane() == 0 ? fill : e;
}
#else
template <int CTRL, int ROWMASK> __device__ __forceinline__ double vmx_dpp_f64(double old, double v) {
    union { double d; int i[2]; } o, s, r; o.d = old; s.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], CTRL, ROWMASK, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], CTRL, ROWMASK, 0xf, false);
    return r.d;
}
__device__ __forceinline__ double vmx_wave_incl_max_f64(double v) {
    double t;
    t = vmx_dpp_f64<0x111, 0xf>(VMX_F64_NEG, v); v = t > v ? t : v;
    t = vmx_dpp_f64<0x112, 0xf>(VMX_F64_NEG, v); v = t > v ? t : v;
    t = vmx_dpp_f64<0x114, 0xf>(VMX_F64_NEG, v); v = t > v ? t : v;
    t = vmx_dpp_f64<0x118, 0xf>(VMX_F64_NEG, v); v = t > v ? t : v;
    t = vmx_dpp_f64<0x142, 0xa>(VMX_F64_NEG, v); v = t > v ? t : v;
    t = vmx_dpp_f64<0x143, 0xc>(VMX_F64_NEG, v); v = t > v ? t : v;
    return v;
}
__device__ __forceinline__ double vmx_wave_shr1_f64_fill(double v, double fill) { return vmx_dpp_f64<0x138, 0xf>(fill, v); }
#endif

// wave-cooperative count of the x in [0, k) with S[SA[x]] < target (le = false) or <= target (le = true), S ascending along SA:
// 64-ary search, two rounds of (two dependent loads + ballot) up to k = 4096. Every lane of the wave must call it.
__device__ __forceinline__ int vmx_sorted_count(const double* S, const int* SA, int k, double target, bool le, int lane) {
    int lo = 0, hi = k;                   // elements [0, lo) qualify, elements [hi, k) do not
    while (hi - lo > 64) {
        const int stride = (hi - lo + 63) >> 6;
        const int x = lo + (lane + 1) * stride - 1;
        bool in = false;
        if (x < hi) { const double v = S[SA[x]]; in = le ? v <= target : v < target; }
        const int c = __popcll(__ballot(in));
        const int nlo = lo + c * stride;
        int nhi = nlo + stride - 1; if (nhi > hi) nhi = hi;
        lo = nlo; hi = nhi;
    }
    const int x = lo + lane;
    bool in = false;
    if (x < hi) { const double v = S[SA[x]]; in = le ? v <= target : v < target; }
    return lo + __popcll(__ballot(in));
}

// wave-cooperative SA[loc+1 : k+1] = SA[loc : k]; SA[loc] = k (256 elements per round)
__device__ __forceinline__ void vmx_sarg_insert4(int* SA, int loc, int k, int lane) {
    for (int hi = k; hi > loc; hi -= 256) {
        const int x0 = hi - lane;             // consecutive lanes touch consecutive words: no LDS bank conflicts
        int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (x0 > loc) v0 = SA[x0 - 1];
        if (x0 - 64 > loc) v1 = SA[x0 - 65];
        if (x0 - 128 > loc) v2 = SA[x0 - 129];
        if (x0 - 192 > loc) v3 = SA[x0 - 193];
        __syncthreads();
        if (x0 > loc) SA[x0] = v0;
        if (x0 - 64 > loc) SA[x0 - 64] = v1;
        if (x0 - 128 > loc) SA[x0 - 128] = v2;
        if (x0 - 192 > loc) SA[x0 - 192] = v3;
        __syncthreads();
    }
    if (lane == 0) SA[loc] = k;
    __syncthreads();
}

// block-wide exclusive scan of one int per thread (blockDim.x <= 1024, multiple of 64); returns exclusive prefix,
// *total gets the block sum. scratch: >= 17 ints of shared memory.
__device__ __forceinline__ int vmx_block_excl_scan(int v, int* scratch, int* total) {
    int lane = vmx_lane(), w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
    int inc = vmx_wave_incl_scan_i32(v);
    __syncthreads();
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    if (w == 0) {                                                  // the (<= 16) wave totals: one more wave-level scan, not a loop in one thread
        const int t = lane < nw ? scratch[lane] : 0;
        const int i2 = vmx_wave_incl_scan_i32(t);
        if (lane < nw) scratch[lane] = i2 - t;
        if (lane == nw - 1) scratch[16] = i2;
    }
    __syncthreads();
    int base = scratch[w];
    *total = scratch[16];
    return base + inc - v;
}
// histogram cut: with bins summed from 1023 downwards (bins below `lo` left out), the bin where the running count reaches `want`:
// res[0] = that bin (0: never reached), res[1] = the count in the bins above it. hist: 1024 bins in LDS; res: two ints in LDS; every
// thread of the workgroup calls it. (One thread walking the bins cost a 1024-thread workgroup 65 us per read: more than the rest of
// k_cluster_big together.)
__device__ __forceinline__ void vmx_hist_cut(const uint32_t* hist, int lo, int want, int* scratch, int* res) {
    if (threadIdx.x == 0) { res[0] = 0; res[1] = 0; }
    __syncthreads();
    int acc = 0;
    for (int b0 = 1023; b0 >= lo; b0 -= (int)blockDim.x) {
        const int b = b0 - (int)threadIdx.x;
        const int v = b >= lo ? (int)hist[b] : 0;
        int tot; const int ex = acc + vmx_block_excl_scan(v, scratch, &tot);
        if (ex < want && ex + v >= want) { res[0] = b; res[1] = ex; }
        acc += tot;
        __syncthreads();
        if (acc >= want) break;
    }
}

// block-wide bitonic sort of N (power of two) uint64 keys living in HBM at g; staged through `lds` (lds_cap keys) when they fit.
// every thread of the workgroup must call it.
__device__ __forceinline__ void vmx_block_bitonic_passes(uint64_t* a, int N) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    uint64_t x = a[i], y = a[ixj];
                    bool asc = (i & k) == 0;
                    if ((x > y) == asc) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}
// ---- register-blocked bitonic sort of an LDS tile -------------------------------------------------------------------------------
// The plain passes above make one LDS round trip and one barrier per (k, j) stage — 105 of them for 16384 keys — with half of the
// threads idle. Here a thread takes SIXTEEN keys whose indices differ in four consecutive bits [b, b + 4) into registers and runs up to
// four stages (j = 2^(b+3) .. 2^b) on them before they go back: 29 round trips for 16384 keys, every thread busy, the compare-exchange
// network fully unrolled. The first round trip sorts every run of 16 consecutive keys outright (phases k = 2 .. 16, ten stages).
// Keys live in LDS in a swizzled order, slot(i) = i ^ ((i >> 4) & 15): whatever four bits a round trip spreads over the registers, the
// sixteen lanes that issue together touch sixteen different 8-byte bank pairs (a linear layout would put the 128-byte strides of the
// low-bit round trips on one bank). vmx_sw() maps an index to its slot for callers that read the sorted tile in place.
__device__ __forceinline__ int vmx_sw(int i) { return i ^ ((i >> 4) & 15); }
__device__ __forceinline__ void vmx_ce_u64(uint64_t& x, uint64_t& y, bool asc) {
    const bool sw = (x > y) == asc;
    const uint64_t lo = sw ? y : x, hi = sw ? x : y;
    x = lo; y = hi;
}
// one round trip: stages j = 2^(b + nst - 1) .. 2^b of phase k (k > 2^(b + LR - 1): one direction per thread), or with first = true the
// whole of phases 2 .. R on runs of R consecutive keys (b = 0). gbase: global index of lds[0] (tiled sorts).
// R = 2^VMX_SORT_LOGR keys per thread: 16 by default; a kernel held to few registers defines VMX_SORT_LOGR 3 before including this
// header (k_local_seed: at its 80-VGPR budget the 16-key form lived in scratch memory).
#ifndef VMX_SORT_LOGR
#define VMX_SORT_LOGR 4
#endif
#define VMX_SORT_R (1 << VMX_SORT_LOGR)
template <int LOGR>
__device__ __forceinline__ void vmx_bitonic_roundtrip_t(uint64_t* lds, int N, int gbase, int k, int b, int nst, bool first) {
    constexpr int R = 1 << LOGR;
    const int T = (int)blockDim.x;
    for (int g = (int)threadIdx.x; g < (N >> LOGR); g += T) {
        const int base = ((g >> b) << (b + LOGR)) | (g & ((1 << b) - 1));
        uint64_t r[R];
#pragma unroll
        for (int m = 0; m < R; ++m) r[m] = lds[vmx_sw(base | (m << b))];
        if (first) {
            const bool ascR = ((gbase + base) & R) == 0;
#pragma unroll
            for (int kk = 2; kk <= R; kk <<= 1)
#pragma unroll
                for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
                    for (int m = 0; m < R; ++m)
                        if ((m ^ jj) > m) vmx_ce_u64(r[m], r[m ^ jj], kk == R ? ascR : ((m & kk) == 0));
        } else {
            const bool asc = ((gbase + base) & k) == 0;
#pragma unroll
            for (int s = LOGR - 1; s >= 0; --s) {
                if (nst > s) {
#pragma unroll
                    for (int m = 0; m < R; ++m) if (!(m & (1 << s))) vmx_ce_u64(r[m], r[m | (1 << s)], asc);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < R; ++m) lds[vmx_sw(base | (m << b))] = r[m];
    }
    __syncthreads();
}
// stages j = 2^e .. 1 of phase k on the swizzled tile (e < log2 N; k > 2^e)
template <int LOGR>
__device__ __forceinline__ void vmx_bitonic_phase_tail_t(uint64_t* lds, int N, int gbase, int k, int e) {
    while (e >= 0) {
        const int b = e >= LOGR - 1 ? e - (LOGR - 1) : 0;
        vmx_bitonic_roundtrip_t<LOGR>(lds, N, gbase, k, b, e - b + 1, false);
        e = b - 1;
    }
}
// all phases k = 2 .. kmax of a swizzled tile of N >= R keys (N a power of two). every thread of the workgroup must call it.
template <int LOGR>
__device__ __forceinline__ void vmx_bitonic_tile_sw_t(uint64_t* lds, int N, int gbase, int kmax) {
    constexpr int R = 1 << LOGR;
    vmx_bitonic_roundtrip_t<LOGR>(lds, N, gbase, R, 0, LOGR, true);
    for (int k = 2 * R, p = LOGR + 1; k <= kmax; k <<= 1, ++p) vmx_bitonic_phase_tail_t<LOGR>(lds, N, gbase, k, p - 1);
}
__device__ __forceinline__ void vmx_bitonic_tile_sw(uint64_t* lds, int N, int gbase, int kmax) { vmx_bitonic_tile_sw_t<VMX_SORT_LOGR>(lds, N, gbase, kmax); }
__device__ __forceinline__ void vmx_bitonic_phase_tail(uint64_t* lds, int N, int gbase, int k, int e) { vmx_bitonic_phase_tail_t<VMX_SORT_LOGR>(lds, N, gbase, k, e); }
// is the register-blocked form worth it? it needs R keys per working thread; below a quarter of the workgroup the plain passes win
__device__ __forceinline__ bool vmx_bitonic_fast_ok(int N) { return N >= 4 * VMX_SORT_R && (N >> VMX_SORT_LOGR) >= ((int)blockDim.x >> 2); }

// ---- rank-merge sort of an LDS tile --------------------------------------------------------------------------------------------
// For a few thousand keys under a 1024-thread workgroup the bitonic forms are bound by their barriers (66 stages for 2048 keys) or, register-
// blocked, by the two wavefronts that hold all the keys. Here every thread sorts four consecutive keys in registers, then runs double in length
// round by round: an element's slot in the merged run is its offset in its own run plus its rank in the sibling run (a branch-free binary
// search; left runs count the sibling's smaller keys, right runs the smaller-or-equal ones, so equal keys — the padding — get distinct slots).
// log2(N) - 2 rounds of one barrier each, every thread busy, keys in linear order. A: the keys (N a power of two >= 4), B: N more slots;
// the result is in A. Every thread of the workgroup must call it.
__device__ __forceinline__ void vmx_rank_merge_sort_lds(uint64_t* A, uint64_t* B, int N) {
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    int rounds = 0; for (int L = 4; L < N; L <<= 1) ++rounds;
    uint64_t* src = (rounds & 1) ? B : A;                         // an odd number of rounds starts from B, so that the last one lands in A
    uint64_t* dst = (rounds & 1) ? A : B;
    for (int g = tid; g < (N >> 2); g += T) {
        uint64_t r0 = A[4 * g], r1 = A[4 * g + 1], r2 = A[4 * g + 2], r3 = A[4 * g + 3];
        vmx_ce_u64(r0, r1, true); vmx_ce_u64(r2, r3, true); vmx_ce_u64(r0, r2, true); vmx_ce_u64(r1, r3, true); vmx_ce_u64(r1, r2, true);
        src[4 * g] = r0; src[4 * g + 1] = r1; src[4 * g + 2] = r2; src[4 * g + 3] = r3;
    }
    __syncthreads();
    for (int L = 4; L < N; L <<= 1) {
        // two elements per thread and step: the two searches are independent, their LDS reads overlap
        for (int i0 = tid; i0 < N; i0 += 2 * T) {
            const int i1 = i0 + T < N ? i0 + T : i0;
            const uint64_t x0 = src[i0], x1 = src[i1];
            const bool l0 = (i0 & L) == 0, l1 = (i1 & L) == 0;
            const int b0 = i0 & ~(2 * L - 1), b1 = i1 & ~(2 * L - 1);
            const uint64_t* s0 = src + b0 + (l0 ? L : 0);
            const uint64_t* s1 = src + b1 + (l1 ? L : 0);
            // right runs count the keys <= x, i.e. < x + 1; the all-ones padding in a right run goes behind everything on the left
            const bool top0 = !l0 && x0 == ~0ULL, top1 = !l1 && x1 == ~0ULL;
            const uint64_t a0 = l0 || top0 ? x0 : x0 + 1, a1 = l1 || top1 ? x1 : x1 + 1;
            int p0 = 0, p1 = 0;
            for (int st = L >> 1; st > 0; st >>= 1) {
                const uint64_t v0 = s0[p0 + st - 1], v1 = s1[p1 + st - 1];
                p0 += v0 < a0 ? st : 0; p1 += v1 < a1 ? st : 0;
            }
            { const uint64_t v0 = s0[p0], v1 = s1[p1]; p0 += v0 < a0 ? 1 : 0; p1 += v1 < a1 ? 1 : 0; }
            if (top0) p0 = L;
            if (top1) p1 = L;
            dst[b0 + (i0 & (L - 1)) + p0] = x0;
            if (i1 != i0) dst[b1 + (i1 & (L - 1)) + p1] = x1;
        }
        __syncthreads();
        uint64_t* t = src; src = dst; dst = t;
    }
}

// (the passes are instantiated once on the LDS buffer and once on the HBM array: a pointer chosen at run time would make them flat accesses)
__device__ inline void vmx_block_sort_u64_impl(uint64_t* g, int N, uint64_t* lds, int lds_cap) {
    if (N <= lds_cap && vmx_bitonic_fast_ok(N)) {
        for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) lds[vmx_sw(i)] = g[i];
        __syncthreads();
        vmx_bitonic_tile_sw(lds, N, 0, N);
        for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) g[i] = lds[vmx_sw(i)];
        __syncthreads();
    } else if (N <= lds_cap) {
        for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) lds[i] = g[i];
        __syncthreads();
        vmx_block_bitonic_passes(lds, N);
        for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) g[i] = lds[i];
        __syncthreads();
    } else {
        __syncthreads();
        vmx_block_bitonic_passes(g, N);
    }
}

// The same sort for N beyond the LDS tile (`tile` keys, a power of two): every stage whose partner distance fits a tile runs in LDS (load
// a tile, run the stages, store it), only the few steps with partner distance >= tile touch HBM (coalesced: thread i and thread i + 1
// touch neighbouring keys). N = 4 tiles: 3 HBM steps and 3 LDS residencies per tile instead of 120 HBM passes.
// every thread of the workgroup must call it.
// Returns 0 when the LDS buffer holds nothing the caller may use, 1 when it holds the N sorted keys in plain order, 2 when it holds them
// in swizzled order (key i at lds[vmx_sw(i)]).
__device__ inline int vmx_block_sort_u64_tiled(uint64_t* g, int N, uint64_t* lds, int tile) {
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    if (N <= tile) {
        if (vmx_bitonic_fast_ok(N)) {
            for (int i = tid; i < N; i += T) lds[vmx_sw(i)] = g[i];
            __syncthreads();
            vmx_bitonic_tile_sw(lds, N, 0, N);
            for (int i = tid; i < N; i += T) g[i] = lds[vmx_sw(i)];
            __syncthreads();
            return 2;
        }
        for (int i = tid; i < N; i += T) lds[i] = g[i];
        __syncthreads();
        vmx_block_bitonic_passes(lds, N);
        for (int i = tid; i < N; i += T) g[i] = lds[i];
        __syncthreads();
        return 1;
    }
    const bool fast = vmx_bitonic_fast_ok(tile);
    int ltile = 0; while ((1 << ltile) < tile) ++ltile;
    for (int base = 0; base < N; base += tile) {                // all stages k <= tile, tile by tile (direction from the global index)
        if (fast) {
            for (int i = tid; i < tile; i += T) lds[vmx_sw(i)] = g[base + i];
            __syncthreads();
            vmx_bitonic_tile_sw(lds, tile, base, tile);
            for (int i = tid; i < tile; i += T) g[base + i] = lds[vmx_sw(i)];
            __syncthreads();
            continue;
        }
        for (int i = tid; i < tile; i += T) lds[i] = g[base + i];
        __syncthreads();
        for (int k = 2; k <= tile; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < tile; i += T) {
                    const int ixj = i ^ j;
                    if (ixj > i) { const uint64_t x = lds[i], y = lds[ixj]; const bool asc = ((base + i) & k) == 0; if ((x > y) == asc) { lds[i] = y; lds[ixj] = x; } }
                }
                __syncthreads();
            }
        for (int i = tid; i < tile; i += T) g[base + i] = lds[i];
        __syncthreads();
    }
    for (int k = 2 * tile; k <= N; k <<= 1) {
        for (int j = k >> 1; j >= tile; j >>= 1) {              // partner in another tile: through HBM
            for (int i = tid; i < N; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) { const uint64_t x = g[i], y = g[ixj]; const bool asc = (i & k) == 0; if ((x > y) == asc) { g[i] = y; g[ixj] = x; } }
            }
            __syncthreads();
        }
        for (int base = 0; base < N; base += tile) {            // the remaining steps of this stage stay inside a tile
            if (fast) {
                for (int i = tid; i < tile; i += T) lds[vmx_sw(i)] = g[base + i];
                __syncthreads();
                vmx_bitonic_phase_tail(lds, tile, base, k, ltile - 1);
                for (int i = tid; i < tile; i += T) g[base + i] = lds[vmx_sw(i)];
                __syncthreads();
                continue;
            }
            for (int i = tid; i < tile; i += T) lds[i] = g[base + i];
            __syncthreads();
            const bool asc = (base & k) == 0;                   // k >= 2 * tile: one direction per tile
            for (int j = tile >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < tile; i += T) {
                    const int ixj = i ^ j;
                    if (ixj > i) { const uint64_t x = lds[i], y = lds[ixj]; if ((x > y) == asc) { lds[i] = y; lds[ixj] = x; } }
                }
                __syncthreads();
            }
            for (int i = tid; i < tile; i += T) g[base + i] = lds[i];
            __syncthreads();
        }
    }
    return 0;
}

// block-wide STABLE LSD radix sort of n uint64 keys on the bit field [(key >> shift) - base] & (2^nbits - 1), 4 bits per pass,
// ping-ponging between a and b (both in HBM). cnt: 16 * blockDim.x ints of LDS, scan: >= 17 ints. returns the buffer holding the result.
// every thread of the workgroup must call it. blockDim.x <= 256.
__device__ inline uint64_t* vmx_block_radix_sort_u64(uint64_t* a, uint64_t* b, int n, int shift, uint64_t base, int nbits, int* cnt, int* scan) {
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int tile = (n + T - 1) / T;
    const int lo = tid * tile < n ? tid * tile : n;
    const int hi = lo + tile < n ? lo + tile : n;
    for (int bit = 0; bit < nbits; bit += 4) {
        for (int d = 0; d < 16; ++d) cnt[d * T + tid] = 0;
        for (int i = lo; i < hi; ++i) { const int d = (int)((((a[i] >> shift) - base) >> bit) & 15); cnt[d * T + tid]++; }
        __syncthreads();
        // exclusive scan over the 16*T counters in (digit, thread) order: thread tid owns entries [16*tid, 16*tid+16)
        int s = 0;
        for (int e = 0; e < 16; ++e) s += cnt[16 * tid + e];
        int tot; int ex = vmx_block_excl_scan(s, scan, &tot);
        for (int e = 0; e < 16; ++e) { const int v = cnt[16 * tid + e]; cnt[16 * tid + e] = ex; ex += v; }
        __syncthreads();
        for (int i = lo; i < hi; ++i) { const uint64_t k = a[i]; const int d = (int)((((k >> shift) - base) >> bit) & 15); b[cnt[d * T + tid]++] = k; }
        __syncthreads();
        uint64_t* t = a; a = b; b = t;
    }
    return a;
}

__host__ __device__ __forceinline__ uint8_t vmx_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

#endif
