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

// block-wide exclusive scan of one int per thread (blockDim.x <= 1024, multiple of 64); returns exclusive prefix,
// *total gets the block sum. scratch: >= 17 ints of shared memory.
__device__ __forceinline__ int vmx_block_excl_scan(int v, int* scratch, int* total) {
    int lane = vmx_lane(), w = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63) >> 6);
    int inc = vmx_wave_incl_scan_i32(v);
    __syncthreads();
    if (lane == 63) scratch[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int i = 0; i < nw; ++i) { int t = scratch[i]; scratch[i] = s; s += t; } scratch[16] = s; }
    __syncthreads();
    int base = scratch[w];
    *total = scratch[16];
    return base + inc - v;
}

// block-wide bitonic sort of N (power of two) uint64 keys living in HBM at g; staged through `lds` (lds_cap keys) when they fit.
// every thread of the workgroup must call it.
__device__ inline void vmx_block_sort_u64_impl(uint64_t* g, int N, uint64_t* lds, int lds_cap) {
    uint64_t* a = g;
    const bool in_lds = N <= lds_cap;
    if (in_lds) { for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) lds[i] = g[i]; a = lds; }
    __syncthreads();
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
    if (in_lds) { for (int i = (int)threadIdx.x; i < N; i += (int)blockDim.x) g[i] = lds[i]; __syncthreads(); }
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
