// hip_emu.h — TEST-ONLY CPU emulator of the small HIP subset libvacmapx uses (tests/emu/, never shipped).
//
// Purpose: run the PRODUCT's own kernels (vacmap_amd/csrc/*.hip, unchanged) in the `-m "not gpu"` test suite of the
// GPU-less build container, so that kernel logic is checked against the oracle before GPU minutes are spent.
// It is NOT a fallback: vacmap_amd/ never loads the emulator library; it is built by tests/emu/build_emu.py into
// tests/emu/_build/ and opened only by tests. Model: one workgroup at a time per OS thread; each work-item is a
// ucontext fiber; cross-lane operations (__shfl*, __ballot, __syncthreads, ...) are rendezvous points, so — exactly as
// on the hardware when the code is written with wave-uniform collectives — every live lane of a wave must reach
// each collective. Wave size is 64.
#ifndef HIP_EMU_H
#define HIP_EMU_H
#include <ucontext.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::cur()->dynshared;

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef void* hipStream_t;
typedef struct hipEmuEvent { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; size_t totalGlobalMem; };

namespace hipemu {

struct Block;
struct Lane {
    ucontext_t ctx;
    void* sp = nullptr;             // saved stack pointer of the fiber (x86-64 fast switch, hip_emu.cpp)
    char* stack = nullptr;
    Block* blk = nullptr;
    unsigned tid = 0;
    bool done = false;
};

struct Rendezvous { int count = 0; int gen = 0; };

struct Block {
    dim3 grid, block, bidx;
    unsigned nthreads = 0;
    std::vector<Lane> lanes;
    ucontext_t sched;
    void* sched_sp = nullptr;
    unsigned cur = 0;
    std::vector<uint64_t> slots;          // one exchange slot per lane
    std::vector<uint64_t> slots2;
    std::vector<Rendezvous> wave_rv;      // per wave
    std::vector<Rendezvous> row_rv;       // per 16-lane row (the chain kernels' row-scoped collectives, vmx_rows.h)
    std::vector<int> row_alive;
    Rendezvous block_rv;
    std::vector<int> wave_alive;
    int block_alive = 0;
    void* dynshared = nullptr;
    std::function<void()> body;
};

extern thread_local Block* g_blk;
inline Block* cur() { return g_blk; }
inline Lane& me() { return g_blk->lanes[g_blk->cur]; }
void yield();
void wave_barrier();
void row_barrier();
void block_barrier();
void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
double now_ms();

}  // namespace hipemu

struct EmuIdx { unsigned x, y, z; };
#define threadIdx (EmuIdx{hipemu::me().tid % hipemu::cur()->block.x, (hipemu::me().tid / hipemu::cur()->block.x) % hipemu::cur()->block.y, hipemu::me().tid / (hipemu::cur()->block.x * hipemu::cur()->block.y)})
#define blockIdx (EmuIdx{hipemu::cur()->bidx.x, hipemu::cur()->bidx.y, hipemu::cur()->bidx.z})
#define blockDim (hipemu::cur()->block)
#define gridDim (hipemu::cur()->grid)
#define warpSize 64

inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T> inline T emu_exchange(T v, int src_lane_in_wave) {
    static_assert(sizeof(T) <= 8, "emu shfl: type too wide");
    hipemu::Block* b = hipemu::cur();
    unsigned tid = hipemu::me().tid;
    unsigned wbase = tid & ~63u;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    b->slots[tid] = raw;
    hipemu::wave_barrier();
    unsigned src = wbase + ((unsigned)src_lane_in_wave & 63u);
    uint64_t got = (src < b->nthreads) ? b->slots[src] : raw;
    hipemu::wave_barrier();
    T r; memcpy(&r, &got, sizeof(T));
    return r;
}
// row-scoped collectives: only the (up to) 16 lanes of the caller's row meet, so the rows of a wave may sit in different branches
// (on the hardware: DPP row operations and ds_bpermute under an exec mask that switches whole rows on and off)
template <class T> inline T emu_row_exchange(T v, int src_lane_in_row) {
    static_assert(sizeof(T) <= 8, "emu row exchange: type too wide");
    hipemu::Block* b = hipemu::cur();
    unsigned tid = hipemu::me().tid;
    unsigned rbase = tid & ~15u;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    b->slots2[tid] = raw;
    hipemu::row_barrier();
    unsigned src = rbase + ((unsigned)src_lane_in_row & 15u);
    uint64_t got = (src < b->nthreads && !b->lanes[src].done) ? b->slots2[src] : raw;
    hipemu::row_barrier();
    T r; memcpy(&r, &got, sizeof(T));
    return r;
}
inline unsigned emu_row_ballot(int pred) {
    hipemu::Block* b = hipemu::cur();
    unsigned tid = hipemu::me().tid, rbase = tid & ~15u;
    b->slots2[tid] = pred ? 1 : 0;
    hipemu::row_barrier();
    unsigned m = 0;
    for (unsigned l = 0; l < 16 && rbase + l < b->nthreads; ++l)
        if (!b->lanes[rbase + l].done && b->slots2[rbase + l]) m |= 1u << l;
    hipemu::row_barrier();
    return m;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::me().tid & 63; int base = lane & ~(width - 1);
    return emu_exchange(v, base + (src & (width - 1)));
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::me().tid & 63; int base = lane & ~(width - 1);
    int s = lane - (int)d; if (s < base) s = lane;
    return emu_exchange(v, s);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::me().tid & 63; int base = lane & ~(width - 1);
    int s = lane + (int)d; if (s >= base + width) s = lane;
    return emu_exchange(v, s);
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
    int lane = hipemu::me().tid & 63;
    return emu_exchange(v, lane ^ m);
}
inline unsigned long long __ballot(int pred) {
    hipemu::Block* b = hipemu::cur();
    unsigned tid = hipemu::me().tid, wbase = tid & ~63u;
    b->slots[tid] = pred ? 1 : 0;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64 && wbase + l < b->nthreads; ++l)
        if (!b->lanes[wbase + l].done && b->slots[wbase + l]) m |= 1ULL << l;
    hipemu::wave_barrier();
    return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) {
    hipemu::Block* b = hipemu::cur();
    unsigned wbase = hipemu::me().tid & ~63u;
    unsigned long long live = 0;
    for (unsigned l = 0; l < 64 && wbase + l < b->nthreads; ++l) if (!b->lanes[wbase + l].done) live |= 1ULL << l;
    return __ballot(p) == live;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

// atomics: work-items of one block are fibers on one OS thread; blocks may run on several OS threads -> real atomics
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicCAS(T* p, T c, T v) { __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return c; }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// host API subset
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamDefault = 0, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }      // one level: the priority scheme stays off
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); p->multiProcessorCount = 8; strcpy(p->name, "hip_emu"); strcpy(p->gcnArchName, "emu"); p->totalGlobalMem = 1ull << 34; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "emu error"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEmuEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEmuEvent{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = hipemu::now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); })

#endif
