// vmx_rows.h — ROW-scoped cross-lane operations (gfx950): a row is 16 consecutive lanes, the unit of the DPP row controls. The chain kernels of
// k_chain_rows.hip give every row of a wavefront its own read (four reads per wave), so everything one lane needs from another stays inside the
// row, and rows may sit in different branches: the exec mask switches whole rows on and off, row_shr never crosses a row, ds_bpermute reads a
// lane of the caller's own row, and a ballot is cut down to the 16 bits of the caller's row.
// Under -DVMX_EMU (tests/emu: lanes are fibers) the same operations are rendezvous points of the row's 16 lanes only.
#ifndef VMX_ROWS_H
#define VMX_ROWS_H
#include "vmx_device.h"

#ifdef VMX_EMU
__device__ __forceinline__ unsigned vmx_row_ballot(bool p) { return emu_row_ballot(p ? 1 : 0); }
// lane t of the row <- lane t - N of `v`; the first N lanes keep `old`
template <int N> __device__ __forceinline__ int vmx_row_shr_i32(int old, int v) {
    const int l16 = vmx_lane() & 15;
    const int e = emu_row_exchange(v, (l16 - N) & 15);
    return l16 >= N ? e : old;
}
// value of lane `src16` (0..15, the same in every lane of the row) of the caller's row
__device__ __forceinline__ int vmx_row_get_i32(int v, int src16) { return emu_row_exchange(v, src16); }
// the row's earlier stores to memory are visible to the row's later loads
__device__ __forceinline__ void vmx_row_sync() { hipemu::row_barrier(); }
#else
__device__ __forceinline__ unsigned vmx_row_ballot(bool p) { return (unsigned)(__ballot(p) >> (vmx_lane() & 48)) & 0xffffu; }
template <int N> __device__ __forceinline__ int vmx_row_shr_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x110 + N, 0xf, 0xf, false); }
__device__ __forceinline__ int vmx_row_get_i32(int v, int src16) { return __builtin_amdgcn_ds_bpermute(((vmx_lane() & 48) + src16) << 2, v); }
// one wavefront: its memory operations are issued in order; what is needed is that the stores have been acknowledged (s_waitcnt vmcnt(0))
// before a later load of the same address is issued, and that the compiler keeps the order
__device__ __forceinline__ void vmx_row_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
#endif

template <int N> __device__ __forceinline__ double vmx_row_shr_f64(double old, double v) {
    union { double d; int i[2]; } o, s, r; o.d = old; s.d = v;
    r.i[0] = vmx_row_shr_i32<N>(o.i[0], s.i[0]); r.i[1] = vmx_row_shr_i32<N>(o.i[1], s.i[1]);
    return r.d;
}
template <int N> __device__ __forceinline__ long long vmx_row_shr_i64(long long old, long long v) {
    union { long long d; int i[2]; } o, s, r; o.d = old; s.d = v;
    r.i[0] = vmx_row_shr_i32<N>(o.i[0], s.i[0]); r.i[1] = vmx_row_shr_i32<N>(o.i[1], s.i[1]);
    return r.d;
}
__device__ __forceinline__ double vmx_row_get_f64(double v, int src16) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_row_get_i32(u.i[0], src16); u.i[1] = vmx_row_get_i32(u.i[1], src16);
    return u.d;
}
__device__ __forceinline__ long long vmx_row_get_i64(long long v, int src16) {
    union { long long d; int i[2]; } u; u.d = v;
    u.i[0] = vmx_row_get_i32(u.i[0], src16); u.i[1] = vmx_row_get_i32(u.i[1], src16);
    return u.d;
}
// inclusive prefix maximum along the row (Kogge-Stone on row_shr:1/2/4/8; a lane without a source keeps its own value)
__device__ __forceinline__ double vmx_row_incl_max_f64(double v) {
    double t;
    t = vmx_row_shr_f64<1>(v, v); v = t > v ? t : v;
    t = vmx_row_shr_f64<2>(v, v); v = t > v ? t : v;
    t = vmx_row_shr_f64<4>(v, v); v = t > v ? t : v;
    t = vmx_row_shr_f64<8>(v, v); v = t > v ? t : v;
    return v;
}
#endif
