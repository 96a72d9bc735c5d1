// vmx_rows.h — ROW-scoped cross-lane operations (gfx950): a row is 16 consecutive lanes, the unit of the DPP row controls. The chain kernels of
// k_chain_rows.hip give every row of a wavefront its own read (four reads per wave), so everything one lane needs from another stays inside the
// row, and rows may sit in different branches: the exec mask switches whole rows on and off, row_shr never crosses a row, ds_bpermute reads a
// lane of the caller's own row, and a ballot is cut down to the 16 bits of the caller's row.
// Under -DVMX_EMU (tests/emu: lanes are fibers) the same operations are rendezvous points of the row's 16 lanes only.
#ifndef VMX_ROWS_H
#define VMX_ROWS_H
#include "vmx_device.h"

// vmx_rmask: a predicate of the lanes as a bit mask that can be combined with & and | before the row's 16 bits are taken out of it
// (vmx_row_bits): on the hardware the 64-bit SGPR pair a compare writes — combining masks is scalar work, one v_lshrrev_b64 per mask
// that a lane needs for itself; on the emulator the row's 16 bits from the start.
#ifdef VMX_EMU
typedef unsigned vmx_rmask;
__device__ __forceinline__ vmx_rmask vmx_mask(bool p) { return emu_row_ballot(p ? 1 : 0); }
__device__ __forceinline__ unsigned vmx_row_bits(vmx_rmask m) { return m; }
__device__ __forceinline__ unsigned vmx_row_ballot(bool p) { return emu_row_ballot(p ? 1 : 0); }
// lane t of the row <- lane t - N of `v`; the first N lanes keep `old`
template <int N> __device__ __forceinline__ int vmx_row_shr_i32(int old, int v) {
    const int l16 = vmx_lane() & 15;
    const int e = emu_row_exchange(v, (l16 - N) & 15);
    return l16 >= N ? e : old;
}
// lane t of the row <- lane t - N of `v`; the first N lanes get 0
template <int N> __device__ __forceinline__ int vmx_row_shr0_i32(int v) {
    const int l16 = vmx_lane() & 15;
    const int e = emu_row_exchange(v, (l16 - N) & 15);
    return l16 >= N ? e : 0;
}
__device__ __forceinline__ double vmx_max_f64(double a, double b) { return a > b ? a : b; }
template <int N> __device__ __forceinline__ int vmx_row_ror_i32(int v) { return emu_row_exchange(v, ((vmx_lane() & 15) - N) & 15); }
// value of lane `src16` (0..15, the same in every lane of the row) of the caller's row
__device__ __forceinline__ int vmx_row_get_i32(int v, int src16) { return emu_row_exchange(v, src16); }
// the row's earlier stores to memory are visible to the row's later loads
__device__ __forceinline__ void vmx_row_sync() { hipemu::row_barrier(); }
#else
typedef unsigned long long vmx_rmask;
__device__ __forceinline__ vmx_rmask vmx_mask(bool p) { return __builtin_amdgcn_ballot_w64(p); }       // p: ONE compare (a combination of predicates goes through a 0 / 1 register)
__device__ __forceinline__ unsigned vmx_row_bits(vmx_rmask m) { return (unsigned)(m >> (vmx_lane() & 48)) & 0xffffu; }
__device__ __forceinline__ unsigned vmx_row_ballot(bool p) { return (unsigned)(__builtin_amdgcn_ballot_w64(p) >> (vmx_lane() & 48)) & 0xffffu; }   // (the compare's own SGPR pair: no 0 / 1 detour)
template <int N> __device__ __forceinline__ int vmx_row_shr_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x110 + N, 0xf, 0xf, false); }
template <int N> __device__ __forceinline__ int vmx_row_shr0_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true); }      // bound_ctrl:1 — nothing to preset
template <int N> __device__ __forceinline__ int vmx_row_ror_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false); }      // rotation inside the row
// one v_max_f64 (the scores are never NaN; `a > b ? a : b` is a compare and two selects under -fno-fast-math)
__device__ __forceinline__ double vmx_max_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
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
template <int N> __device__ __forceinline__ double vmx_row_shr0_f64(double v) {
    union { double d; int i[2]; } s, r; s.d = v;
    r.i[0] = vmx_row_shr0_i32<N>(s.i[0]); r.i[1] = vmx_row_shr0_i32<N>(s.i[1]);
    return r.d;
}
// inclusive prefix maximum along the row of max(v, 0) (Kogge-Stone on row_shr:1/2/4/8). A lane without a source reads +0.0 (bound_ctrl:1:
// two v_mov_dpp and one v_max_f64 per step, no copy to preset), so the result is exact wherever it is positive — the callers compare it with
// a running maximum that is positive from the start.
__device__ __forceinline__ double vmx_row_incl_max0_f64(double v) {
    v = vmx_max_f64(v, vmx_row_shr0_f64<1>(v));
    v = vmx_max_f64(v, vmx_row_shr0_f64<2>(v));
    v = vmx_max_f64(v, vmx_row_shr0_f64<4>(v));
    v = vmx_max_f64(v, vmx_row_shr0_f64<8>(v));
    return v;
}
template <int N> __device__ __forceinline__ double vmx_row_ror_f64(double v) {
    union { double d; int i[2]; } s, r; s.d = v;
    r.i[0] = vmx_row_ror_i32<N>(s.i[0]); r.i[1] = vmx_row_ror_i32<N>(s.i[1]);
    return r.d;
}
// maximum over the row, in every lane (butterfly on row_ror:8/4/2/1; no LDS)
__device__ __forceinline__ double vmx_row_allmax_f64(double v) {
    v = vmx_max_f64(v, vmx_row_ror_f64<8>(v)); v = vmx_max_f64(v, vmx_row_ror_f64<4>(v));
    v = vmx_max_f64(v, vmx_row_ror_f64<2>(v)); v = vmx_max_f64(v, vmx_row_ror_f64<1>(v));
    return v;
}
__device__ __forceinline__ int vmx_row_allmax_i32(int v) {
    int t;
    t = vmx_row_ror_i32<8>(v); v = t > v ? t : v; t = vmx_row_ror_i32<4>(v); v = t > v ? t : v;
    t = vmx_row_ror_i32<2>(v); v = t > v ? t : v; t = vmx_row_ror_i32<1>(v); v = t > v ? t : v;
    return v;
}
#endif
