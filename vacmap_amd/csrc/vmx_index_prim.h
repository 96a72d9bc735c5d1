// vmx_index_prim.h — device-wide sort / run-length / scan used by the index build (vmx_index_prim.hip, rocPRIM underneath).
// Every function follows the two-call convention: tmp == nullptr only reports *tmp_bytes. Returns a hipError_t as int.
#ifndef VMX_INDEX_PRIM_H
#define VMX_INDEX_PRIM_H
#ifdef VMX_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stddef.h>
int vmx_prim_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout, size_t n, int end_bit, hipStream_t s);
int vmx_prim_rle_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, size_t n, uint64_t* uniq, uint32_t* counts, uint64_t* n_runs, hipStream_t s);
int vmx_prim_excl_scan_u32(void* tmp, size_t* tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s);
int vmx_prim_excl_scan_i64(void* tmp, size_t* tmp_bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t s);
#endif
