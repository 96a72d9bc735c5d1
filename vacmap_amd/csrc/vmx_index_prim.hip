// vmx_index_prim.hip — device-wide primitives of the index BUILD (not on the per-read path): radix sort of the reference's
// (hash, position) pairs, run-length encoding of the sorted hashes, exclusive scans. These are plain library operations, so they come
// from rocPRIM (the ROCm counterpart of "hipBLASLt only for plain library GEMMs"); every kernel of the read path is hand-written.
// Kept in its own translation unit: the rocPRIM templates compile slowly and nothing else needs them.
#ifdef VMX_EMU
// TEST-ONLY emulator build (tests/emu): "device" memory is host memory, so the same contracts are met with the C++ standard library.
#include "vmx_index_prim.h"
#include <algorithm>
#include <numeric>
#include <vector>
int vmx_prim_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout, size_t n, int end_bit, hipStream_t) {
    if (!tmp) { *tmp_bytes = 8; return 0; }
    const uint64_t mask = end_bit >= 64 ? ~0ULL : ((1ULL << end_bit) - 1);
    std::vector<size_t> ix(n); std::iota(ix.begin(), ix.end(), (size_t)0);
    std::stable_sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return (kin[a] & mask) < (kin[b] & mask); });
    for (size_t i = 0; i < n; ++i) { kout[i] = kin[ix[i]]; vout[i] = vin[ix[i]]; }
    return 0;
}
int vmx_prim_rle_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, size_t n, uint64_t* uniq, uint32_t* counts, uint64_t* n_runs, hipStream_t) {
    if (!tmp) { *tmp_bytes = 8; return 0; }
    uint64_t r = 0;
    for (size_t i = 0; i < n;) { size_t j = i; while (j < n && kin[j] == kin[i]) ++j; uniq[r] = kin[i]; counts[r] = (uint32_t)(j - i); ++r; i = j; }
    *n_runs = r;
    return 0;
}
int vmx_prim_excl_scan_u32(void* tmp, size_t* tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t) {
    if (!tmp) { *tmp_bytes = 8; return 0; }
    uint32_t a = 0; for (size_t i = 0; i < n; ++i) { const uint32_t v = in[i]; out[i] = a; a += v; }
    return 0;
}
int vmx_prim_excl_scan_i64(void* tmp, size_t* tmp_bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t) {
    if (!tmp) { *tmp_bytes = 8; return 0; }
    int64_t a = 0; for (size_t i = 0; i < n; ++i) { const int64_t v = in[i]; out[i] = a; a += v; }
    return 0;
}
#else
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "vmx_index_prim.h"

int vmx_prim_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint64_t* vin, uint64_t* vout, size_t n, int end_bit, hipStream_t s) {
    return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)end_bit, s);
}
int vmx_prim_rle_u64(void* tmp, size_t* tmp_bytes, const uint64_t* kin, size_t n, uint64_t* uniq, uint32_t* counts, uint64_t* n_runs, hipStream_t s) {
    return (int)rocprim::run_length_encode(tmp, *tmp_bytes, kin, n, uniq, counts, n_runs, s);
}
int vmx_prim_excl_scan_u32(void* tmp, size_t* tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    return (int)rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}
int vmx_prim_excl_scan_i64(void* tmp, size_t* tmp_bytes, const int64_t* in, int64_t* out, size_t n, hipStream_t s) {
    return (int)rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), s);
}
#endif
