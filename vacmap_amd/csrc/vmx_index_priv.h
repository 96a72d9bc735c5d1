// vmx_index_priv.h — the index object and the build steps shared by vmx_index.hip (build / .vmx) and vmx_mmi.hip (minimap2 .mmi files).
#ifndef VMX_INDEX_PRIV_H
#define VMX_INDEX_PRIV_H
#include "vmx_host.h"
#include "vmx_index_prim.h"
#include <string>
#include <vector>

#include <mutex>
struct vm_index {
    vm_ctx* ctx = nullptr;
    int k = 0, w = 0, mid_occ = 10, table_bits = 0;
    std::vector<std::string> names;
    std::vector<int64_t> lens, offsets;     // offsets has nseq+1 entries
    std::string bases;                      // upper-case concatenation (host copy for Aligner.seq); empty on replicas (has_host_seq = false)
    int64_t n_min = 0, n_distinct = 0;
    vmx::DevBuf d_codes, d_pos, d_table, d_off;
    bool has_host_seq = true;
    std::once_flag host_seq_once;           // replicas decode their host copy of the bases from HBM once, whichever emitter thread asks first (vmx_sam.hip)
};

static inline unsigned grid1d(int64_t n, int64_t cap = 65536) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, cap)); }
#define VMX_PRIM(expr) do { int _e = (expr); if (_e != 0) return vmx::hip_fail((hipError_t)_e, #expr, __FILE__, __LINE__); } while (0)

// host bases (ASCII, one pointer per contig) -> mi->d_codes
int vmx_index_upload_codes(vm_index* mi, const char* const* seqs);
// sorted (hash, position) pairs on the device (positions in mi->d_pos) -> distinct keys, occurrence cap, hash table, contig offsets.
// d_keys is consumed (released).
int vmx_index_finish_device(vm_index* mi, vmx::DevBuf& d_keys, int64_t n);
// hash of every stored position recomputed from the codes (err[0] += positions that are not valid canonical k-mer starts inside one
// contig or whose strand bit is wrong) and the strict (hash, position) order check (err[1] += violations)
__global__ void k_idx_pos_keys(const uint8_t* codes, const int64_t* coff, int nseq, const uint64_t* pos, int64_t n, int k, uint64_t* keys, int32_t* err);
__global__ void k_idx_check_sorted(const uint64_t* keys, const uint64_t* pos, int64_t n, int32_t* err);
__global__ void k_idx_fill_hashes(const vmx_slot* tab, int64_t nslots, uint64_t* hashes);
#endif
