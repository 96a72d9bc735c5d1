// vmx_link.h — device-resident state of the batch-linked chain DPs of -mode asm (k_chain_linked.hip, vmx_asm.hip).
#ifndef VMX_LINK_H
#define VMX_LINK_H
#include "vmx_kernels.h"

#define VM_LINK_BAILED (-31)        // GC-exact's opcount bail-out (mammap_asm.py:21757): the fork's linked GC-fast (:21871) is not on the device
#define VM_LINK_UNSUPPORTED (-32)   // the carried slice would reach the cold entries, or more anchors to carry than the staging area holds
#define VM_LINK_RAISED (-33)        // the reference raises here (:23266, IndexError in the traceback :23284-23292)

// what one assembly contig carries from batch to batch (assembly_get_readmap_DP_test :23222-23272). Lives in HBM; only the kernels touch it.
struct vmx_link_state {
    double g_max_scores; int32_t g_max_index; int32_t n_pre; long long prereadloc;
    int32_t status;               // 0, or VM_LINK_*
    int32_t pre_g_max_index, have; // :23245 pre_g_max_index of the last batch that ran (the traceback starts there), whether any ran
    int32_t last_base, last_n;
    int32_t n_saved;              // batches "saved" so far (the reference's datacount - 1)
    int32_t cap_pre;              // capacity of the three staging arrays
    double* pre_S; int32_t* pre_P; vmx_anchor* pre_rows;
};

// one batch of one contig
struct vmx_link_job {
    vmx_link_state* state;
    vmx_anchor* rows;             // [cap_pre + n_new]: the carried rows right-aligned below cap_pre (k_link_place), the new anchors (sorted by q) from cap_pre on
    double* S; int32_t* P;        // same geometry as rows
    int32_t* SA;                  // hot part of the score-sorted index (up to n_pre + n_new entries)
    int32_t cap_pre, n_new;
    int32_t* Si; int64_t* T; int32_t* CNT;   // linked GC-fast only (k_chain_linked_fast): int(S), diagonal key, S_i_count[last q + 50]; null otherwise
    long long* dbg;               // optional tuning counters (VMX_ASM_TIME): anchors, insertions through HBM, scan blocks past the window, position advances, candidates
    // results
    int32_t ran, saved, n, hot; long long n_cold; double cold_max; long long gmax, opcount;
};
#endif
