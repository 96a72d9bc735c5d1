// vmx_ext_state.h — per-read state and kernel arguments of the extend stage (k_ext.hip / vmx_stage_extend.hip).
#ifndef VMX_EXT_STATE_H
#define VMX_EXT_STATE_H
#include "vmx_kernels.h"
#include "vmx_extend.h"
#include "../../include/vacmapx.h"

struct vmx_ext_read {
    int32_t status;        // 0 or VM_READ_*
    int32_t active;        // has a local chain of >= 2 anchors
    int32_t nseg, nseg_snap;
    int32_t filtered;      // drop_misplaced removed something (:19278)
    int32_t redo, pass;    // pass 1 = rerun with nofilter (:24080)
    int32_t skip_ext;      // no second extend_edge_test round
    int32_t prob_base, prob_n;   // problem slots of the current round
    int32_t dp_base, dp_n;       // gap-fill problems actually used
    int32_t nrec;
    int32_t pad;
};

struct vmx_ext_args {
    int32_t n_reads, nseq, local_maxdiff, nodiscard, hardclip, redo_only, mode, spread;     // spread: lanes of the grid per read in k_ext_phase / k_ext_records
    int32_t asm_long, pad_;       // -mode asm, contig of 500 kb and more: ass_extend_func (mammap_asm.py:23423) — small_alignment 30, no divergence filter
    double maxdivergence;
    const uint8_t* ocodes; const int64_t* roff; const uint8_t* ref; const int64_t* coff;
    const vmx_anchor* chain; const int32_t* chain_len; const int64_t* la_off; const int32_t* lstatus;
    const double* gscore; const int32_t* mapq;
    vmx_ext_read* er;
    // per-read pools: coff3 = 3*chain_len+8 anchors, soff = chain_len+2 ints
    const int64_t* coff3; const int64_t* soff;
    vmx_anchor* segA; int32_t* st; int32_t* en; vmx_anchor* segA_snap; int32_t* st_snap; int32_t* en_snap;
    int32_t* seg_prob; int32_t* dup;
    // problems of the current round
    vmx_pair_desc* desc; const vmx_pair_desc* desc_prev; int32_t* round_count; int64_t round_cap; int32_t* overflow;
    int32_t* prob_read;          // owner (read) of every problem slot of the round, written when the slots are allocated
    const int64_t* ed_out; const int32_t* ext_te; const int32_t* ext_qe;
    // records
    vm_record* rec; char* rec_blob; const int64_t* blob_off; int64_t* rec_coff; int32_t* rec_clen; double* dup_d;
};
#endif
