// vmx_stage.h — device-buffer bundles and internal stage functions shared by the stage entry points and vm_align_batch.
#ifndef VMX_STAGE_H
#define VMX_STAGE_H
#include "vmx_host.h"

struct vm_index_view {
    const uint8_t* codes; const int64_t* coff; int nseq; int64_t total_len;
    const uint64_t* pos; const vmx_slot* table; int table_bits; int k, w, mid_occ;
};
void vmx_index_view(const vm_index* mi, vm_index_view* v);

// a chain handed to the extend stage instead of the seed / chain stages (host rows, DESCENDING read order like a local chain)
struct vmx_preset { const vmx_anchor* chain_desc; int64_t len; };      // one per read of the call (align_device takes an array of n)

struct vmx_local_bufs {
    vmx::DevBuf sq, dst, rorder, pc2, stg, si, tg, cntp, fp, pp, prep_ws;
    vmx::DevBuf guide_rows, guide_len, ng_used, ng_total, cnt, cur, tpos, hkey, hkey2, dbg, hval, hq, goff, pcnt, gkey, gq, gr, epoch;
    vmx::DevBuf la_rows, la_ekey, la_sorted, la_off, la_cnt, status, gap, rlist, S, P, SA, chain, chain_len, score, variant;
    int64_t la_pool_rows = 0;            // rows of the local-anchor pools (regular slots + overflow area)
    std::vector<int64_t> h_la_off;
    std::vector<int32_t> h_la_cnt;
    void release() {
        vmx::DevBuf* all[] = {&prep_ws, &sq, &dst, &rorder, &pc2, &stg, &si, &tg, &cntp, &fp, &pp, &dbg, &epoch, &hkey2, &guide_rows, &guide_len, &ng_used, &ng_total, &cnt, &cur, &tpos, &hkey, &hval, &hq, &goff, &pcnt, &gkey, &gq, &gr,
                              &la_rows, &la_ekey, &la_sorted, &la_off, &la_cnt, &status, &gap, &rlist, &S, &P, &SA, &chain, &chain_len, &score, &variant};
        for (auto* b : all) b->release();
    }
};
vmx_local_bufs* vmx_ctx_local_bufs(vm_ctx* c);

extern "C" int vmx_seed_stage(vm_ctx* c, const vm_index* mi, int check_num, int mid_occ, int64_t n, const uint8_t* d_codes, const int64_t* d_roff, int64_t total_bases,
                   vmx::DevBuf* B, std::vector<int64_t>& h_koff, std::vector<int64_t>& h_nhits, vmx::DevBuf* arena = nullptr, int64_t** rows_out = nullptr);
int vmx_local_stage(vm_ctx* c, const vm_index_view& ix, const vm_params* prm, int64_t n, const uint8_t* d_ocodes, const int64_t* d_roff,
                    const std::vector<int64_t>& h_roff, const vmx_anchor* d_path_rows, const int32_t* d_path_len, const int32_t* d_npaths,
                    const int64_t* d_aoff, const std::vector<int64_t>& h_aoff, const double* d_gscore, vmx_local_bufs& L, bool fast = false);
// G1 selection (k_chain_select) over n reads, one launch per LDS size class of the reads' anchor counts (h_aoff), side by side
int vmx_launch_chain_select(vm_ctx* c, int64_t n, const int64_t* h_aoff, vmx::DevBuf& d_list, const vmx_anchor* sorted, const int64_t* d_aoff, const int64_t* d_lens,
                            const double* S, const int32_t* P, const int32_t* SA, const int64_t* gmax, const int32_t* flip, int mode, char* scr, const int64_t* soff,
                            int32_t* d_mapq, double* d_score, int32_t* d_np, int32_t* plen, vmx_anchor* prow);
#endif
