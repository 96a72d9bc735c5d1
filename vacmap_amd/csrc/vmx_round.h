// vmx_round.h — arguments of k_round_prep (k_round.hip): one launch per DP round of the extend stage for lengths, offset scans, pool-size scans, problem
// table and traceback chunk plan.
#ifndef VMX_ROUND_H
#define VMX_ROUND_H
#include "vmx_kernels.h"
#include "vmx_extend.h"

#define VMX_MAX_CHUNKS 24
#define VMX_ROUND_WGS 128            // workgroups of k_round_prep (<= 256: one look-back step per thread)
#define VMX_ROUND_PART_WORDS (256 * 8 + 8)      // int64 words of the publication area: 8 per workgroup (sums [0..5], flag [7]) + the finished-workgroups counter

struct vmx_round_args {
    const vmx_pair_desc* desc; const int32_t* n_prob;
    int64_t* off[6];            // exclusive prefix sums, n + 1 entries each: [0] target string bytes, [1] query string bytes; gap-fill rounds: [2] traceback bytes,
                                // [3] boundary ints, [4] run words, [5] CIGAR bytes
    int64_t* tb_size;           // gap-fill rounds: the problem's QUEUE KEY for the size-order kernels (k_size_hist / k_size_scatter sort by quarter octaves of it, largest first):
                                // [2^42, ..) problems taken one per task (traceback bytes > VMX_HEAD_THRESH), [2^33, 2^42) other problems beyond the small class,
                                // [2^23, 2^33) small-class problems with 4-byte slots (ns 3 / 4), [2^13, 2^23) 2-byte (ns 2), [2^3, 2^13) 1-byte (ns 1), [1, 8) no band; inside a range by size
    vmx_dp_prob* probs;         // gap-fill rounds: the problem table
    int64_t* part;              // VMX_ROUND_PART_WORDS words, zeroed once when allocated
    int32_t* stat_out;          // the round's problem count for the host's statistics block (may be null)
    int64_t* plan_out;          // gap-fill rounds: [0] n, [1..4] pool totals, [5] target bytes, [6] query bytes, [7] number of chunks m (-1: more than VMX_MAX_CHUNKS),
                                // [8 ..] first problem of every chunk (m + 1 entries), then the traceback offsets at those problems (m + 1 entries)
    int64_t tb_limit;           // traceback bytes per chunk
    int64_t cap[4];             // gap-fill rounds, unplanned pass (cap[0] > 0): capacity of the traceback (one chunk) / boundary / run / CIGAR pools in their units; a problem
                                // that would pass one of them is EMPTIED (tl = ql = 0, offsets 0): nothing is filled or traced for it, the host sees the totals and retries
    int ad_on, ad_match, ad_o1, ad_e1, ad_o2, ad_e2, ad_pct;      // gap-fill rounds of the batched path: the band-width rule's inputs (vmx_ad_ns) — a small problem's traceback space
                                // is sized for the slot width of its own band (VMX_AD_W) and its queue key carries that width (below)
    long long epoch;            // launch number of this context (never 0): flags of earlier launches are never mistaken for this one's
};
template <bool DP> __global__ void k_round_prep(vmx_round_args A);
#endif
