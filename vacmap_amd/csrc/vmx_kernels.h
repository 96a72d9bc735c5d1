// vmx_kernels.h — device-side data types shared by the kernels and the host orchestration.
#ifndef VMX_KERNELS_H
#define VMX_KERNELS_H
#include <stdint.h>

// one DP problem of the gap-fill stage (E5); offsets into the per-batch pools
struct vmx_dp_prob {
    int64_t t_off, q_off;     // into the target / query code pools
    int32_t tl, ql;
    int64_t tb_off;           // traceback bytes: ((tl+63)/64) * (ql+63) * 64
    int64_t bnd_off;          // int32 units: 3*(ql+1)
    int64_t run_off;          // uint32 units: tl+ql+2
    int64_t cig_off;          // bytes: 2*(tl+ql)+16
};

// anchor in the layout the chain kernels use (16 B): SURVEY §8(a) row T
struct vmx_anchor {
    int32_t q;     // read position
    int16_t l;     // length
    int16_t s;     // strand +1 / -1
    int64_t r;     // global reference position
};

// cost tables on the device
struct vmx_tables {
    const float* extra; int32_t extra_n;
    const float* readgap_h; const float* readgap_r; const float* large_readgap;
    const double* log2cache; int32_t log2cache_n;
    const double* log2int;
};

#define VMX_GC_BYTES_PER_ANCHOR 33   // q4 + r8 + ls4 + S8 + P4 + SA4 + cov1 (LDS bytes per anchor in k_chain_global)

#endif
