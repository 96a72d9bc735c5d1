// vmx_local_dev.h — device helpers shared by the two local re-seeding kernels (k_local.hip: the general form over HBM-staged hit pools;
// k_local_band.hip: the guide-banded, LDS-tiled form). SURVEY §8(a) row L2, /root/reference/src/vacmap/mammap_clrnano.py:23069-23345.
#ifndef VMX_LOCAL_DEV_H
#define VMX_LOCAL_DEV_H
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_local.h"

__device__ __forceinline__ vmx_anchor vmx_mk_anchor(long long q, long long r, int s, long long l) {
    vmx_anchor a; a.q = (int32_t)q; a.r = r; a.s = (int16_t)s; a.l = (int16_t)l; return a;
}

__device__ __forceinline__ int vmx_pos2contig(const int64_t* __restrict__ coff, int nseq, long long pos) {   // :51-59
    int lo = 0, hi = nseq;                        // bisection: same index as the reference's linear scan of the contig starts
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}

// findClosest_1 :17560-17582 on the guide sorted by read position (gq ascending)
__device__ __forceinline__ void vmx_find_closest(const int* gq, int n, int target, int& b0, int& b1, int& i0, int& i1) {
    if (target <= gq[0]) { b0 = b1 = gq[0] - target; i0 = i1 = 0; return; }
    if (target >= gq[n - 1]) { b0 = b1 = target - gq[n - 1]; i0 = i1 = n - 1; return; }
    int i = 0, j = n, mid = 0;
    while (i < j) {
        mid = (i + j) >> 1;
        if (gq[mid] == target) { b0 = b1 = 0; i0 = i1 = mid; return; }
        if (target < gq[mid]) j = mid; else i = mid + 1;
    }
    b0 = gq[j - 1] - target; if (b0 < 0) b0 = -b0;
    b1 = gq[j] - target; if (b1 < 0) b1 = -b1;
    i0 = j - 1; i1 = j;
}

__device__ __forceinline__ uint32_t vmx_kmer_at(const uint8_t* s, long long x, int k, bool& ok) {
    uint32_t v = 0; ok = true;
    for (int i = 0; i < k; ++i) { uint8_t c = s[x + i]; if (c > 3) ok = false; v = (v << 2) | (uint32_t)(c & 3); }
    return v;
}
__device__ __forceinline__ uint32_t vmx_kmer_rc(uint32_t fw, int k) {
    uint32_t rv = 0;
    for (int i = 0; i < k; ++i) { rv = (rv << 2) | (3 - (fw & 3)); fw >>= 2; }
    return rv;
}

#endif
