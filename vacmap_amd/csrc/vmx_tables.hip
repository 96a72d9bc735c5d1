// vmx_tables.hip — cost tables C0, built on the host with libm + the NumPy patch list and uploaded once per
// context. Bit-identical to the reference's tables: extra (/root/reference/src/vacmap/mammap_clrnano.py:15371-15376),
// readgapcost_list (:26567-26569; R mode mammap_noprefercloser.py:16534), large_readgapcost_list (:28270-28275),
// log2cache (:27530) and log2(g) for the per-call gapcost_list (:24846, :27320).
#include "vmx_host.h"
#include "vmx_tables_patch.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace vmx {

static void patch32(std::vector<float>& v, const uint64_t (*p)[2], int n) {
    for (int i = 0; i < n; ++i) { uint32_t b = (uint32_t)p[i][1]; memcpy(&v[p[i][0]], &b, 4); }
}
static void patch64(std::vector<double>& v, const uint64_t (*p)[2], int n) {
    for (int i = 0; i < n; ++i) { uint64_t b = p[i][1]; memcpy(&v[p[i][0]], &b, 8); }
}

static HostTables build_tables() {
    HostTables t;
    for (int64_t g = 0;; ++g) {
        double a = 30 + 0.5 * std::log((double)std::max<int64_t>(g, 1));
        double b = std::min(10.0, (double)g / 100) + std::min(30.0, (double)g / 1000);
        double v = std::min(36.0, std::min(a, b));
        t.extra.push_back((float)v);
        if (t.extra.size() > 1 && v == 36.0) break;
    }
    patch32(t.extra, EXTRA_PATCH, EXTRA_NPATCH);
    t.readgap_h.assign(100, 0.f); t.readgap_r.assign(100, 0.f); t.large_readgap.assign(100, 0.f);
    for (int r = 1; r < 100; ++r) {
        t.readgap_h[r] = (float)(0.1 * std::log2((double)(r + 1)));
        t.readgap_r[r] = (float)(0.1 * std::log2((double)r));
        t.large_readgap[r] = r >= 30 ? (float)(0.5 * r) : (float)(0.1 * std::log2((double)(r + 1)));
    }
    patch32(t.readgap_h, READGAP_H_PATCH, READGAP_H_NPATCH);
    patch32(t.readgap_r, READGAP_R_PATCH, READGAP_R_NPATCH);
    patch32(t.large_readgap, LARGE_READGAP_PATCH, LARGE_READGAP_NPATCH);
    t.log2cache.resize(100000);
    for (int g = 0; g < 100000; ++g) t.log2cache[g] = 0.5 * std::log2((double)(g + 1));
    patch64(t.log2cache, LOG2CACHE_PATCH, LOG2CACHE_NPATCH);
    t.log2int.resize(1025); t.log2int[0] = 0;
    for (int g = 1; g < 1025; ++g) t.log2int[g] = std::log2((double)g);
    patch64(t.log2int, LOG2INT_PATCH, LOG2INT_NPATCH);
    return t;
}

const HostTables& host_tables() { static const HostTables T = build_tables(); return T; }

}  // namespace vmx
