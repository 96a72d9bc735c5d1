// vmo_tables.cc — CPU ORACLE (test infrastructure): cost tables C0, bit-identical to the reference's NumPy tables.
#include "vmo_internal.h"
#include "vmo_tables_patch.h"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace vmo {

static void patch32(std::vector<float>& v, const uint64_t (*p)[2], int n) {
    for (int i = 0; i < n; ++i) { uint32_t b = (uint32_t)p[i][1]; memcpy(&v[p[i][0]], &b, 4); }
}
static void patch64(std::vector<double>& v, const uint64_t (*p)[2], int n) {
    for (int i = 0; i < n; ++i) { uint64_t b = p[i][1]; memcpy(&v[p[i][0]], &b, 8); }
}

static Tables build() {
    Tables t;
    // extra (mammap_clrnano.py:15371-15376): min(36, 30 + 0.5*ln(max(g,1)), min(10, g/100) + min(30, g/1000)) until == 36
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

const Tables& tables() { static const Tables T = build(); return T; }

}  // namespace vmo

extern "C" {
// table export for tests (golden V8): which = 0 extra,1 readgap_h,2 readgap_r,3 large_readgap (f32); 4 log2cache,5 log2int (f64)
int64_t vmo_table(int which, const void** data) {
    const vmo::Tables& t = vmo::tables();
    switch (which) {
        case 0: *data = t.extra.data(); return (int64_t)t.extra.size();
        case 1: *data = t.readgap_h.data(); return (int64_t)t.readgap_h.size();
        case 2: *data = t.readgap_r.data(); return (int64_t)t.readgap_r.size();
        case 3: *data = t.large_readgap.data(); return (int64_t)t.large_readgap.size();
        case 4: *data = t.log2cache.data(); return (int64_t)t.log2cache.size();
        case 5: *data = t.log2int.data(); return (int64_t)t.log2int.size();
    }
    return -1;
}
}
