// vmo_chain_fast.cc — CPU ORACLE (test infrastructure): heuristic "_fast" chain variants.
// PLACEHOLDER in this commit: both entry points report "not restated yet" (negative status -> read skipped).
#include "vmo_internal.h"
namespace vmo {
int64_t chain_global_fast(const std::vector<Anchor>&, int, double, int, int, std::vector<double>&, std::vector<int64_t>&, std::vector<int64_t>&) {
    set_error("GC-fast not restated yet");
    return -2;
}
int local_chain_fast(const std::vector<Anchor>&, int, double, int, int, bool, int, double*, Path&) {
    set_error("LC-fast not restated yet");
    return -3;
}
}  // namespace vmo
