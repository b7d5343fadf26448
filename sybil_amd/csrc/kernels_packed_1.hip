// kernels_packed_1.hip -- k_scan_packed<NF=1, ...> instantiations (see scan_packed.h).
#include "scan_packed.h"

namespace sybl {

hipError_t launch_count_packed_nf1(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_packed_launch_nf<1>(E, ng, n_wg, st); }

hipError_t launch_emit_packed_nf1(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_packed_launch_nf<1>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_packed_nf1(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st) {
    return packed_launch_nf<1>(P, ng, na, mode, time, n_wg, lds, st);
}

}  // namespace sybl
