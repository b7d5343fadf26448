// kernels_packed_2.hip -- k_scan_packed<NF=2, ...> instantiations (see scan_packed.h).
#include "scan_packed.h"

namespace sybl {

hipError_t launch_count_packed_nf2(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_packed_launch_nf<2>(E, ng, n_wg, st); }

hipError_t launch_emit_packed_nf2(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_packed_launch_nf<2>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_packed_nf2(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st) {
    return packed_launch_nf<2>(P, ng, na, mode, time, n_wg, lds, st);
}

}  // namespace sybl
