// kernels_fast_3.hip -- k_scan_fast<NF=3, ...> instantiations (see scan_fast.h).
#include "scan_fast.h"

namespace sybl {

hipError_t launch_count_nf3(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_launch_nf<3>(E, ng, n_wg, st); }

hipError_t launch_emit_nf3(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_launch_nf<3>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_fast_nf3(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds,
                                hipStream_t st) {
    return fast_launch_nf<3>(P, ng, na, mode, time, gen, n_wg, lds, st);
}

}  // namespace sybl
