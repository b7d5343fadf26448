// kernels_fast_4.hip -- k_scan_fast<NF=4, ...> instantiations (see scan_fast.h).
#include "scan_fast.h"

namespace sybl {

hipError_t launch_count_nf4(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_launch_nf<4>(E, ng, n_wg, st); }

hipError_t launch_emit_nf4(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_launch_nf<4>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_fast_nf4(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds,
                                hipStream_t st) {
    return fast_launch_nf<4>(P, ng, na, mode, time, gen, n_wg, lds, st);
}

}  // namespace sybl
