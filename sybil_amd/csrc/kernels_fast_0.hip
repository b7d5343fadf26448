// kernels_fast_0.hip -- k_scan_fast<NF=0, ...> instantiations (see scan_fast.h).
#include "scan_fast.h"

namespace sybl {

hipError_t launch_count_nf0(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_launch_nf<0>(E, ng, n_wg, st); }

hipError_t launch_emit_nf0(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_launch_nf<0>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_fast_nf0(const FastPlan &P, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds,
                                hipStream_t st) {
    return fast_launch_nf<0>(P, ng, na, mode, time, gen, n_wg, lds, st);
}

hipError_t launch_scan_fast(const FastPlan &P, int nf, int ng, int na, int mode, bool time, bool gen, int n_wg, size_t lds_bytes,
                            hipStream_t st) {
    if (ng < 0 || ng > kFastMaxG || na < 0 || na > kFastTemplatedA) return hipErrorInvalidValue;
    switch (nf) {
    case 0: return launch_scan_fast_nf0(P, ng, na, mode, time, gen, n_wg, lds_bytes, st);
    case 1: return launch_scan_fast_nf1(P, ng, na, mode, time, gen, n_wg, lds_bytes, st);
    case 2: return launch_scan_fast_nf2(P, ng, na, mode, time, gen, n_wg, lds_bytes, st);
    case 3: return launch_scan_fast_nf3(P, ng, na, mode, time, gen, n_wg, lds_bytes, st);
    case 4: return launch_scan_fast_nf4(P, ng, na, mode, time, gen, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_count(const EmitPlan &E, int nf, int ng, int n_wg, hipStream_t st) {
    switch (nf) {
    case 0: return launch_count_nf0(E, ng, n_wg, st);
    case 1: return launch_count_nf1(E, ng, n_wg, st);
    case 2: return launch_count_nf2(E, ng, n_wg, st);
    case 3: return launch_count_nf3(E, ng, n_wg, st);
    case 4: return launch_count_nf4(E, ng, n_wg, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_emit(const EmitPlan &E, int nf, int ng, int na, int n_wg, hipStream_t st) {
    switch (nf) {
    case 0: return launch_emit_nf0(E, ng, na, n_wg, st);
    case 1: return launch_emit_nf1(E, ng, na, n_wg, st);
    case 2: return launch_emit_nf2(E, ng, na, n_wg, st);
    case 3: return launch_emit_nf3(E, ng, na, n_wg, st);
    case 4: return launch_emit_nf4(E, ng, na, n_wg, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sybl
