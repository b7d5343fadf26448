// kernels_packed_0.hip -- k_scan_packed<NF=0, ...> instantiations (see scan_packed.h).
#include "scan_packed.h"

namespace sybl {

hipError_t launch_count_packed_nf0(const EmitPlan &E, int ng, int n_wg, hipStream_t st) { return count_packed_launch_nf<0>(E, ng, n_wg, st); }

hipError_t launch_emit_packed_nf0(const EmitPlan &E, int ng, int na, int n_wg, hipStream_t st) {
    return emit_packed_launch_nf<0>(E, ng, na, n_wg, st);
}

hipError_t launch_scan_packed_nf0(const FastPlan &P, int ng, int na, int mode, bool time, int n_wg, size_t lds, hipStream_t st) {
    return packed_launch_nf<0>(P, ng, na, mode, time, n_wg, lds, st);
}

hipError_t launch_count_packed(const EmitPlan &E, int nf, int ng, int n_wg, hipStream_t st) {
    switch (nf) {
    case 0: return launch_count_packed_nf0(E, ng, n_wg, st);
    case 1: return launch_count_packed_nf1(E, ng, n_wg, st);
    case 2: return launch_count_packed_nf2(E, ng, n_wg, st);
    case 3: return launch_count_packed_nf3(E, ng, n_wg, st);
    case 4: return launch_count_packed_nf4(E, ng, n_wg, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_emit_packed(const EmitPlan &E, int nf, int ng, int na, int n_wg, hipStream_t st) {
    switch (nf) {
    case 0: return launch_emit_packed_nf0(E, ng, na, n_wg, st);
    case 1: return launch_emit_packed_nf1(E, ng, na, n_wg, st);
    case 2: return launch_emit_packed_nf2(E, ng, na, n_wg, st);
    case 3: return launch_emit_packed_nf3(E, ng, na, n_wg, st);
    case 4: return launch_emit_packed_nf4(E, ng, na, n_wg, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_packed(const FastPlan &P, int nf, int ng, int na, int mode, bool time, int n_wg, size_t lds_bytes,
                              hipStream_t st) {
    if (ng < 0 || ng > kFastMaxG || na < 0 || na > kFastTemplatedA) return hipErrorInvalidValue;
    switch (nf) {
    case 0: return launch_scan_packed_nf0(P, ng, na, mode, time, n_wg, lds_bytes, st);
    case 1: return launch_scan_packed_nf1(P, ng, na, mode, time, n_wg, lds_bytes, st);
    case 2: return launch_scan_packed_nf2(P, ng, na, mode, time, n_wg, lds_bytes, st);
    case 3: return launch_scan_packed_nf3(P, ng, na, mode, time, n_wg, lds_bytes, st);
    case 4: return launch_scan_packed_nf4(P, ng, na, mode, time, n_wg, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sybl
