// wg_header.h -- device code shared by every scan kernel: the header counters at the end of a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sybl {

#ifdef __HIPCC__
// Header counters (matched rows, overflow, ...) at the end of a kernel: the waves' sums are added up in LDS and ONE lane of the
// workgroup adds the total to the header word.  An atomic per WAVE on one device-scope address is 4096 atomics that the
// memory side serialises -- measured at ~35 us at the end of every scan launch (s_memtime around fast_finish: config 2 took
// 0.183 ms for 100 M rows of which 0.13 are streaming; profiles/r06_cfg2_fixed_cost.txt).  v: the wave's sum (any lane's copy);
// every thread of the workgroup must call this the same number of times (it holds two barriers).
template <int N>
__device__ __forceinline__ void wg_header_add(int64_t *hdr, const int (&slot)[N], const int64_t (&v)[N]) {
    __shared__ unsigned long long acc[N];
    const uint32_t tid = threadIdx.x;
    if (tid < (uint32_t)N) acc[tid] = 0;
    __syncthreads();
    if ((tid & 63u) == 0) {
#pragma unroll
        for (int i = 0; i < N; i++)
            if (v[i]) __hip_atomic_fetch_add(&acc[i], (unsigned long long)v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (tid < (uint32_t)N && acc[tid]) __hip_atomic_fetch_add(hdr + slot[tid], (int64_t)acc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#endif

}  // namespace sybl
