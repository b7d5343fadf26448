// fake_hip.cpp -- TEST INFRASTRUCTURE for the test infrastructure: the handful of HIP runtime calls rccl_standin.cpp makes,
// over plain host memory, so that tests/test_rccl_standin.py can run the stand-in's protocol (chunking, barriers,
// reductions, in-place operands, mismatch detection) between processes on a box WITHOUT a GPU.  Preloaded in front of
// libamdhip64.so by that test only.
#include <stdlib.h>
#include <string.h>

extern "C" {
typedef int hipError_t;
typedef void *hipStream_t;
typedef void (*hipHostFn_t)(void *);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int, hipStream_t) { memmove(dst, src, n); return 0; }
hipError_t hipLaunchHostFunc(hipStream_t, hipHostFn_t fn, void *arg) { fn(arg); return 0; }
hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
hipError_t hipDeviceSynchronize(void) { return 0; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 2; }
hipError_t hipHostFree(void *p) { free(p); return 0; }
const char *hipGetErrorString(hipError_t) { return "fake hip"; }
}
