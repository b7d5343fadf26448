// Issue rate of the vector instructions the row bodies lean on (gfx950): every SIMD runs 8 waves, each a loop of 32
// independent copies of one instruction; reported as cycles per wave-instruction per SIMD relative to v_add_u32.
// build: hipcc --offload-arch=gfx950 -O3 -o valurate valurate.hip ; run: ./valurate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
// operands: %0..%3 32-bit registers, %4..%7 64-bit register pairs (read and written), %8 %9 32-bit inputs, %10 %11 64-bit
// inputs; a statement issues four independent instructions, REP8 makes it 32 per iteration
#define KERNEL(NAME, INS4)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t x, uint32_t y) {                     \
        uint32_t a0 = threadIdx.x, a1 = x, a2 = y, a3 = 3;                                                              \
        uint64_t q0 = threadIdx.x, q1 = x, q2 = y, q3 = 3;                                                              \
        const uint64_t x64 = x | 1ull << 40, y64 = y | 3ull << 33;                                                      \
        for (int it = 0; it < iters; it++) {                                                                            \
            REP8(asm volatile(INS4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)    \
                              : "v"(x), "v"(y), "v"(x64), "v"(y64) : "vcc", "s20", "s21", "s22", "s23");)               \
        }                                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3);                        \
    }

KERNEL(k_add_u32, "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %9, %2\n v_add_u32 %3, %9, %3\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %9, %2\n v_mul_lo_u32 %3, %9, %3\n")
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %8, %0\n v_mul_hi_u32 %1, %8, %1\n v_mul_hi_u32 %2, %9, %2\n v_mul_hi_u32 %3, %9, %3\n")
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %8, %0\n v_mul_u32_u24 %1, %8, %1\n v_mul_u32_u24 %2, %9, %2\n v_mul_u32_u24 %3, %9, %3\n")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %8, %0, %9\n v_mad_u32_u24 %1, %8, %1, %9\n v_mad_u32_u24 %2, %9, %2, %8\n v_mad_u32_u24 %3, %9, %3, %8\n")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n")
KERNEL(k_mul_f32, "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %9, %2\n v_mul_f32 %3, %9, %3\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %9\n v_lshl_add_u32 %3, %3, 3, %9\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %9\n v_cmp_lt_u32 vcc, %3, %9\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n")
KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n")
KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %9, %8\n v_perm_b32 %3, %3, %9, %8\n")
KERNEL(k_cvt_f64_u32, "v_cvt_f64_u32 %4, %0\n v_cvt_f64_u32 %5, %1\n v_cvt_f64_u32 %6, %2\n v_cvt_f64_u32 %7, %3\n")
KERNEL(k_cvt_u32_f64, "v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %5\n v_cvt_u32_f64 %2, %6\n v_cvt_u32_f64 %3, %7\n")
KERNEL(k_mul_f64, "v_mul_f64 %4, %10, %4\n v_mul_f64 %5, %10, %5\n v_mul_f64 %6, %11, %6\n v_mul_f64 %7, %11, %7\n")
KERNEL(k_fma_f64, "v_fma_f64 %4, %10, %4, %11\n v_fma_f64 %5, %10, %5, %11\n v_fma_f64 %6, %11, %6, %10\n v_fma_f64 %7, %11, %7, %10\n")
KERNEL(k_cmp_i64, "v_cmp_lt_i64 vcc, %4, %10\n v_cmp_lt_i64 vcc, %5, %10\n v_cmp_lt_i64 vcc, %6, %11\n v_cmp_lt_i64 vcc, %7, %11\n")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %9, %8, %6\n v_mad_u64_u32 %7, vcc, %9, %8, %7\n")
KERNEL(k_add_co_u32, "v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %9, %2\n v_add_co_u32 %3, vcc, %9, %3\n")

// LDS atomics without a returned value, every lane its own word (no conflicts): 32 per iteration
template <typename T>
__global__ __launch_bounds__(256) void k_lds_add(uint32_t *out, int iters, uint32_t x, uint32_t y) {
    __shared__ T tab[256 * 4];
    for (int i = threadIdx.x; i < 256 * 4; i += 256) tab[i] = 0;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            __hip_atomic_fetch_add(tab + threadIdx.x + 256 * (k & 3), (T)(x + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)tab[threadIdx.x];
}
// the same with a random-looking word per lane out of 1024 (what a cell table sees): bank conflicts and same-address hits
template <typename T>
__global__ __launch_bounds__(256) void k_lds_add_rand(uint32_t *out, int iters, uint32_t x, uint32_t y) {
    __shared__ T tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = 0;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            h = h * 1664525u + 1013904223u;
            __hip_atomic_fetch_add(tab + (h >> 22), (T)(x + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)tab[threadIdx.x];
}

typedef void (*Kern)(uint32_t *, int, uint32_t, uint32_t);

static double run(Kern k, uint32_t *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n_wg = 256 * 8;  // 256 CUs x 8 workgroups of 4 waves: 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), 0, 0, out, 10, 12345u, 678u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), 0, 0, out, iters, 12345u, 678u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 5000;
    struct { const char *name; Kern k; } ks[] = {
        {"v_add_u32", k_add_u32},         {"v_mul_lo_u32", k_mul_lo_u32},   {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_mul_f32", k_mul_f32},
        {"v_lshl_add_u32", k_lshl_add},   {"v_cndmask_b32", k_cndmask},     {"v_cmp_lt_u32", k_cmp_u32},    {"v_readlane_b32", k_readlane},
        {"v_bfe_u32", k_bfe_u32},         {"v_perm_b32", k_perm_b32},       {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_u32_f64", k_cvt_u32_f64},
        {"v_mul_f64", k_mul_f64},         {"v_fma_f64", k_fma_f64},         {"v_cmp_lt_i64", k_cmp_i64},    {"v_lshlrev_b64", k_lshl_b64},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_add_co_u32", k_add_co_u32},
        {"ds_add_u32", k_lds_add<uint32_t>}, {"ds_add_u64", k_lds_add<unsigned long long>},
        {"ds_add_u32 rand", k_lds_add_rand<uint32_t>}, {"ds_add_u64 rand", k_lds_add_rand<unsigned long long>}};
    const double base = run(k_add_u32, out, iters);
    // wave-instructions per SIMD: 8 waves x iters x 32
    const double per_simd = 8.0 * iters * 32;
    printf("# %d iterations x 32 instructions x 8 waves per SIMD; v_add_u32: %.3f ms = %.2f ns per wave-instruction per SIMD\n", iters, base,
           base * 1e6 / per_simd);
    for (auto &k : ks) {
        const double ms = run(k.k, out, iters);
        printf("%-16s %8.3f ms  %5.2f x v_add_u32\n", k.name, ms, ms / base);
    }
    return 0;
}
