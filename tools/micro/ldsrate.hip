// LDS atomic issue rates on gfx950, per compute unit: what bounds the row bodies once the loads are hidden (round 6, config 2).
// 256 workgroups x 1024 threads; every lane owns replica (lane & 63) of a cell table [field][16 cells][64 replicas] -- the layout of
// k_scan_packed's cell table at low cardinality -- and performs, per "row", a mix of operations on a pseudo-random cell.
//   build: hipcc --offload-arch=gfx950 -O3 -o ldsrate ldsrate.hip ; run: ./ldsrate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kThreads = 1024, kCells = 16, kRows = 4096;

template <int MIX>
__global__ __launch_bounds__(kThreads) void k(uint32_t seed, uint64_t *out) {
    __shared__ uint64_t tab[6 * kCells * 64];
    for (int i = threadIdx.x; i < 6 * kCells * 64; i += kThreads) tab[i] = 0;
    __syncthreads();
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    const uint32_t rep = threadIdx.x & 63u;
    constexpr uint32_t F = kCells * 64;  // words per field
    for (int r = 0; r < kRows; r++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t cell = x >> 28, v = (x >> 8) & 0xFFFFFu;
        uint64_t *p = tab + cell * 64u + rep;
        uint32_t *p32 = (uint32_t *)p;
        if (MIX == 0) {  // 3 x ds_add_u64 (count, sum, sum)
            __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + F, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 2 * F, (uint64_t)(v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 1) {  // 2 x ds_add_u64 (count packed into the first sum)
            __hip_atomic_fetch_add(p, (1ull << 40) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + F, (uint64_t)(v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 2) {  // 3 x ds_add_u32
            __hip_atomic_fetch_add(p32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p32 + 2 * F, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p32 + 4 * F, v ^ 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 3) {  // 1 x ds_add_u64
            __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 4) {  // 1 x ds_add_u32
            __hip_atomic_fetch_add(p32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 5) {  // config 2 as it is: 3 x ds_add_u64 + 2 x ds_max_u32
            __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + F, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + 2 * F, (uint64_t)(v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(p32 + 6 * F, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(p32 + 8 * F, v ^ 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 6) {  // packed: 2 x ds_add_u64 + 2 x ds_max_u32
            __hip_atomic_fetch_add(p, (1ull << 40) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + F, (uint64_t)(v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(p32 + 6 * F, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(p32 + 8 * F, v ^ 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 9) {  // round 6's config 2: count packed, 2 x ds_add_u64, gate reads of an unreplicated uint32 table
            __hip_atomic_fetch_add(p, (1ull << 40) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(p + F, (uint64_t)(v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t *g = (uint32_t *)(tab + 4 * F) + cell;
            if (v > g[0]) __hip_atomic_fetch_max(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((v ^ 5u) > g[kCells]) __hip_atomic_fetch_max(g + kCells, v ^ 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MIX == 7) {  // no LDS at all: the loop's own cost
            if (v == 0xFFFFFu && cell == 99u) tab[0] = 1;
        } else if (MIX == 8) {  // 1 x ds_add_u64 + 1 x ds_max_u64 packed? (two maxima cannot share a word: one 64-bit max, for the rate)
            __hip_atomic_fetch_add(p, (1ull << 40) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(p + F, (uint64_t)v << 32 | (v ^ 5u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[1] == 0x1234567u) out[0] = tab[3];
}

int main() {
    uint64_t *out;
    hipMalloc((void **)&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"3 x ds_add_u64", "2 x ds_add_u64 (count packed)", "3 x ds_add_u32", "1 x ds_add_u64", "1 x ds_add_u32",
                           "3 x ds_add_u64 + 2 x ds_max_u32 (config 2 today)", "2 x ds_add_u64 + 2 x ds_max_u32 (packed)", "no LDS operation", "1 x ds_add_u64 + 1 x ds_max_u64",
                           "2 x ds_add_u64 + 2 gated maxima, unreplicated uint32 (round 6)"};
    auto run = [&](int mix, auto kern) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), 0, 0, 12345u + it, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        const double rows = 256.0 * kThreads * kRows;
        printf("%-52s %.4f ms  %.2f G rows/s  %.3f ns per wave-row per CU\n", names[mix], best, rows / best / 1e6, best * 1e6 / (kRows * 16.0));
    };
    run(0, k<0>); run(1, k<1>); run(2, k<2>); run(3, k<3>); run(4, k<4>); run(5, k<5>); run(6, k<6>); run(7, k<7>); run(8, k<8>); run(9, k<9>);
    return 0;
}
