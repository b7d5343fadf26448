// Loads-only microbenchmark of row -> lane mappings for the config-3 column mix (3 x 2-byte, 2 x 1-byte, 2 x 4-byte columns,
// 16 stored bytes per row): what the load pattern alone delivers, without any row body.  R rows per lane and tile:
//   R = 4  : what k_scan_packed does (one dwordx4 per 2- / 4-byte column -- half of it range-checked away for 2 bytes --
//            and one dword per 1-byte column)
//   R = 8  : 4-byte columns two dwordx4 per lane (lane stride 32 B), 2-byte one dwordx4, 1-byte one dwordx2
//   R = 16 : 4-byte four dwordx4 (lane stride 64 B), 2-byte two, 1-byte one
// build: hipcc --offload-arch=gfx950 -O3 -o loadpat loadpat.hip ; run: ./loadpat [rows]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct Cols { const uint8_t *c[7]; int w[7]; };
constexpr int kThreads = 1024;

template <int BYTES>  // bytes this lane reads of one column for one tile, as 16 / 8 / 4-byte buffer loads
__device__ __forceinline__ uint32_t load_lane(const uint8_t *wave_base, uint32_t wave_bytes, uint32_t lane_off) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wave_base, 0, (int)wave_bytes, 0x00020000);
    uint32_t acc = 0;
    if (BYTES >= 16) {
#pragma unroll
        for (int k = 0; k < BYTES / 16; k++) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off + 16 * k), 0, 2);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    } else if (BYTES == 8) {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, 0, 2);
        acc ^= v.x ^ v.y;
    } else {
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off, 0, 2);
    }
    return acc;
}

// R rows per lane and tile; OVER: 2-byte columns read with a full dwordx4 per 4 rows (the current kernel's range-checked over-read)
template <int R, bool OVER>
__global__ __launch_bounds__(kThreads) void k_loads(Cols C, int64_t rows, uint32_t *out) {
    const int64_t per_wg = (rows / gridDim.x) / (kThreads * R) * (kThreads * R);
    const int64_t start = (int64_t)blockIdx.x * per_wg;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int64_t t = 0; t < per_wg; t += kThreads * R) {
        const int64_t wave_row = start + t + (int64_t)wave * 64 * R;
#pragma unroll
        for (int c = 0; c < 7; c++) {
            const int w = c < 3 ? 2 : c < 5 ? 1 : 4;  // compile-time widths of the mix
            const uint8_t *base = C.c[c] + wave_row * w;
            const uint32_t wave_bytes = 64u * R * w, off = lane * R * w;
            if (w == 4) acc ^= load_lane<R * 4>(base, wave_bytes, off);
            else if (w == 2) acc ^= (OVER && R == 4) ? load_lane<16>(base, wave_bytes, off) : load_lane<R * 2>(base, wave_bytes, off);
            else acc ^= load_lane<R>(base, wave_bytes, off);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 1000000000LL;
    Cols C;
    const int widths[7] = {2, 2, 2, 1, 1, 4, 4};
    for (int c = 0; c < 7; c++) {
        C.w[c] = widths[c];
        void *p;
        if (hipMalloc(&p, (size_t)rows * widths[c] + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(p, c + 1, (size_t)rows * widths[c] + 4096);
        C.c[c] = (const uint8_t *)p;
    }
    uint32_t *out;
    hipMalloc((void **)&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int wgs) {
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), 0, 0, C, rows, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        printf("%-34s wgs %4d  %.3f ms  %.0f GB/s of 16 B/row\n", name, wgs, best, rows * 16.0 / (best * 1e-3) / 1e9);
    };
    for (int wgs : {256, 512}) {
        run("R=4 (current: 2-byte over-read)", k_loads<4, true>, wgs);
        run("R=4 exact (dwordx2 for 2-byte)", k_loads<4, false>, wgs);
        run("R=8", k_loads<8, false>, wgs);
        run("R=16", k_loads<16, false>, wgs);
    }
    return 0;
}
