// What one loader worker does per block, without a GPU: prepare_block (csrc/loader.cpp: read the block's column files,
// gob-decode them, lay the pieces out in a slab) over the two fixture blocks of loader_parse_blocks.py, on T threads.
// This is the number the load's CPU side is made of (profiles/r04_loader_sweep.txt: 0.75 ms per block alone, 0.93 ms with
// 16 threads busy, on the GPU box's EPYC 9575F).  Build (from this directory, after the library) and run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip loader_parse.cpp -o loader_parse -I../../sybil_amd/csrc -I../../include \
//         -L../../sybil_amd -lsybilgpu -Wl,-rpath,$PWD/../../sybil_amd -Wl,-rpath,/opt/rocm/lib
//   python loader_parse_blocks.py /tmp/lp && ./loader_parse /tmp/lp 300 16
// Switches (read when the library loads): SYBL_GOB_NO_VBMI, SYBL_LOADER_NO_AVX512, SYBL_LOADER_WIDE_DECODE.
#include "../../sybil_amd/csrc/loader.cpp"

#include <thread>
using namespace sybl;

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: loader_parse <dir written by loader_parse_blocks.py> [blocks per thread = 300] [threads = 1]\n");
        return 2;
    }
    const std::string root = argv[1];
    const int reps = argc > 2 ? atoi(argv[2]) : 300, threads = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<ColSpec> specs;
    for (const char *n : {"c04", "c05", "c06", "c01", "c02", "c07", "c08"}) specs.push_back({n, SYBL_INT_VAL});
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&]() {
            std::vector<char> slab((size_t)8 << 20);
            for (int r = 0; r < reps; r++) {
                PreparedBlock pb = prepare_block(root + "/t/block00000000" + (r & 1 ? "1" : "2"), specs, slab.data(), slab.size(), 0);
                if (pb.unreadable || pb.broken) {
                    fprintf(stderr, "bad block\n");
                    exit(1);
                }
            }
        });
    for (auto &x : th) x.join();
    const double s = seconds_since(t0);
    printf("%d threads x %d blocks: %.3f s wall, %.3f ms per block per thread, %.2f ns per value\n", threads, reps, s, s * 1e3 / reps,
           s * 1e9 / reps / (7.0 * 65536));
    return 0;
}
