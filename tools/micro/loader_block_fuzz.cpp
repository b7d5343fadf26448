// prepare_block (csrc/loader.cpp: what a loader worker does with a block directory -- decode the column files, lay the pieces
// out in a slab) on damaged blocks, without a GPU: a block directory is copied, one of its files damaged (bytes overwritten,
// inserted, removed, the tail cut off, a huge length planted, the file removed), and the block prepared into a heap slab of
// exactly the size load_blocks gives a worker.  Nothing is checked but that the call returns a verdict -- build with the
// sanitizers so that a write past the slab or a read past a decoded array stops the run:
//   hipcc --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -std=c++17 -x hip loader_block_fuzz.cpp -o loader_block_fuzz \
//         -I../../sybil_amd/csrc -I../../include -L../../sybil_amd -lsybilgpu -Wl,-rpath,$PWD/../../sybil_amd -Wl,-rpath,/opt/rocm/lib
//   ./loader_block_fuzz <block dir> <scratch dir> <trials> name:type ...        (type: 1 int, 2 str, 3 set)
// SYBL_LOADER_TWO_PASS=1: the two-pass form of every block.  SYBL_LOADER_GPU_VARINT=1: the worker half of the default load.
#include "../../sybil_amd/csrc/loader.cpp"

#include <dirent.h>
#include <random>
using namespace sybl;

static std::vector<uint8_t> slurp(const std::string &p) {
    std::vector<uint8_t> b;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return b;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}
static void spit(const std::string &p, const std::vector<uint8_t> &b) {
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) exit(3);
    if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: loader_block_fuzz <block dir> <scratch dir> <trials> name:type ...\n");
        return 2;
    }
    const std::string src = argv[1], dst = argv[2];
    const int trials = atoi(argv[3]);
    std::vector<ColSpec> specs;
    for (int a = 4; a < argc; a++) {
        std::string s = argv[a];
        const size_t c = s.find(':');
        specs.push_back({s.substr(0, c), atoi(s.c_str() + c + 1)});
    }
    std::vector<std::string> files;
    if (DIR *d = opendir(src.c_str())) {
        while (dirent *e = readdir(d))
            if (ends_with(e->d_name, ".db")) files.push_back(e->d_name);
        closedir(d);
    }
    if (files.empty()) return 2;
    std::vector<std::vector<uint8_t>> orig;
    mkdir(dst.c_str(), 0755);
    for (auto &f : files) {
        orig.push_back(slurp(src + "/" + f));
        spit(dst + "/" + f, orig.back());
    }
    // (the slab a worker gets: load_blocks' per-column bound, as sybl_debug_block_layout computes it)
    size_t cap = 65536;
    for (auto &sp : specs) cap += sp.type == SYBL_SET_VAL ? 0 : (size_t)65536 * (sp.type == SYBL_STR_VAL ? 12 : 8) + ((size_t)96 << 10);
    // SYBL_LOADER_GPU_VARINT=1: the worker half of the default load -- the int columns' slices located, their file bytes copied
    // into the slab (what the GPU then makes of them is tests/test_gpu_loader_varint.py's fuzzer's business)
    size_t xcap = 0;
    if (const char *e = getenv("SYBL_LOADER_GPU_VARINT"))
        if (atoi(e) != 0)
            for (auto &sp : specs)
                if (sp.type == SYBL_INT_VAL) cap += (size_t)65536 + 4096, xcap += (size_t)65536 * 8 + (size_t)kGobMaxBins * 72 + 4096;
    std::mt19937_64 rng(777);
    long ok = 0, broken = 0, unreadable = 0, thrown = 0, small_slab = 0;
    for (int t = 0; t <= trials; t++) {
        const size_t fi = rng() % files.size();
        std::vector<uint8_t> b = orig[fi];
        bool removed = false;
        if (t > 0 && !b.empty()) {
            const int kind = (int)(rng() % 6);
            const size_t n = b.size();
            auto place = [&]() { return (size_t)(rng() % 3 == 0 ? rng() % n : rng() % std::min<size_t>(n, 400)); };
            if (kind == 0) {
                for (int k = 0, m = 1 + (int)(rng() % 4); k < m; k++) b[place()] = (uint8_t)rng();
            } else if (kind == 1) {
                b.resize(rng() % n);
            } else if (kind == 2) {
                const size_t at = place();
                b.insert(b.begin() + (long)at, (size_t)(1 + rng() % 9), (uint8_t)rng());
            } else if (kind == 3) {
                const size_t at = place();
                b.erase(b.begin() + (long)at, b.begin() + (long)std::min(n, at + 1 + rng() % 9));
            } else if (kind == 4) {
                const size_t at = place();
                b[at] = 0xF8;
                for (size_t k = 1; k <= 8 && at + k < n; k++) b[at + k] = (uint8_t)(rng() % 4 == 0 ? 0xFF : rng());
            } else {
                removed = true;
            }
        }
        if (removed) remove((dst + "/" + files[fi]).c_str());
        else spit(dst + "/" + files[fi], b);
        // (now and then a slab too small for the block: the streamed pass must notice BEFORE it writes)
        const size_t this_cap = rng() % 8 == 0 ? (size_t)(rng() % 200000) : cap;
        small_slab += this_cap != cap;
        std::unique_ptr<char[]> slab(new char[this_cap ? this_cap : 1]);
        try {
            PreparedBlock pb = prepare_block_unguarded(dst, specs, slab.get(), this_cap, -1, true, xcap);
            if (pb.unreadable) unreadable++;
            else if (pb.broken) broken++;
            else ok++;
        } catch (const std::exception &) {
            thrown++;  // (prepare_block turns it into "unreadable")
        }
        spit(dst + "/" + files[fi], orig[fi]);
    }
    printf("%ld laid out, %ld broken, %ld unreadable, %ld exceptions; %ld with a slab too small\n", ok, broken, unreadable, thrown, small_slab);
    return 0;
}
