/* Does it matter how many hipMemcpyAsync calls carry a batch?  The packed form of a whole genome is ten arrays, 85 MB in all; a rank's shard an eighth of that.
 * Times the bytes as ONE pinned H2D copy, as the ten pieces (sizes of the real arrays) on one stream, and the ten pieces from ONE pinned block.
 * build: hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_split_probe tools/copy_split_probe.cpp ; usage: /tmp/copy_split_probe [divide by: 1 | 8] */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t div = argc > 1 ? (size_t)atoi(argv[1]) : 1;
    const size_t nr = 3567178 / div, nv = 7953595 / div, na = 16000000 / div;
    /* start (8 B), length (2), two counts (1 + 1) per region; position (2), type | zygosity (1), two allele lengths (1 + 1) per call; allele bytes */
    const size_t sizes[10] = {nr * 8, nr * 2, nr, nr, nr /* contig */, nv * 2, nv, nv, nv, na};
    size_t total = 0;
    for (size_t s : sizes) total += (s + 255) & ~(size_t)255;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char *dev = nullptr, *block = nullptr;
    CK(hipMalloc((void **)&dev, total));
    CK(hipHostMalloc((void **)&block, total, hipHostMallocDefault));
    std::vector<char *> own(10);
    for (int i = 0; i < 10; ++i) CK(hipHostMalloc((void **)&own[i], sizes[i], hipHostMallocDefault));
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipStreamSynchronize(st));
            const double t0 = now_ms();
            size_t at = 0;
            if (mode == 0) CK(hipMemcpyAsync(dev, block, total, hipMemcpyHostToDevice, st));
            else
                for (int i = 0; i < 10; ++i) {
                    CK(hipMemcpyAsync(dev + at, mode == 1 ? own[i] : block + at, sizes[i], hipMemcpyHostToDevice, st));
                    at += (sizes[i] + 255) & ~(size_t)255;
                }
            CK(hipStreamSynchronize(st));
            const double t = now_ms() - t0;
            if (t < best) best = t;
        }
        printf("%-44s %.1f MB: %.3f ms = %.1f GB/s\n", mode == 0 ? "one copy" : (mode == 1 ? "ten copies, ten pinned allocations" : "ten copies out of one pinned block"), total / 1e6, best, total / best / 1e6);
    }
    return 0;
}
