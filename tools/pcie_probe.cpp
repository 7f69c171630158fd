/* Measures what the host boundary of avk_compare_batch can count on, on the box it runs on: pinned and pageable H2D / D2H rates, the rate at which
 * the granted host threads copy pageable memory into pinned memory, hipHostRegister's cost, and a kernel reading pinned host memory directly.
 * build: hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_probe tools/pcie_probe.cpp -lpthread ; usage: /tmp/pcie_probe [MB] */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void sum_kernel(const uint4 *src, size_t n16, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    atomicAdd(out, s);
}
__global__ void fill_kernel(uint4 *dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const size_t bytes = mb << 20;
    int nthreads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("PROBE_THREADS")) nthreads = atoi(e);
    if (nthreads > 16) nthreads = 16;
    CK(hipSetDevice(0));
    void *d = nullptr, *hp = nullptr, *hp2 = nullptr;
    CK(hipMalloc(&d, bytes));
    double t0 = now_ms();
    CK(hipHostMalloc(&hp, bytes, hipHostMallocDefault));
    printf("hipHostMalloc %zu MB: %.1f ms\n", mb, now_ms() - t0);
    CK(hipHostMalloc(&hp2, bytes, hipHostMallocDefault));
    std::vector<char> pageable(bytes, 1);
    memset(hp, 2, bytes);
    memset(hp2, 3, bytes);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (size_t sz : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, bytes}) {
        if (sz > bytes) continue;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now_ms();
            CK(hipMemcpyAsync(d, hp, sz, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            const double h2d = now_ms() - t0;
            t0 = now_ms();
            CK(hipMemcpyAsync(hp2, d, sz, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            const double d2h = now_ms() - t0;
            if (rep == 2) printf("pinned %5zu MB: H2D %.3f ms = %.1f GB/s, D2H %.3f ms = %.1f GB/s\n", sz >> 20, h2d, sz / h2d / 1e6, d2h, sz / d2h / 1e6);
        }
    }
    { /* both directions at once */
        void *d2 = nullptr;
        CK(hipMalloc(&d2, bytes));
        t0 = now_ms();
        CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync(hp2, d2, bytes, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        const double both = now_ms() - t0;
        printf("pinned %zu MB H2D + %zu MB D2H at once: %.3f ms = %.1f GB/s each way\n", mb, mb, both, bytes / both / 1e6);
        /* the same H2D split over two streams */
        t0 = now_ms();
        CK(hipMemcpyAsync(d, hp, bytes / 2, hipMemcpyHostToDevice, s));
        CK(hipMemcpyAsync((char *)d + bytes / 2, (char *)hp + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s2));
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        const double two = now_ms() - t0;
        printf("pinned %zu MB H2D on two streams: %.3f ms = %.1f GB/s\n", mb, two, bytes / two / 1e6);
        (void)hipFree(d2);
    }
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now_ms();
        CK(hipMemcpy(d, pageable.data(), bytes, hipMemcpyHostToDevice));
        const double h2d = now_ms() - t0;
        t0 = now_ms();
        CK(hipMemcpy(pageable.data(), d, bytes, hipMemcpyDeviceToHost));
        const double d2h = now_ms() - t0;
        printf("pageable %zu MB: H2D %.3f ms = %.1f GB/s, D2H %.3f ms = %.1f GB/s\n", mb, h2d, bytes / h2d / 1e6, d2h, bytes / d2h / 1e6);
    }
    for (int nt : {1, 4, 8, nthreads}) { /* host threads copying pageable -> pinned */
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now_ms();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    const size_t lo = bytes * t / nt, hi = bytes * (t + 1) / nt;
                    memcpy((char *)hp + lo, pageable.data() + lo, hi - lo);
                });
            for (auto &x : th) x.join();
            const double c = now_ms() - t0;
            if (rep == 1) printf("host memcpy pageable -> pinned, %2d threads: %.3f ms = %.1f GB/s\n", nt, c, bytes / c / 1e6);
        }
    }
    {
        t0 = now_ms();
        hipError_t e = hipHostRegister(pageable.data(), bytes, hipHostRegisterDefault);
        const double reg = now_ms() - t0;
        if (e == hipSuccess) {
            t0 = now_ms();
            CK(hipMemcpyAsync(d, pageable.data(), bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            const double h2d = now_ms() - t0;
            t0 = now_ms();
            CK(hipHostUnregister(pageable.data()));
            printf("hipHostRegister %zu MB: %.1f ms; H2D from it %.3f ms = %.1f GB/s; unregister %.1f ms\n", mb, reg, h2d, bytes / h2d / 1e6, now_ms() - t0);
        } else
            printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
    }
    { /* kernels reading / writing pinned host memory directly */
        unsigned long long *dsum = nullptr;
        CK(hipMalloc((void **)&dsum, 8));
        CK(hipMemset(dsum, 0, 8));
        void *dev_view = nullptr;
        CK(hipHostGetDevicePointer(&dev_view, hp, 0));
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now_ms();
            hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, s, (const uint4 *)dev_view, bytes / 16, dsum);
            CK(hipStreamSynchronize(s));
            const double k = now_ms() - t0;
            if (rep == 1) printf("kernel reading %zu MB of pinned host memory: %.3f ms = %.1f GB/s\n", mb, k, bytes / k / 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now_ms();
            hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, s, (uint4 *)dev_view, bytes / 16);
            CK(hipStreamSynchronize(s));
            const double k = now_ms() - t0;
            if (rep == 1) printf("kernel writing %zu MB of pinned host memory: %.3f ms = %.1f GB/s\n", mb, k, bytes / k / 1e6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now_ms();
            hipLaunchKernelGGL(sum_kernel, dim3(2048), dim3(256), 0, s, (const uint4 *)d, bytes / 16, dsum);
            CK(hipStreamSynchronize(s));
            const double k = now_ms() - t0;
            if (rep == 1) printf("kernel reading %zu MB of HBM: %.3f ms = %.1f GB/s\n", mb, k, bytes / k / 1e6);
        }
    }
    { /* the price of a tiny synchronous round trip (a 4-byte D2H + sync) */
        double best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            t0 = now_ms();
            CK(hipMemcpyAsync(hp2, d, 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            const double k = now_ms() - t0;
            best = k < best ? k : best;
        }
        printf("4-byte D2H + stream synchronise: %.1f us\n", best * 1e3);
    }
    printf("host threads used: %d (hardware_concurrency %u)\n", nthreads, std::thread::hardware_concurrency());
    return 0;
}
