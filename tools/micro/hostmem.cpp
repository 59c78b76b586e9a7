// hostmem.cpp -- what page-locked host memory costs and buys on the MI355X box (decides the ingest design):
// hipHostMalloc / hipHostRegister / hipMalloc cost by size, H2D rate from pageable, registered and allocated-pinned memory,
// one copy and several copies in flight.  Build: hipcc -O2 -o hostmem hostmem.cpp ; run: ./hostmem
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void touch(char *p, size_t n, int threads)
{
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=]() { const size_t a = n / threads * t, b = (t + 1 == threads) ? n : n / threads * (t + 1); memset(p + a, 1, b - a); });
    for (auto &th : pool) th.join();
}

int main()
{
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    const size_t sizes[] = {32u << 20, 256u << 20, 1024u << 20};
    char *dev = nullptr;
    double t0 = now();
    CK(hipMalloc((void **)&dev, 1024u << 20));
    printf("hipMalloc 1 GiB: %.3f ms\n", (now() - t0) * 1e3);
    for (size_t n : sizes) {
        // (1) hipHostMalloc
        char *pin = nullptr;
        t0 = now();
        CK(hipHostMalloc((void **)&pin, n, hipHostMallocDefault));
        const double t_alloc = now() - t0;
        t0 = now();
        touch(pin, n, 16);
        const double t_touch = now() - t0;
        // H2D from it
        CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t_h2d = (now() - t0) / 4;
        t0 = now();
        CK(hipHostFree(pin));
        const double t_free = now() - t0;
        printf("%5zu MiB  hipHostMalloc %.2f ms (%.1f GB/s), first touch 16 thr %.2f ms, H2D %.2f ms = %.1f GB/s, hipHostFree %.2f ms\n", n >> 20,
               t_alloc * 1e3, n / t_alloc / 1e9, t_touch * 1e3, t_h2d * 1e3, n / t_h2d / 1e9, t_free * 1e3);
        // (2) pageable + hipHostRegister
        char *pg = nullptr;
        if (posix_memalign((void **)&pg, 2u << 20, n) != 0) return 1;
        touch(pg, n, 16);
        CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int k = 0; k < 2; ++k) CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t_pg = (now() - t0) / 2;
        t0 = now();
        CK(hipHostRegister(pg, n, hipHostRegisterDefault));
        const double t_reg = now() - t0;
        CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double t_rh2d = (now() - t0) / 4;
        t0 = now();
        CK(hipHostUnregister(pg));
        const double t_unreg = now() - t0;
        printf("%5zu MiB  pageable H2D %.2f ms = %.1f GB/s; hipHostRegister %.2f ms (%.1f GB/s), H2D after %.2f ms = %.1f GB/s, unregister %.2f ms\n",
               n >> 20, t_pg * 1e3, n / t_pg / 1e9, t_reg * 1e3, n / t_reg / 1e9, t_rh2d * 1e3, n / t_rh2d / 1e9, t_unreg * 1e3);
        free(pg);
    }
    // (3) a ring of small pinned buffers, copies of 8 / 32 MiB back to back on two streams (what a staging ring would do)
    for (size_t piece : {size_t(8) << 20, size_t(32) << 20}) {
        const int NB = 4;
        char *ring[NB];
        t0 = now();
        for (int k = 0; k < NB; ++k) CK(hipHostMalloc((void **)&ring[k], piece, hipHostMallocDefault));
        const double t_ring = now() - t0;
        for (int k = 0; k < NB; ++k) touch(ring[k], piece, 8);
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const int reps = 32;
        t0 = now();
        for (int k = 0; k < reps; ++k) CK(hipMemcpyAsync(dev + (size_t)(k % 8) * piece, ring[k % NB], piece, hipMemcpyHostToDevice, (k & 1) ? s2 : st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(s2));
        const double t = now() - t0;
        printf("ring of %d x %zu MiB: alloc %.2f ms; %d copies on 2 streams: %.1f GB/s\n", NB, piece >> 20, t_ring * 1e3, reps, reps * piece / t / 1e9);
        // D2H too
        t0 = now();
        for (int k = 0; k < reps; ++k) CK(hipMemcpyAsync(ring[k % NB], dev + (size_t)(k % 8) * piece, piece, hipMemcpyDeviceToHost, (k & 1) ? s2 : st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(s2));
        printf("   D2H: %.1f GB/s\n", reps * piece / (now() - t0) / 1e9);
        for (int k = 0; k < NB; ++k) CK(hipHostFree(ring[k]));
        CK(hipStreamDestroy(s2));
    }
    // (4) small-allocation costs: hipMalloc of 1, 16, 256 MiB, and a hipMallocAsync pool
    for (size_t n : {size_t(1) << 20, size_t(16) << 20, size_t(256) << 20}) {
        char *d2 = nullptr;
        t0 = now();
        CK(hipMalloc((void **)&d2, n));
        const double ta = now() - t0;
        t0 = now();
        CK(hipFree(d2));
        printf("hipMalloc %zu MiB: %.3f ms, hipFree %.3f ms\n", n >> 20, ta * 1e3, (now() - t0) * 1e3);
    }
    return 0;
}
