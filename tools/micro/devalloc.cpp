// devalloc.cpp -- what fresh device memory costs on the MI355X box, and whether any way of asking for it is cheaper: a new process's
// first `process` call takes 13 GB of it and is 0.3 s slower than the calls behind it.  hipMalloc by size,
// several threads asking at once, the virtual-memory calls (hipMemCreate / hipMemMap), hipExtMallocWithFlags, first touch by a
// memset behind each.  Build: hipcc -O2 -o devalloc devalloc.cpp -lpthread ; run: ./devalloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <time.h>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double first_touch(void *p, size_t n)
{
    const double t0 = now();
    CK(hipMemset(p, 0x5a, n));
    CK(hipDeviceSynchronize());
    return now() - t0;
}

int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    const size_t GB = (size_t)1 << 30;
    if (argc >= 3 && !strcmp(argv[1], "dirty")) { // leave `argv[2]` GiB of device memory written to and not freed: what a process that exits leaves the driver
        const int n = atoi(argv[2]);
        double t_alloc = 0;
        std::vector<void *> held;
        for (int k = 0; k < n; k += 4) {
            void *p = nullptr;
            const double t0 = now();
            if (hipMalloc(&p, 4 * GB) != hipSuccess) { printf("dirty: stopped at %d GiB\n", k); break; }
            t_alloc += now() - t0;
            CK(hipMemset(p, 0x77, 4 * GB));
            held.push_back(p);
        }
        CK(hipDeviceSynchronize());
        printf("dirty: %d GiB written, hipMalloc took %.1f ms of it; leaving without freeing\n", n, t_alloc * 1e3);
        if (argc >= 4 && !strcmp(argv[3], "free")) { // ... or freeing first: what does the process's end cost then?
            const double t0 = now();
            for (void *p : held) CK(hipFree(p));
            printf("dirty: hipFree of all of it %.1f ms\n", (now() - t0) * 1e3);
        }
        {
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            printf("leaving at %.6f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
        }
        fflush(stdout);
        _exit(0);
    }
    if (argc >= 2 && !strcmp(argv[1], "probe")) { // what a process that starts now pays for 16 GiB: one request behind the other, then four at once
        double t0 = now();
        std::vector<void *> ps;
        for (int k = 0; k < 4; ++k) {
            void *p = nullptr;
            const double a = now();
            CK(hipMalloc(&p, 4 * GB));
            const double t_alloc = now() - a, t_touch = first_touch(p, 4 * GB);
            printf("probe: hipMalloc 4 GiB #%d %.2f ms, first memset %.2f ms\n", k, t_alloc * 1e3, t_touch * 1e3);
            ps.push_back(p);
        }
        printf("probe: 16 GiB one behind the other: %.1f ms\n", (now() - t0) * 1e3);
        std::vector<void *> qs(4, nullptr);
        std::vector<double> took(4, 0.0);
        t0 = now();
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < 4; ++t)
                pool.emplace_back([&, t]() { CK(hipSetDevice(0)); const double a = now(); CK(hipMalloc(&qs[(size_t)t], 4 * GB)); took[(size_t)t] = now() - a; });
            for (auto &th : pool) th.join();
        }
        printf("probe: 16 GiB as four requests at once: wall %.1f ms (calls %.1f %.1f %.1f %.1f ms)\n", (now() - t0) * 1e3, took[0] * 1e3, took[1] * 1e3, took[2] * 1e3, took[3] * 1e3);
        t0 = now();
        for (void *p : qs) first_touch(p, 4 * GB);
        printf("probe: first memsets of those: %.1f ms\n", (now() - t0) * 1e3);
        fflush(stdout);
        _exit(0);
    }
    {   // warm the runtime (the first allocation of a process pays for more than its bytes)
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, 64u << 20));
        printf("first hipMalloc of the process (64 MiB): %.2f ms\n", (now() - t0) * 1e3);
        first_touch(p, 64u << 20);
        CK(hipFree(p));
    }
    for (size_t n : {GB / 4, GB, 2 * GB, 4 * GB}) {
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, n));
        const double t_alloc = now() - t0;
        const double t_touch = first_touch(p, n), t_again = first_touch(p, n);
        t0 = now();
        CK(hipFree(p));
        printf("hipMalloc %5.2f GiB: %7.2f ms (%.1f ms/GiB)   first memset %.2f ms, second %.2f ms, hipFree %.2f ms\n", (double)n / GB, t_alloc * 1e3,
               t_alloc * 1e3 / ((double)n / GB), t_touch * 1e3, t_again * 1e3, (now() - t0) * 1e3);
    }
    for (int threads : {2, 4, 8}) { // the same 8 GiB, asked for by several threads at once
        std::vector<void *> ps((size_t)threads, nullptr);
        std::vector<double> took((size_t)threads, 0.0);
        const size_t each = 8 * GB / (size_t)threads;
        const double t0 = now();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() { CK(hipSetDevice(0)); const double a = now(); CK(hipMalloc(&ps[(size_t)t], each)); took[(size_t)t] = now() - a; });
        for (auto &th : pool) th.join();
        const double wall = now() - t0;
        double longest = 0;
        for (double v : took) longest = v > longest ? v : longest;
        for (void *p : ps) CK(hipFree(p));
        printf("8 GiB as %d x %.1f GiB from %d threads: wall %.2f ms (%.1f ms/GiB), the longest call %.2f ms\n", threads, (double)each / GB, threads, wall * 1e3, wall * 1e3 / 8.0, longest * 1e3);
    }
    {   // one thread, 8 GiB in one piece (the reference for the above)
        void *p = nullptr;
        const double t0 = now();
        CK(hipMalloc(&p, 8 * GB));
        const double t = now() - t0;
        CK(hipFree(p));
        printf("8 GiB in one hipMalloc: %.2f ms (%.1f ms/GiB)\n", t * 1e3, t * 1e3 / 8.0);
    }
    {   // again, right behind a free of the same size (does the driver hand the pages back as they are?)
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, 4 * GB));
        const double a = now() - t0;
        CK(hipFree(p));
        t0 = now();
        CK(hipMalloc(&p, 4 * GB));
        const double b = now() - t0;
        CK(hipFree(p));
        printf("4 GiB, freed, 4 GiB again: %.2f ms then %.2f ms\n", a * 1e3, b * 1e3);
    }
    {   // hipExtMallocWithFlags
        struct { unsigned flag; const char *name; } kinds[] = {{hipDeviceMallocDefault, "default"}, {hipDeviceMallocUncached, "uncached"}, {hipDeviceMallocFinegrained, "fine-grained"}};
        for (auto &k : kinds) {
            void *p = nullptr;
            const double t0 = now();
            const hipError_t e = hipExtMallocWithFlags(&p, 4 * GB, k.flag);
            const double t = now() - t0;
            if (e != hipSuccess) { (void)hipGetLastError(); printf("hipExtMallocWithFlags(%s): %s\n", k.name, hipGetErrorString(e)); continue; }
            const double touch = first_touch(p, 4 * GB);
            CK(hipFree(p));
            printf("hipExtMallocWithFlags(%s) 4 GiB: %.2f ms (%.1f ms/GiB), first memset %.2f ms\n", k.name, t * 1e3, t * 1e3 / 4.0, touch * 1e3);
        }
    }
    {   // virtual memory management: reserve, create, map, set access
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("hipMemGetAllocationGranularity: %s\n", hipGetErrorString(e)); }
        else {
            printf("virtual memory: recommended granularity %zu bytes\n", gran);
            const size_t n = 4 * GB;
            void *va = nullptr;
            double t0 = now();
            CK(hipMemAddressReserve(&va, n, gran, nullptr, 0));
            const double t_res = now() - t0;
            hipMemGenericAllocationHandle_t h;
            t0 = now();
            e = hipMemCreate(&h, n, &prop, 0);
            const double t_create = now() - t0;
            if (e != hipSuccess) { (void)hipGetLastError(); printf("hipMemCreate: %s\n", hipGetErrorString(e)); }
            else {
                t0 = now();
                CK(hipMemMap(va, n, 0, h, 0));
                const double t_map = now() - t0;
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                t0 = now();
                CK(hipMemSetAccess(va, n, &acc, 1));
                const double t_acc = now() - t0;
                const double touch = first_touch(va, n);
                printf("4 GiB by hipMemCreate: reserve %.2f + create %.2f + map %.2f + access %.2f ms = %.1f ms/GiB, first memset %.2f ms\n", t_res * 1e3, t_create * 1e3, t_map * 1e3,
                       t_acc * 1e3, (t_res + t_create + t_map + t_acc) * 1e3 / 4.0, touch * 1e3);
                CK(hipMemUnmap(va, n));
                CK(hipMemRelease(h));
            }
            CK(hipMemAddressFree(va, n));
        }
    }
    {   // the stream-ordered pool
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMallocAsync(&p, 4 * GB, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        const double t = now() - t0;
        if (e != hipSuccess) { (void)hipGetLastError(); printf("hipMallocAsync: %s\n", hipGetErrorString(e)); }
        else {
            const double touch = first_touch(p, 4 * GB);
            CK(hipFreeAsync(p, st));
            CK(hipStreamSynchronize(st));
            printf("hipMallocAsync 4 GiB: %.2f ms (%.1f ms/GiB), first memset %.2f ms\n", t * 1e3, t * 1e3 / 4.0, touch * 1e3);
        }
    }
    return 0;
}
