// filepin.cpp -- can a file's bytes go to the device straight from the page cache?  (run on the GPU box)
// The device decoder's readers pread 32 MB pieces into page-locked buffers (a CPU copy of the whole file, 3-4 ms a piece a thread)
// and the copy engine takes them from there.  The alternative measured here: mmap the file, page-lock a piece of the MAPPING
// (hipHostRegister), copy from it, unlock it -- no CPU copy.  Per piece: register, copy, unregister times, for 1, 2, 4, 6 threads.
// Build: hipcc -O2 -o filepin filepin.cpp -lpthread ; run: ./filepin <file> [piece MB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: filepin <file> [piece MB]\n"); return 2; }
    const size_t piece = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 20;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { perror(argv[1]); return 1; }
    const size_t fsize = (size_t)sb.st_size, n_pieces = std::min<size_t>(fsize / piece, 192);
    char *map = (char *)mmap(nullptr, fsize, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 1; }
    CK(hipSetDevice(0));
    char *dev = nullptr;
    CK(hipMalloc((void **)&dev, n_pieces * piece));
    printf("%s: %zu pieces of %zu MB\n", argv[1], n_pieces, piece >> 20);
    // (a) the product's way: pread into a registered buffer, copy
    for (int T : {3, 6}) {
        std::vector<char *> buf((size_t)T);
        std::vector<hipStream_t> st((size_t)T);
        for (int t = 0; t < T; ++t) {
            if (posix_memalign((void **)&buf[(size_t)t], 2u << 20, piece)) return 1;
            CK(hipHostRegister(buf[(size_t)t], piece, hipHostRegisterDefault));
            CK(hipStreamCreateWithFlags(&st[(size_t)t], hipStreamNonBlocking));
        }
        const double t0 = now();
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t]() {
                CK(hipSetDevice(0));
                for (size_t k = (size_t)t; k < n_pieces; k += (size_t)T) {
                    size_t have = 0;
                    while (have < piece) { const ssize_t g = pread(fd, buf[(size_t)t] + have, piece - have, (off_t)(k * piece + have)); if (g <= 0) break; have += (size_t)g; }
                    CK(hipMemcpyAsync(dev + k * piece, buf[(size_t)t], piece, hipMemcpyHostToDevice, st[0]));
                    CK(hipStreamSynchronize(st[0]));
                }
            });
        for (auto &th : pool) th.join();
        const double dt = now() - t0;
        printf("pread into pinned buffers, %d threads: %.3f s = %.1f GB/s\n", T, dt, n_pieces * piece / dt / 1e9);
        for (int t = 0; t < T; ++t) { CK(hipHostUnregister(buf[(size_t)t])); free(buf[(size_t)t]); }
    }
    // (b) page-lock the mapping piece by piece
    for (int T : {1, 2, 4, 6}) {
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        std::atomic<long> t_reg(0), t_copy(0), t_unreg(0), failed(0);
        const double t0 = now();
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t]() {
                CK(hipSetDevice(0));
                for (size_t k = (size_t)t; k < n_pieces; k += (size_t)T) {
                    const double a = now();
                    hipError_t e = hipHostRegister(map + k * piece, piece, hipHostRegisterDefault);
                    if (e != hipSuccess) { if (failed.fetch_add(1) == 0) fprintf(stderr, "hipHostRegister on the mapping: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return; }
                    const double b = now();
                    CK(hipMemcpyAsync(dev + k * piece, map + k * piece, piece, hipMemcpyHostToDevice, st));
                    CK(hipStreamSynchronize(st));
                    const double c = now();
                    CK(hipHostUnregister(map + k * piece));
                    const double d = now();
                    t_reg += (long)((b - a) * 1e6); t_copy += (long)((c - b) * 1e6); t_unreg += (long)((d - c) * 1e6);
                }
            });
        for (auto &th : pool) th.join();
        const double dt = now() - t0;
        if (failed.load()) { printf("page-locking the mapping does not work here\n"); break; }
        printf("page-locked mapping, %d threads: %.3f s = %.1f GB/s; per piece register %.2f ms, copy (+ wait) %.2f ms, unregister %.2f ms\n", T, dt, n_pieces * piece / dt / 1e9,
               t_reg / 1e3 / n_pieces, t_copy / 1e3 / n_pieces, t_unreg / 1e3 / n_pieces);
    }
    return 0;
}
