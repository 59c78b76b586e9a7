// Pointer-chase latency probe: one lane (or 64 lanes, same address pattern per lane group) following a random cycle
// through arrays of different footprints.  hipcc --offload-arch=gfx950 -O3 chase.hip -o chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const uint32_t *next, uint32_t start, int steps, uint32_t *out, long long *cyc)
{
    uint32_t i = start + threadIdx.x % 1; // all lanes same chain
    long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = next[i];
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = i; cyc[blockIdx.x] = t1 - t0; }
}

int main()
{
    const size_t sizes[] = {256u << 10, 2u << 20, 16u << 20, 128u << 20, 1024u << 20};
    for (size_t bytes : sizes) {
        const size_t n = bytes / 64; // one hop per 64-byte line
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> next(bytes / 4, 0u);
        for (size_t k = 0; k < n; ++k) next[(size_t)perm[k] * 16] = perm[(k + 1) % n] * 16;
        uint32_t *d_next, *d_out; long long *d_cyc;
        hipMalloc(&d_next, bytes); hipMalloc(&d_out, 4096 * 4); hipMalloc(&d_cyc, 4096 * 8);
        hipMemcpy(d_next, next.data(), bytes, hipMemcpyHostToDevice);
        for (int blocks : {1, 256, 2048}) {
            const int steps = 2000;
            chase<<<blocks, 64>>>(d_next, perm[0] * 16, steps, d_out, d_cyc); // warm
            hipDeviceSynchronize();
            chase<<<blocks, 64>>>(d_next, perm[7] * 16, steps, d_out, d_cyc);
            hipDeviceSynchronize();
            std::vector<long long> cyc(blocks);
            hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
            long long mx = *std::max_element(cyc.begin(), cyc.end());
            printf("footprint %6zu KiB  blocks %5d  wall_clock ticks/hop (100MHz ticks): %.2f  => ns/hop %.1f\n", bytes >> 10, blocks,
                   (double)mx / steps, (double)mx / steps * 10.0);
        }
        hipFree(d_next); hipFree(d_out); hipFree(d_cyc);
    }
    return 0;
}
