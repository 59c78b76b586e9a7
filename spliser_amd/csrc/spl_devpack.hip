// spl_devpack.hip -- the packer of spl_pack.h on the device: BAM-native reads that are in device memory (the device decoder's
// arrays, spl_bam_decode_device; a caller's arrays brought up as they are, spl_soa_upload) -> the chunked, class-partitioned
// record layout the counting kernels read.  The host packer's classification (splrec::classify_ops) in fewer instructions (splrec::classify_lean: held to it on the host), same records.
//
// ONE launch per read set, one workgroup per chunk, every read fetched and classified ONCE (round 4: three launches per
// reference, every read fetched and classified twice, 5.4 ms per 200 M reads against 0.8 ms of counting):
//   * a thread takes FOUR consecutive reads: POS and the CIGAR offsets are one aligned 16-byte load each, the flags 8 bytes --
//     the chunks are cells of a grid over the arrays' indexes, so that this holds wherever a reference begins;
//   * the chunk's CIGAR ops are one contiguous stretch of the array: the workgroup stages it in LDS with coalesced 16-byte
//     loads and the classifier reads ops out of LDS (a stretch longer than the stage -- long-read CIGARs -- is read from
//     memory beyond it);
//   * the four records stay in registers while the workgroup counts: a read's rank in its run comes from a prefix sum over the
//     lanes' per-run counts, the waves' totals through LDS -- the runs keep file order;
//   * the chunk writes into a record slot of its own (worst-case size), so no workgroup waits for another's count; WIDE reads'
//     ops are not copied at all: their index points into the array they came from, which the read set keeps alive;
//   * the chunk's descriptor (spl_chunk_meta) is written by the kernel: nothing comes down to the host.
// The range kernel's chunk order (XCD share by XCD share, longest first) is made by spl_chunk_order_kernel BEFORE the layout, from
// cost estimates that need the chunks' numbers of reads and ops only (spl_layout_map_kernel): layout -> range with nothing between.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "spl_pack.h"
#include "spl_devpack.h"
#include "spl_layout_tile.h"

using namespace spllay;

// A thread per chunk of the launch: which segment it is of (a bisection over the segments' first chunks), where its reads and
// its ops lie, and what it will cost the range kernel, roughly -- from its numbers of reads and ops alone (a simple read is one op
// and weighs 2, a once-spliced one three and 5, a twice-spliced one five and 9: (3 ops + reads) / 2), so that the chunk order can be
// made BEFORE the records are (spl_chunk_order_kernel) and nothing stands between the layout kernel and the range kernel.
__global__ __launch_bounds__(256) void spl_layout_map_kernel(const spl_devreads src, const spl_layout_seg *segs, uint32_t n_segs, uint32_t n_chunks, uint32_t chunk,
                                                             spl_layout_chunk *chunks, uint32_t *cost)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_chunks) return;
    uint32_t a = 0, b = n_segs; // the last segment whose first chunk is <= k
    while (b - a > 1u) {
        const uint32_t mid = (a + b) >> 1;
        if (segs[mid].dev0 <= k) a = mid; else b = mid;
    }
    const spl_layout_seg s = segs[a];
    const int64_t cell = s.first / chunk + (int64_t)(k - s.dev0), c0 = cell * (int64_t)chunk, c1 = c0 + chunk;
    spl_layout_chunk c;
    c.lo = s.first > c0 ? s.first : c0;
    const int64_t hi = s.first + s.n_reads < c1 ? s.first + s.n_reads : c1;
    c.n = (uint32_t)(hi - c.lo);
    c.o_lo = src.cig_off[c.lo]; c.o_hi = src.cig_off[hi]; c.seg_op0 = src.cig_off[s.first];
    c.shift = s.shift;
    c.flat = s.chunk0 + (k - s.dev0);
    chunks[k] = c;
    cost[c.flat] = (3u * (c.o_hi - c.o_lo) + c.n) / 2u;
}

template <int C>
__global__ __launch_bounds__(C / 4) __attribute__((amdgpu_waves_per_eu(8, 8))) void spl_layout_kernel(const spl_layout_params p)
{
    constexpr uint32_t T = C / 4, NW = T / 64, STAGE = 4 * C;
    __shared__ uint32_t s_ops[STAGE];
    __shared__ uint32_t s_cnt[NW][2];
    __shared__ int32_t s_first; // POS of the chunk's first read in file order: base of the range kernel's LDS window
    const uint32_t k = blockIdx.x;
    const spl_layout_chunk ch = p.chunks[k]; // (wave-uniform: one scalar load)
    uint32_t n[4];
    layout_tile<C>(p.src, p.n_rec, p.n_ops, ch, (lay_lds_w32 *)s_ops, (lay_lds_w32 *)&s_cnt[0][0], (lay_lds_i32 *)&s_first, RecordsInMemory{p.rec_base, SPL_LAYOUT_SLOT(C)}, n);
    if (threadIdx.x == 0) {
        uint8_t *const rec = p.rec_base + (size_t)k * SPL_LAYOUT_SLOT(C);
        spl_chunk_meta m;
        m.rec = (uint64_t)(uintptr_t)rec;
        m.wide = (uint64_t)(uintptr_t)(p.src.cigar + ch.seg_op0);
        m.shift = ch.shift;
        m.first_pos = s_first;
        m.n[0] = (uint16_t)n[0]; m.n[1] = (uint16_t)n[1]; m.n[2] = (uint16_t)n[2]; m.n[3] = (uint16_t)n[3];
        p.meta[ch.flat] = m;
    }
}

// The range kernel's workgroup b works on slot (b & 7) * per + (b >> 3): XCD x -- the hardware deals workgroups round-robin over
// the eight -- walks slots [x * per, (x + 1) * per).  WHICH chunks an XCD gets decides when it is done, and the launch ends with
// the slowest: the chunks are dealt to the XCDs in blocks of 8 consecutive chunks, round-robin (every stretch of the genome is
// spread over all eight; neighbouring chunks, which share lines of the position index, still go to one L2 together), and inside
// an XCD's share they go longest first by the cost estimate (a counting sort on cost / 16; equal keys in any order).  One
// workgroup per XCD share; slots past a share's chunks hold 0xffffffff.
__global__ __launch_bounds__(1024) void spl_chunk_order_kernel(const uint32_t *cost, uint32_t n, uint32_t per, uint32_t *order)
{
    constexpr uint32_t NB = 4096;
    __shared__ uint32_t s_hist[NB];
    __shared__ uint32_t s_part[1024];
    const uint32_t x = blockIdx.x, t = threadIdx.x;
    for (uint32_t b = t; b < NB; b += 1024u) s_hist[b] = 0;
    __syncthreads();
    // member e of share x: chunk (x + 8 * (e / 8)) * 8 + e % 8
    auto chunk_of = [&](uint32_t e) { return (x + 8u * (e >> 3)) * 8u + (e & 7u); };
    auto key_of = [&](uint32_t c) { const uint32_t q = c >> 4; return NB - 1u - (q < NB ? q : NB - 1u); };
    for (uint32_t e = t; e < per; e += 1024u) {
        const uint32_t j = chunk_of(e);
        if (j < n) atomicAdd(&s_hist[key_of(cost[j])], 1u);
    }
    __syncthreads();
    // exclusive prefix sums over the keys: four per thread, the threads' sums through LDS
    const uint32_t h0 = s_hist[4 * t], h1 = s_hist[4 * t + 1], h2 = s_hist[4 * t + 2], h3 = s_hist[4 * t + 3];
    s_part[t] = h0 + h1 + h2 + h3;
    __syncthreads();
    for (uint32_t s = 1; s < 1024u; s <<= 1) {
        const uint32_t a = t >= s ? s_part[t - s] : 0u;
        __syncthreads();
        s_part[t] += a;
        __syncthreads();
    }
    const uint32_t before = s_part[t] - (h0 + h1 + h2 + h3), total = s_part[1023];
    __syncthreads();
    s_hist[4 * t] = before; s_hist[4 * t + 1] = before + h0; s_hist[4 * t + 2] = before + h0 + h1; s_hist[4 * t + 3] = before + h0 + h1 + h2;
    __syncthreads();
    uint32_t *const mine = order + (size_t)x * per;
    for (uint32_t e = t; e < per; e += 1024u) {
        const uint32_t j = chunk_of(e);
        if (j < n) mine[atomicAdd(&s_hist[key_of(cost[j])], 1u)] = j;
    }
    for (uint32_t s = total + t; s < per; s += 1024u) mine[s] = 0xffffffffu;
}

extern "C" int spl_dev_launch_layout_map(const spl_devreads *src, const spl_layout_seg *segs, uint32_t n_segs, uint32_t n_chunks, uint32_t chunk, spl_layout_chunk *chunks,
                                         uint32_t *cost, void *stream)
{
    if (!n_segs || !n_chunks) return 0;
    hipLaunchKernelGGL(spl_layout_map_kernel, dim3((n_chunks + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, *src, segs, n_segs, n_chunks, chunk, chunks, cost);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_layout(const spl_layout_params *p, uint32_t n_dev_chunks, uint32_t chunk, void *stream, void *ev_start, void *ev_stop)
{
    if (!n_dev_chunks) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
    if (chunk == (uint32_t)SPL_CHUNK) hipExtLaunchKernelGGL(spl_layout_kernel<SPL_CHUNK>, dim3(n_dev_chunks), dim3(SPL_CHUNK / 4), 0, st, e0, e1, 0, *p);
    else if (chunk == (uint32_t)SPL_CHUNK_BIG) hipExtLaunchKernelGGL(spl_layout_kernel<SPL_CHUNK_BIG>, dim3(n_dev_chunks), dim3(SPL_CHUNK_BIG / 4), 0, st, e0, e1, 0, *p);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_chunk_order(const uint32_t *cost, uint32_t n_chunks, uint32_t chunk, uint32_t *order, void *stream)
{
    if (!n_chunks) return 0;
    (void)chunk; // (the map kernel's estimate, (3 ops + reads) / 2, has no bound of its own for long-read chunks: the order kernel's key_of clamps it to its 4096 keys)
    hipLaunchKernelGGL(spl_chunk_order_kernel, dim3(8), dim3(1024), 0, (hipStream_t)stream, cost, n_chunks, spl_order_per(n_chunks), order);
    return (int)hipGetLastError();
}
