// spl_devpack.hip -- the packer of spl_pack.h on the device: BAM-native reads that are in device memory (the device decoder's
// arrays, spl_bam_decode_device; a caller's arrays brought up as they are, spl_soa_upload) -> the chunked, class-partitioned
// record layout the counting kernels read.  Same classifier as the host packer (splrec::classify_ops), same records.
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
//   * the chunk's descriptor (spl_chunk_meta) and its cost estimate are written by the kernel: nothing comes down to the host.
// The range kernel's chunk order (XCD share by XCD share, longest first) is made from the costs by spl_chunk_order_kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "spl_pack.h"
#include "spl_devpack.h"

namespace {

typedef uint32_t lay_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lay_u32x4 __attribute__((ext_vector_type(4)));
typedef lay_u32x2 lay_u32x2_a4 __attribute__((aligned(4)));
typedef lay_u32x4 lay_u32x4_a8 __attribute__((aligned(8)));

// The k-th op of a read whose first op is word rel0 of the stage (or beyond it: then from memory).  The two places are typed by
// address space: a pointer that could be either makes every read a flat load.
typedef __attribute__((address_space(3))) const uint32_t lay_lds_u32;
typedef __attribute__((address_space(1))) const uint32_t lay_glob_u32;
template <uint32_t STAGE>
struct StagedOps {
    lay_lds_u32 *lds;
    lay_glob_u32 *glob; // the read's first op in the array
    uint32_t rel0;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const
    {
        const uint32_t at = rel0 + k;
        if (at < STAGE) return lds[at];
        return glob[k];
    }
};

} // namespace

__global__ __launch_bounds__(256) void spl_layout_map_kernel(const spl_layout_seg *segs, uint32_t *chunk_seg)
{
    const spl_layout_seg s = segs[blockIdx.x];
    for (uint32_t j = threadIdx.x; j < s.n_chunks; j += 256u) chunk_seg[s.dev0 + j] = blockIdx.x;
}

template <int C>
__global__ __launch_bounds__(C / 4) __attribute__((amdgpu_waves_per_eu(8, 8))) void spl_layout_kernel(const spl_layout_params p)
{
    constexpr uint32_t T = C / 4, NW = T / 64, STAGE = 4 * C; // (threads, waves, words of the op stage: 4 ops a read on average)
    __shared__ uint32_t s_ops[STAGE];
    __shared__ uint32_t s_cnt[NW][2];
    __shared__ uint32_t s_cost;
    const uint32_t k = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t si = p.chunk_seg[k];
    const spl_layout_seg seg = p.segs[si];
    const int64_t cell = seg.first / C + (int64_t)(k - seg.dev0);
    const int64_t lo = seg.first > cell * C ? seg.first : cell * C;
    const int64_t seg_end = seg.first + seg.n_reads, cell_end = (cell + 1) * C;
    const int64_t hi = seg_end < cell_end ? seg_end : cell_end;
    const int64_t g = cell * C + 4 * (int64_t)t; // the thread's first read
    if (t == 0) s_cost = 0;

    // ---- trip 1: the thread's four reads (one load per array) and, wave-uniform, where the chunk's ops lie
    int32_t pos[4] = {0, 0, 0, 0};
    uint32_t flag[4] = {0, 0, 0, 0}, co[5] = {0, 0, 0, 0, 0};
    const bool mine = g + 4 > lo && g < hi; // (a partial cell's threads outside the segment load nothing)
    if (mine) {
        if (g + 4 <= p.n_rec) {
            const lay_u32x4 pv = *(const lay_u32x4 *)(p.src.pos + g);
            const lay_u32x2 fv = *(const lay_u32x2 *)(p.src.flag + g);
            const lay_u32x4 cv = *(const lay_u32x4 *)(p.src.cig_off + g);
            co[4] = p.src.cig_off[g + 4];
            pos[0] = (int32_t)pv.x; pos[1] = (int32_t)pv.y; pos[2] = (int32_t)pv.z; pos[3] = (int32_t)pv.w;
            flag[0] = fv.x & 0xffffu; flag[1] = fv.x >> 16; flag[2] = fv.y & 0xffffu; flag[3] = fv.y >> 16;
            co[0] = cv.x; co[1] = cv.y; co[2] = cv.z; co[3] = cv.w;
        } else { // (the arrays' last reads: one by one)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = g + j;
                if (i < p.n_rec) { pos[j] = p.src.pos[i]; flag[j] = p.src.flag[i]; co[j] = p.src.cig_off[i]; co[j + 1] = p.src.cig_off[i + 1]; }
                else co[j + 1] = co[j];
            }
        }
    }
    const uint32_t o_lo = p.src.cig_off[lo], o_hi = p.src.cig_off[hi], seg_op0 = p.src.cig_off[seg.first];
    const int32_t first_pos = p.src.pos[lo];

    // ---- trip 2: the chunk's ops into LDS, 16 bytes a lane and load
    const uint32_t a0 = o_lo & ~3u;
    {
        const uint64_t o_end = (uint64_t)o_hi < (uint64_t)a0 + STAGE ? (uint64_t)o_hi : (uint64_t)a0 + STAGE;
        lay_u32x4 v[STAGE / (4 * T)];
#pragma unroll
        for (uint32_t q = 0; q < STAGE / (4 * T); ++q) {
            const uint64_t w = (uint64_t)a0 + 4ull * (t + q * T);
            v[q] = lay_u32x4{0u, 0u, 0u, 0u};
            if (w < o_end) {
                if (w + 4 <= (uint64_t)p.n_ops) v[q] = *(const lay_u32x4 *)(p.src.cigar + w);
                else {
                    v[q].x = p.src.cigar[w];
                    if (w + 1 < (uint64_t)p.n_ops) v[q].y = p.src.cigar[w + 1];
                    if (w + 2 < (uint64_t)p.n_ops) v[q].z = p.src.cigar[w + 2];
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < STAGE / (4 * T); ++q) *(lay_u32x4 *)(s_ops + 4u * (t + q * T)) = v[q];
    }
    __syncthreads();

    // ---- classify: four records in registers
    uint32_t w[4][6], run[4];
    uint32_t c01 = 0, c23 = 0, cost = 0; // reads per run: run 0 | run 1 << 16, run 2 | run 3 << 16
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = g + j;
        run[j] = SPL_RC_RUNS;
        if (i >= lo && i < hi) {
            splrec::Rec r;
            const StagedOps<STAGE> ops{(lay_lds_u32 *)s_ops, (lay_glob_u32 *)(p.src.cigar + co[j]), co[j] - a0};
            splrec::classify_ops(pos[j], flag[j], ops, co[j + 1] - co[j], co[j] - seg_op0, r);
            run[j] = r.run;
            cost += r.weight;
#pragma unroll
            for (int q = 0; q < 6; ++q) w[j][q] = r.w[q];
            c01 += r.run == SPL_RC_SIMPLE ? 1u : (r.run == SPL_RC_MNM ? 0x10000u : 0u);
            c23 += r.run == SPL_RC_M2 ? 1u : (r.run == SPL_RC_OTHER ? 0x10000u : 0u);
        }
    }

    // ---- ranks: prefix sums over the lanes, the waves' totals through LDS
    uint32_t i01 = c01, i23 = c23;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)i01, s), b = (uint32_t)__shfl_up((int)i23, s);
        if (lane >= (uint32_t)s) { i01 += a; i23 += b; }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) cost += (uint32_t)__shfl_down((int)cost, s);
    if (lane == 63u) { s_cnt[wave][0] = i01; s_cnt[wave][1] = i23; }
    if (lane == 0u && cost) atomicAdd(&s_cost, cost);
    __syncthreads();
    uint32_t b01 = 0, b23 = 0, t01 = 0, t23 = 0;
#pragma unroll
    for (uint32_t q = 0; q < NW; ++q) {
        const uint32_t x = s_cnt[q][0], y = s_cnt[q][1];
        if (q < wave) { b01 += x; b23 += y; }
        t01 += x; t23 += y;
    }
    const uint32_t n0 = t01 & 0xffffu, n1 = t01 >> 16, n2 = t23 & 0xffffu, n3 = t23 >> 16;
    const uint32_t off1 = (n0 * SPL_REC_SIMPLE + 15u) & ~15u, off2 = off1 + n1 * SPL_REC_MNM, off3 = off2 + n2 * SPL_REC_M2;
    // (reads of each run before this thread's: the waves below, the lanes below)
    const uint32_t e01 = b01 + i01 - c01, e23 = b23 + i23 - c23;
    uint32_t at[4] = {e01 & 0xffffu, e01 >> 16, e23 & 0xffffu, e23 >> 16};

    // ---- the records, each to its place in its run
    uint8_t *const rec = p.rec_base + (size_t)k * SPL_LAYOUT_SLOT(C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t r = run[j];
        if (r == SPL_RC_SIMPLE) {
            *(lay_u32x2 *)(rec + (size_t)at[0] * SPL_REC_SIMPLE) = lay_u32x2{w[j][0], w[j][1]};
            ++at[0];
        } else if (r == SPL_RC_MNM) {
            *(lay_u32x4 *)(rec + off1 + (size_t)at[1] * SPL_REC_MNM) = lay_u32x4{w[j][0], w[j][1], w[j][2], w[j][3]};
            ++at[1];
        } else if (r < (uint32_t)SPL_RC_RUNS) {
            const bool m2 = r == SPL_RC_M2;
            uint8_t *dst = rec + (m2 ? off2 + (size_t)at[2] * SPL_REC_M2 : off3 + (size_t)at[3] * SPL_REC_OTHER);
            *(lay_u32x4_a8 *)dst = lay_u32x4{w[j][0], w[j][1], w[j][2], w[j][3]};
            *(lay_u32x2 *)(dst + 16) = lay_u32x2{w[j][4], w[j][5]};
            if (m2) ++at[2]; else ++at[3];
        }
    }
    if (t == 0) {
        spl_chunk_meta m;
        m.rec = (uint64_t)(uintptr_t)rec;
        m.wide = (uint64_t)(uintptr_t)(p.src.cigar + seg_op0);
        m.shift = seg.shift;
        m.first_pos = first_pos;
        m.n[0] = (uint16_t)n0; m.n[1] = (uint16_t)n1; m.n[2] = (uint16_t)n2; m.n[3] = (uint16_t)n3;
        const uint32_t flat = seg.chunk0 + (k - seg.dev0);
        p.meta[flat] = m;
        p.cost[flat] = s_cost;
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

extern "C" int spl_dev_launch_layout_map(const spl_layout_seg *segs, uint32_t n_segs, uint32_t *chunk_seg, void *stream)
{
    if (!n_segs) return 0;
    hipLaunchKernelGGL(spl_layout_map_kernel, dim3(n_segs), dim3(256), 0, (hipStream_t)stream, segs, chunk_seg);
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
    (void)chunk; // (costs are at most chunk * SPL_W_WIDE = 57 344: cost / 16 is below the 4096 keys for both chunk sizes)
    hipLaunchKernelGGL(spl_chunk_order_kernel, dim3(8), dim3(1024), 0, (hipStream_t)stream, cost, n_chunks, spl_order_per(n_chunks), order);
    return (int)hipGetLastError();
}
