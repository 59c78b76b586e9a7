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

namespace {

typedef uint32_t lay_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lay_u32x4 __attribute__((ext_vector_type(4)));
typedef lay_u32x2 lay_u32x2_a4 __attribute__((aligned(4)));
typedef lay_u32x4 lay_u32x4_a8 __attribute__((aligned(8)));

// A read's ops out of the workgroup's stage in LDS: its first SPL_PACK_SCAN_OPS words lie inside it whatever the number of its ops
// (what a short CIGAR's accessor gives back beyond them is the neighbour's, and ignored).  Typed by address space: through a
// generic pointer every read would be a flat load.
typedef __attribute__((address_space(3))) const uint32_t lay_lds_u32;
struct StagedOps {
    static constexpr bool padded = true;
    lay_lds_u32 *p;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return p[k]; }
};
// Inclusive prefix sum inside every row of 16 lanes.
__device__ __forceinline__ uint32_t row_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

// Inclusive prefix sum over the 64 lanes in six DPP adds: shifts by 1, 2, 4, 8 inside the rows of 16, then row 0's total into row
// 1 and row 2's into row 3 (row_bcast:15), then the first half's into the second (row_bcast:31).
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

} // namespace

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
    constexpr uint32_t T = C / 4, NW = T / 64, STAGE = 4 * C; // (threads, waves, words of the op stage: 4 ops a read on average)
    constexpr uint32_t PAD = SPL_PACK_SCAN_OPS;               // a read is classified from a stage that holds its first eight ops
    __shared__ uint32_t s_ops[STAGE];
    __shared__ uint32_t s_cnt[NW][2];
    __shared__ int32_t s_first; // POS of the chunk's first read in file order: base of the range kernel's LDS window
    const uint32_t k = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const spl_layout_chunk ch = p.chunks[k]; // (wave-uniform: one scalar load)
    const int64_t lo = ch.lo, hi = ch.lo + ch.n;
    const int64_t g = (lo & ~(int64_t)(C - 1)) + 4 * (int64_t)t; // the thread's first read
    const uint32_t o_lo = ch.o_lo, o_hi = ch.o_hi, seg_op0 = ch.seg_op0;

    // ---- the one memory trip: the thread's four reads (one load per array) and the chunk's ops, everything asked for at once
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
    // The chunk's ops into LDS, 16 bytes a lane and load, a WINDOW of STAGE words at a time: one window for all but long-read
    // CIGARs (more than four ops a read on average), whose chunks take several -- a read is classified from the window that
    // holds its first eight ops (more are never looked at: such a read is WIDE, its ops stay where they are).
    auto fill = [&](uint64_t ws) {
        const uint64_t o_end = (uint64_t)o_hi < ws + STAGE ? (uint64_t)o_hi : ws + STAGE;
        lay_u32x4 v[STAGE / (4 * T)];
#pragma unroll
        for (uint32_t q = 0; q < STAGE / (4 * T); ++q) {
            const uint64_t at = ws + 4ull * (t + q * T);
            v[q] = lay_u32x4{0u, 0u, 0u, 0u};
            if (at < o_end) {
                if (at + 4 <= (uint64_t)p.n_ops) v[q] = *(const lay_u32x4 *)(p.src.cigar + at);
                else {
                    v[q].x = p.src.cigar[at];
                    if (at + 1 < (uint64_t)p.n_ops) v[q].y = p.src.cigar[at + 1];
                    if (at + 2 < (uint64_t)p.n_ops) v[q].z = p.src.cigar[at + 2];
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < STAGE / (4 * T); ++q) *(lay_u32x4 *)(s_ops + 4u * (t + q * T)) = v[q];
    };
    uint64_t ws = o_lo & ~3u;
    fill(ws);
    __syncthreads();

    // ---- classify: four records in registers.  Straight-line for the reads of at most five ops that all consume the reference
    // (classify_fast5: no branch, the four reads' chains side by side); the others -- clips, insertions, long CIGARs, ops beyond
    // the first window -- are left pending and done by the lanes that hold them, through the general classifier.
    uint32_t w[4][6];
    uint32_t runs = 0, pend = 0; // run of read j: bits 3j .. 3j + 2 (4 = no read); pending: bit j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = g + j;
        const bool valid = i >= lo && i < hi;
        const uint32_t rel0 = co[j] - (uint32_t)ws;
        const bool inside = rel0 + PAD <= STAGE;
        lay_lds_u32 *o = (lay_lds_u32 *)s_ops + (inside ? rel0 : 0u);
        splrec::Rec r;
        const bool fast = splrec::classify_fast5(pos[j], flag[j], o[0], o[1], o[2], o[3], o[4], co[j + 1] - co[j], co[j] - seg_op0, r);
        pend |= (valid && !(fast && inside) ? 1u : 0u) << j;
        runs |= (valid ? r.run : (uint32_t)SPL_RC_RUNS) << (3 * j);
#pragma unroll
        for (int q = 0; q < 6; ++q) w[j][q] = r.w[q];
        __builtin_amdgcn_sched_barrier(0); // (one read after the other: four chains side by side do not fit the 64 registers that keep two workgroups on a CU)
    }
    for (;;) {
        if (__any(pend != 0u)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((pend >> j) & 1u) {
                    // (the read's offsets come from memory again, its POS and flag out of the record's first words: nothing but the
                    //  records stays in registers across this branch, which most waves never take)
                    const uint32_t c0 = p.src.cig_off[g + j], c1 = p.src.cig_off[g + j + 1], rel0 = c0 - (uint32_t)ws;
                    if (rel0 + PAD <= STAGE) {
                        splrec::Rec r;
                        splrec::classify_lean((int32_t)w[j][0], w[j][1] & 0xffffu, StagedOps{(lay_lds_u32 *)s_ops + rel0}, c1 - c0, c0 - seg_op0, r);
                        runs = (runs & ~(7u << (3 * j))) | (r.run << (3 * j));
                        pend &= ~(1u << j);
#pragma unroll
                        for (int q = 0; q < 6; ++q) w[j][q] = r.w[q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ws + STAGE >= (uint64_t)o_hi + PAD) break; // (wave-uniform: the chunk's ops, and eight words behind them, were in this window)
        ws += STAGE - PAD;
        __syncthreads();
        fill(ws);
        __syncthreads();
    }
    uint32_t c01 = 0, c23 = 0; // reads per run: run 0 | run 1 << 16, run 2 | run 3 << 16
    uint32_t run[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        run[j] = (runs >> (3 * j)) & 7u;
        c01 += run[j] == SPL_RC_SIMPLE ? 1u : (run[j] == SPL_RC_MNM ? 0x10000u : 0u);
        c23 += run[j] == SPL_RC_M2 ? 1u : (run[j] == SPL_RC_OTHER ? 0x10000u : 0u);
    }

    // ---- ranks: prefix sums over the lanes (DPP: row shifts, then the rows' totals broadcast), the waves' totals through LDS
    const uint32_t i01 = wave_scan(c01), i23 = wave_scan(c23);
    if (lane == 63u) { s_cnt[wave][0] = i01; s_cnt[wave][1] = i23; }
    if (lo >= g && lo < g + 4) {
        const uint32_t e = (uint32_t)(lo - g);
        s_first = (int32_t)(e == 0u ? w[0][0] : (e == 1u ? w[1][0] : (e == 2u ? w[2][0] : w[3][0])));
    }
    __syncthreads();
    // the waves' totals: lane q of every wave takes wave q's, a prefix sum over those NW lanes (one row of 16: four DPP adds), and
    // the sums come out by lane number -- the wave's own number is uniform
    const uint32_t x = row_scan(lane < NW ? s_cnt[lane < NW ? lane : 0u][0] : 0u), y = row_scan(lane < NW ? s_cnt[lane < NW ? lane : 0u][1] : 0u);
    const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)x, NW - 1), t23 = (uint32_t)__builtin_amdgcn_readlane((int)y, NW - 1);
    const uint32_t b01 = wu ? (uint32_t)__builtin_amdgcn_readlane((int)x, wu - 1u) : 0u, b23 = wu ? (uint32_t)__builtin_amdgcn_readlane((int)y, wu - 1u) : 0u;
    const uint32_t n0 = t01 & 0xffffu, n1 = t01 >> 16, n2 = t23 & 0xffffu, n3 = t23 >> 16;
    const uint32_t off1 = (n0 * SPL_REC_SIMPLE + 15u) & ~15u, off2 = off1 + n1 * SPL_REC_MNM, off3 = off2 + n2 * SPL_REC_M2;
    // (reads of each run before this thread's: the waves below, the lanes below) -> byte offsets of the thread's next record of each run
    const uint32_t e01 = b01 + i01 - c01, e23 = b23 + i23 - c23;
    uint32_t at[4] = {(e01 & 0xffffu) * SPL_REC_SIMPLE, off1 + (e01 >> 16) * SPL_REC_MNM, off2 + (e23 & 0xffffu) * SPL_REC_M2, off3 + (e23 >> 16) * SPL_REC_OTHER};

    // ---- the records, each to its place in its run (the slot's address is uniform, the place a 32-bit offset)
    uint8_t *const rec = p.rec_base + (size_t)k * SPL_LAYOUT_SLOT(C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t r = run[j];
        const uint32_t d = r == SPL_RC_SIMPLE ? at[0] : (r == SPL_RC_MNM ? at[1] : (r == SPL_RC_M2 ? at[2] : at[3]));
        if (r == SPL_RC_SIMPLE) *(lay_u32x2 *)(rec + d) = lay_u32x2{w[j][0], w[j][1]};
        else if (r < (uint32_t)SPL_RC_RUNS) {
            *(lay_u32x4_a8 *)(rec + d) = lay_u32x4{w[j][0], w[j][1], w[j][2], w[j][3]};
            if (r != SPL_RC_MNM) *(lay_u32x2 *)(rec + d + 16) = lay_u32x2{w[j][4], w[j][5]};
        }
        at[0] += r == SPL_RC_SIMPLE ? SPL_REC_SIMPLE : 0u;
        at[1] += r == SPL_RC_MNM ? SPL_REC_MNM : 0u;
        at[2] += r == SPL_RC_M2 ? SPL_REC_M2 : 0u;
        at[3] += r == SPL_RC_OTHER ? SPL_REC_OTHER : 0u;
    }
    if (t == 0) {
        spl_chunk_meta m;
        m.rec = (uint64_t)(uintptr_t)rec;
        m.wide = (uint64_t)(uintptr_t)(p.src.cigar + seg_op0);
        m.shift = ch.shift;
        m.first_pos = s_first;
        m.n[0] = (uint16_t)n0; m.n[1] = (uint16_t)n1; m.n[2] = (uint16_t)n2; m.n[3] = (uint16_t)n3;
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
    (void)chunk; // (costs are at most chunk * SPL_W_WIDE = 57 344: cost / 16 is below the 4096 keys for both chunk sizes)
    hipLaunchKernelGGL(spl_chunk_order_kernel, dim3(8), dim3(1024), 0, (hipStream_t)stream, cost, n_chunks, spl_order_per(n_chunks), order);
    return (int)hipGetLastError();
}
