// spl_devpack.hip -- the packer of spl_pack.h on the device: BAM-native reads that are in device memory already (the device
// decoder's arrays, spl_bam_decode_device) -> the chunked, class-partitioned record layout the counting kernels read, without
// the round trip over the host.  Same classification code as the host packer (splrec::classify), same layout, same bytes.
//
// Three launches per segment, nothing between them on the host: SIZES (one workgroup per chunk: reads per run, cost, wide ops),
// OFFSETS (one workgroup: the sizes become places, exactly as splpack::plan makes them), RECORDS (one workgroup per chunk).  A
// thread takes every 256th read of its chunk, so that a wave's loads of POS, FLAG and CIGAR offsets are 64 neighbours (round 2
// gave a thread 8 or 16 consecutive reads: 64 lanes, 64 lines); a read's rank inside its run comes from ballots over the wave,
// the waves' counts through LDS, the rounds' counts carried along -- the runs keep file order, each read is classified once per
// launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spl_pack.h"
#include "spl_devpack.h"

namespace {

__device__ __forceinline__ void read_at(const spl_devreads &src, int64_t i, int32_t &pos, uint32_t &flag, const uint32_t *&ops, uint32_t &n_ops)
{
    pos = src.pos[i];
    flag = src.flag[i];
    const uint32_t o0 = src.cig_off[i];
    ops = src.cigar + o0;
    n_ops = src.cig_off[i + 1] - o0;
}

} // namespace

template <int ROUNDS>
__global__ __launch_bounds__(256) void spl_devpack_count_kernel(const spl_devreads src, int64_t first, int64_t n_reads, splpack::ChunkDesc *descs)
{
    __shared__ uint32_t s_n[SPL_RC_RUNS], s_cost, s_wide;
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    if (t < SPL_RC_RUNS) s_n[t] = 0;
    if (t == 0) { s_cost = 0; s_wide = 0; }
    __syncthreads();
    const int64_t base = (int64_t)c * (256 * ROUNDS);
    uint32_t n[SPL_RC_RUNS] = {0, 0, 0, 0}, cost = 0, wide = 0;
    splrec::Rec r;
#pragma unroll 2
    for (int k = 0; k < ROUNDS; ++k) {
        const int64_t i = base + (int64_t)k * 256 + t;
        if (i >= n_reads) break;
        int32_t pos; uint32_t flag, n_ops; const uint32_t *ops;
        read_at(src, first + i, pos, flag, ops, n_ops);
        splrec::classify(pos, flag, ops, n_ops, 0u, r);
        n[r.run]++;
        cost += r.weight;
        wide += r.n_wide;
    }
    // the workgroup's sums: inside the wave by shuffles, one atomic per wave and number
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int q = 0; q < SPL_RC_RUNS; ++q) n[q] += (uint32_t)__shfl_down((int)n[q], s);
        cost += (uint32_t)__shfl_down((int)cost, s);
        wide += (uint32_t)__shfl_down((int)wide, s);
    }
    if ((t & 63u) == 0u) {
#pragma unroll
        for (int q = 0; q < SPL_RC_RUNS; ++q) if (n[q]) atomicAdd(&s_n[q], n[q]);
        if (cost) atomicAdd(&s_cost, cost);
        if (wide) atomicAdd(&s_wide, wide);
    }
    __syncthreads();
    if (t == 0) {
        splpack::ChunkDesc d;
        for (int q = 0; q < SPL_RC_RUNS; ++q) d.n[q] = (uint16_t)s_n[q];
        d.cost = s_cost;
        d.wide_off = s_wide;                    // a count for now: spl_devpack_offsets_kernel makes places of them
        d.rec_off = spl_run_offset(d.n, 4);     // a size for now
        d.first_pos = src.pos[first + base];
        descs[c] = d;
    }
}

// sizes -> places (exclusive prefix sums over the chunks, as splpack::plan does), the totals to totals[0] (record bytes) and
// totals[1] (wide ops).  One workgroup: a segment has at most a few tens of thousands of chunks.
__global__ __launch_bounds__(1024) void spl_devpack_offsets_kernel(splpack::ChunkDesc *descs, uint32_t n_chunks, uint64_t *totals)
{
    __shared__ uint64_t s_rec[1024], s_wide[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n_chunks + 1023u) / 1024u, c0 = t * per, c1 = c0 + per < n_chunks ? c0 + per : n_chunks;
    uint64_t rec = 0, wide = 0;
    for (uint32_t c = c0; c < c1; ++c) { rec += descs[c].rec_off; wide += descs[c].wide_off; }
    s_rec[t] = rec; s_wide[t] = wide;
    __syncthreads();
    for (uint32_t s = 1; s < 1024u; s <<= 1) { // inclusive scan over the threads' sums
        const uint64_t a = t >= s ? s_rec[t - s] : 0, b = t >= s ? s_wide[t - s] : 0;
        __syncthreads();
        s_rec[t] += a; s_wide[t] += b;
        __syncthreads();
    }
    uint64_t r0 = s_rec[t] - rec, w0 = s_wide[t] - wide;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t bytes = descs[c].rec_off, ops = descs[c].wide_off;
        descs[c].rec_off = r0;
        descs[c].wide_off = w0;
        r0 += bytes;
        w0 += ops;
    }
    if (t == 1023u) { totals[0] = s_rec[t]; totals[1] = s_wide[t]; }
}

template <int ROUNDS>
__global__ __launch_bounds__(256) void spl_devpack_emit_kernel(const spl_devreads src, int64_t first, int64_t n_reads, const splpack::ChunkDesc *descs,
                                                               uint8_t *rec_base, uint32_t *wide_base)
{
    __shared__ uint32_t s_wave[4][SPL_RC_RUNS + 1];
    const uint32_t c = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const splpack::ChunkDesc d = descs[c];
    const int64_t base = (int64_t)c * (256 * ROUNDS);
    uint8_t *const chunk_rec = rec_base + d.rec_off;
    const uint32_t rec_size[SPL_RC_RUNS] = {SPL_REC_SIMPLE, SPL_REC_MNM, SPL_REC_M2, SPL_REC_OTHER};
    uint32_t run_off[SPL_RC_RUNS];
#pragma unroll
    for (int q = 0; q < SPL_RC_RUNS; ++q) run_off[q] = spl_run_offset(d.n, q);
    uint32_t done[SPL_RC_RUNS + 1] = {0, 0, 0, 0, 0}; // reads per run, then wide ops, of the rounds before this one
    const uint64_t below = (1ull << lane) - 1ull;
    splrec::Rec r;
    for (int k = 0; k < ROUNDS; ++k) {
        const int64_t i = base + (int64_t)k * 256 + t;
        const bool have = i < n_reads;
        if (base + (int64_t)k * 256 >= n_reads) break; // (the whole workgroup)
        int32_t pos = 0; uint32_t flag = 0, n_ops = 0; const uint32_t *ops = src.cigar;
        uint32_t run = SPL_RC_RUNS, n_wide = 0;
        if (have) {
            read_at(src, first + i, pos, flag, ops, n_ops);
            splrec::classify(pos, flag, ops, n_ops, 0u, r);
            run = r.run;
            n_wide = r.n_wide;
        }
        // the read's rank in its run among this round's reads: the lanes below it in the wave, the waves below it in the workgroup
        uint32_t rank = 0, in_wave[SPL_RC_RUNS];
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)SPL_RC_RUNS; ++q) {
            const uint64_t m = __ballot(run == q);
            in_wave[q] = (uint32_t)__popcll(m);
            if (run == q) rank = (uint32_t)__popcll(m & below);
        }
        uint32_t wide_rank = 0, wide_wave = 0;
        if (__ballot(n_wide != 0u) != 0ull) { // (rare: CIGARs of more than three reference-consuming ops)
            uint32_t v = n_wide;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)v, s);
                if (lane >= (uint32_t)s) v += up;
            }
            wide_rank = v - n_wide;
            wide_wave = (uint32_t)__shfl((int)v, 63);
        }
        if (lane == 0u) {
#pragma unroll
            for (int q = 0; q < SPL_RC_RUNS; ++q) s_wave[wave][q] = in_wave[q];
            s_wave[wave][SPL_RC_RUNS] = wide_wave;
        }
        __syncthreads();
        uint32_t before_run = 0, before_wide = 0, round_n[SPL_RC_RUNS + 1];
#pragma unroll
        for (int q = 0; q <= SPL_RC_RUNS; ++q) {
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                const uint32_t x = s_wave[w][q];
                if (w < wave) { if ((uint32_t)q == run) before_run += x; if (q == SPL_RC_RUNS) before_wide += x; }
                sum += x;
            }
            round_n[q] = sum;
        }
        if (have) {
            const uint32_t size = run == SPL_RC_SIMPLE ? SPL_REC_SIMPLE : (run == SPL_RC_MNM ? SPL_REC_MNM : SPL_REC_OTHER); // (M2 and OTHER: 24)
            const uint32_t at_run = run == 0u ? done[0] : (run == 1u ? done[1] : (run == 2u ? done[2] : done[3]));
            const uint32_t off_run = run == 0u ? run_off[0] : (run == 1u ? run_off[1] : (run == 2u ? run_off[2] : run_off[3]));
            uint32_t *dst = (uint32_t *)(chunk_rec + off_run + (size_t)(at_run + before_run + rank) * size);
            const uint64_t wide_at = d.wide_off + done[SPL_RC_RUNS] + before_wide + wide_rank;
            if (n_wide) r.w[4] = (uint32_t)wide_at; // (the index of the read's first wide op: classify puts it in the third op's place)
            dst[0] = r.w[0]; dst[1] = r.w[1];
            if (size >= 16u) { dst[2] = r.w[2]; dst[3] = r.w[3]; }
            if (size >= 24u) { dst[4] = r.w[4]; dst[5] = r.w[5]; }
            for (uint32_t w = 0; w < n_wide; ++w) wide_base[wide_at + w] = ops[w];
        }
#pragma unroll
        for (int q = 0; q <= SPL_RC_RUNS; ++q) done[q] += round_n[q];
        __syncthreads();
    }
    (void)rec_size;
}

extern "C" int spl_dev_launch_pack_count(const spl_devreads *src, int64_t first, int64_t n_reads, uint32_t chunk, void *descs, void *stream)
{
    const uint32_t n_chunks = (uint32_t)((n_reads + chunk - 1) / chunk);
    if (!n_chunks) return 0;
    if (chunk == 256u * 8u) hipLaunchKernelGGL(spl_devpack_count_kernel<8>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, *src, first, n_reads, (splpack::ChunkDesc *)descs);
    else if (chunk == 256u * 16u) hipLaunchKernelGGL(spl_devpack_count_kernel<16>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, *src, first, n_reads, (splpack::ChunkDesc *)descs);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_pack_offsets(void *descs, uint32_t n_chunks, void *totals, void *stream)
{
    if (!n_chunks) return 0;
    hipLaunchKernelGGL(spl_devpack_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (splpack::ChunkDesc *)descs, n_chunks, (uint64_t *)totals);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_pack_emit(const spl_devreads *src, int64_t first, int64_t n_reads, uint32_t chunk, const void *descs, void *rec_base, void *wide_base, void *stream)
{
    const uint32_t n_chunks = (uint32_t)((n_reads + chunk - 1) / chunk);
    if (!n_chunks) return 0;
    if (chunk == 256u * 8u)
        hipLaunchKernelGGL(spl_devpack_emit_kernel<8>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, *src, first, n_reads, (const splpack::ChunkDesc *)descs, (uint8_t *)rec_base, (uint32_t *)wide_base);
    else if (chunk == 256u * 16u)
        hipLaunchKernelGGL(spl_devpack_emit_kernel<16>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, *src, first, n_reads, (const splpack::ChunkDesc *)descs, (uint8_t *)rec_base, (uint32_t *)wide_base);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
