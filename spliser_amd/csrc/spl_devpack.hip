// spl_devpack.hip -- the packer of spl_pack.h on the device: BAM-native reads that are in device memory already (the device
// decoder's arrays, spl_bam_decode_device) -> the chunked, class-partitioned record layout the counting kernels read, without
// the round trip over the host.  Same classification code as the host packer (splrec::classify), same layout, same bytes.
//
// Two launches per segment, one workgroup per chunk: count (reads per run, cost, wide ops of every chunk -- the host turns the
// sizes into offsets, exactly as splpack::plan does) and emit (every thread writes the records of its R consecutive reads behind
// those of the threads before it: a block-wide exclusive scan of the per-thread counts keeps the runs in file order).
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

template <int R>
__global__ __launch_bounds__(256) void spl_devpack_count_kernel(const spl_devreads src, int64_t first, int64_t n_reads, splpack::ChunkDesc *descs)
{
    __shared__ uint32_t s_n[SPL_RC_RUNS], s_cost, s_wide;
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    if (t < SPL_RC_RUNS) s_n[t] = 0;
    if (t == 0) { s_cost = 0; s_wide = 0; }
    __syncthreads();
    const int64_t i0 = (int64_t)c * (256 * R) + (int64_t)t * R;
    uint32_t n[SPL_RC_RUNS] = {0, 0, 0, 0}, cost = 0, wide = 0;
    splrec::Rec r;
    for (int k = 0; k < R; ++k) {
        const int64_t i = i0 + k;
        if (i >= n_reads) break;
        int32_t pos; uint32_t flag, n_ops; const uint32_t *ops;
        read_at(src, first + i, pos, flag, ops, n_ops);
        splrec::classify(pos, flag, ops, n_ops, 0u, r);
        n[r.run]++;
        cost += r.weight;
        wide += r.n_wide;
    }
#pragma unroll
    for (int k = 0; k < SPL_RC_RUNS; ++k) if (n[k]) atomicAdd(&s_n[k], n[k]);
    if (cost) atomicAdd(&s_cost, cost);
    if (wide) atomicAdd(&s_wide, wide);
    __syncthreads();
    if (t == 0) {
        splpack::ChunkDesc d;
        for (int k = 0; k < SPL_RC_RUNS; ++k) d.n[k] = (uint16_t)s_n[k];
        d.cost = s_cost;
        d.wide_off = s_wide;                    // a count for now: the host makes offsets of them (as splpack::plan does)
        d.rec_off = spl_run_offset(d.n, 4);     // a size for now
        d.first_pos = src.pos[first + (int64_t)c * (256 * R)];
        descs[c] = d;
    }
}

template <int R>
__global__ __launch_bounds__(256) void spl_devpack_emit_kernel(const spl_devreads src, int64_t first, int64_t n_reads, const splpack::ChunkDesc *descs,
                                                               uint8_t *rec_base, uint32_t *wide_base)
{
    __shared__ uint32_t s_wave[4][SPL_RC_RUNS + 1];
    const uint32_t c = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const splpack::ChunkDesc d = descs[c];
    const int64_t i0 = (int64_t)c * (256 * R) + (int64_t)t * R;
    uint32_t cnt[SPL_RC_RUNS + 1] = {0, 0, 0, 0, 0}; // reads per run, then wide ops, of this thread
    splrec::Rec r;
    for (int k = 0; k < R; ++k) {
        const int64_t i = i0 + k;
        if (i >= n_reads) break;
        int32_t pos; uint32_t flag, n_ops; const uint32_t *ops;
        read_at(src, first + i, pos, flag, ops, n_ops);
        splrec::classify(pos, flag, ops, n_ops, 0u, r);
        cnt[r.run]++;
        cnt[SPL_RC_RUNS] += r.n_wide;
    }
    // exclusive scan over the workgroup's threads, five values at a time: inside the wave by shuffles, across the waves through LDS
    uint32_t before[SPL_RC_RUNS + 1];
#pragma unroll
    for (int k = 0; k <= SPL_RC_RUNS; ++k) {
        uint32_t v = cnt[k];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)v, s);
            if (lane >= (uint32_t)s) v += up;
        }
        if (lane == 63u) s_wave[wave][k] = v;
        before[k] = v - cnt[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k <= SPL_RC_RUNS; ++k)
        for (uint32_t w = 0; w < wave; ++w) before[k] += s_wave[w][k];
    // the records
    uint8_t *const chunk_rec = rec_base + d.rec_off;
    uint8_t *run_at[SPL_RC_RUNS];
    const uint32_t rec_size[SPL_RC_RUNS] = {SPL_REC_SIMPLE, SPL_REC_MNM, SPL_REC_M2, SPL_REC_OTHER};
#pragma unroll
    for (int k = 0; k < SPL_RC_RUNS; ++k) run_at[k] = chunk_rec + spl_run_offset(d.n, k) + (size_t)before[k] * rec_size[k];
    uint64_t wide_at = d.wide_off + before[SPL_RC_RUNS];
    for (int k = 0; k < R; ++k) {
        const int64_t i = i0 + k;
        if (i >= n_reads) break;
        int32_t pos; uint32_t flag, n_ops; const uint32_t *ops;
        read_at(src, first + i, pos, flag, ops, n_ops);
        splrec::classify(pos, flag, ops, n_ops, (uint32_t)wide_at, r);
        uint32_t *dst = (uint32_t *)run_at[r.run];
        const uint32_t words = rec_size[r.run] / 4u;
        for (uint32_t w = 0; w < words; ++w) dst[w] = r.w[w];
        run_at[r.run] += rec_size[r.run];
        for (uint32_t w = 0; w < r.n_wide; ++w) wide_base[wide_at + w] = ops[w];
        wide_at += r.n_wide;
    }
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
