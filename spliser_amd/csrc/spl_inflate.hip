// spl_inflate.hip -- see spl_inflate.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "spl_inflate.h"
#include "spl_wave.h"
#include "spl_inflate_wave.h"
#include "spl_crc.h"
#include "spl_crc_wave.h"

// The Huffman decoding, one BGZF block per WAVE (the block's symbols as a stream of tokens), and the block's bytes made from
// that stream, one block per LANE (spl_inflate_wave.h has the method, and is what the host tests run through the wave emulator).
__global__ __launch_bounds__(64) void spl_inflate_decode_kernel(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status,
                                                                uint8_t *tokens_all, uint32_t *n_tok, uint32_t stride, uint32_t opts)
{
    __shared__ splz::Shared sh;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const spl_zblock zb = blocks[b];
    uint32_t n = 0;
    const uint32_t st = splz::decode_block(sh, image, zb, tokens_all + (size_t)b * stride, n, stride, opts);
    if (threadIdx.x == 0) { status[b] = st; n_tok[b] = st == SPL_Z_OK ? n : 0u; }
}

// The same kernel for files whose blocks deflate well (an aligner's BAM through samtools: 7-8 KB a block): FIVE waves a SIMD instead
// of four.  What holds the kernel at four is its 10 KB of LDS a wave and its 120 registers; a tile of such a block has fewer tokens
// (more of its output comes from matches), so 3 KB of token room a tile do where 5 are kept for blocks of literals, and the compiler
// is held to 96 registers (nine of them spilled, in the header's parsing).  The waves' waits are what the kernel has most of
// (41 % of its cycles): htslib-shaped human file 295 -> 276 ms of it a call, the call 0.379 -> 0.360 s; on the sequence-like file,
// whose tiles would be cut short, it is 13 % slower and not used (profiles/r06G_decode_five_waves.txt).
#ifndef SPLZ_DENSE_WAVES
#define SPLZ_DENSE_WAVES 5 // (6 with SPLZ_TOKCAP_SMALL=1792 was built too: 80 registers, 22 of them spilled -- profiles/r06G_decode_five_waves.txt)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SPLZ_DENSE_WAVES, SPLZ_DENSE_WAVES))) void spl_inflate_decode_dense_kernel(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks,
                                                                                                                     uint32_t *status, uint8_t *tokens_all, uint32_t *n_tok, uint32_t stride,
                                                                                                                     uint32_t opts)
{
    __shared__ uint32_t raw[(sizeof(splz::Shared) - (splz::TOKCAP - splz::TOKCAP_SMALL)) / 4u]; // (Shared without the end of its last member)
    splz::Shared &sh = *reinterpret_cast<splz::Shared *>(raw);
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const spl_zblock zb = blocks[b];
    uint32_t n = 0;
    const uint32_t st = splz::decode_block<splz::TOKCAP_SMALL>(sh, image, zb, tokens_all + (size_t)b * stride, n, stride, opts);
    if (threadIdx.x == 0) { status[b] = st; n_tok[b] = st == SPL_Z_OK ? n : 0u; }
}

__global__ __launch_bounds__(64) void spl_inflate_copy_kernel(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out_all, uint32_t *status, const uint8_t *tokens_all,
                                                              const uint32_t *n_tok, uint32_t stride)
{
    __shared__ uint32_t lds[64u * splz::COPY_LANE_BYTES / 4u];
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != SPL_Z_OK) return;
    uint8_t *const mine = (uint8_t *)lds + threadIdx.x * splz::COPY_LANE_BYTES;
    const uint32_t made = splz::copy_block(out_all + blocks[b].out, blocks[b].out_len, tokens_all + (size_t)b * stride, n_tok[b], mine, mine + splz::RING_BYTES);
    if (made != blocks[b].out_len) status[b] = SPL_Z_SHORT;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// CRC32 (IEEE, reflected) of every block's payload against the value in its trailer: one lane per block -- 768 waves for a window,
// which leaves the CUs to the decoding kernel beside it -- and the lane's block as S streams (spl_crc.h): S chains of table
// look-ups and S 16-byte loads in flight instead of one of each (the lane a word after the other waited 2 000 cycles a load:
// 64 lanes on 64 different lines and nothing else to do).  Tables in LDS: slicing by four, and x^(2^k) for putting the streams'
// registers together.  Blocks that already failed keep their status.
template <int S>
__global__ __launch_bounds__(64) void spl_crc32_kernel(const uint8_t *out_all, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status)
{
    __shared__ uint32_t table[4 * 256]; // table[k * 256 + b] = the CRC register after byte b and k zero bytes
    __shared__ uint32_t x2n[splcrc::N_X2N];
    for (uint32_t i = threadIdx.x; i < 256u; i += 64u) table[i] = splcrc::byte_entry(i);
    if (threadIdx.x < (uint32_t)splcrc::N_X2N) x2n[threadIdx.x] = splcrc::x2n_entry(threadIdx.x);
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
        for (uint32_t i = threadIdx.x; i < 256u; i += 64u) {
            const uint32_t c = table[(k - 1) * 256 + i];
            table[k * 256 + i] = (c >> 8) ^ table[c & 0xffu];
        }
        __syncthreads();
    }
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != SPL_Z_OK) return;
    const spl_zblock zb = blocks[b];
    if (splcrc::block<S>(out_all + zb.out, zb.out_len, table, x2n) != zb.crc) status[b] = SPL_Z_BAD_CRC;
}

// CRC32 a block per WAVE (spl_crc_wave.h, round 6): rows of 1024 bytes, a coalesced 16-byte load and twenty independent look-ups a
// lane and row.  Workgroups of four waves share the twenty tables (20 KB of LDS: seven workgroups a CU) and stay: a wave takes
// block after block (its number, + the grid's waves, ...), so that the tables are made once per workgroup, not per block.
__global__ __launch_bounds__(256) void spl_crc32_wave_kernel(const uint8_t *out_all, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status)
{
    __shared__ uint32_t t[splcrc::W_TABLE_WORDS + splcrc::W_SCRATCH_WORDS];
    splcrc::wave_tables(t, threadIdx.x, 256u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t factor = splcrc::wave_lane_factor(lane);
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6))); b < n_blocks; b += n_waves) {
        if (status[b] != SPL_Z_OK) continue; // (blocks that already failed keep their status)
        const spl_zblock zb = blocks[b];
        const uint32_t c = splcrc::wave_block(out_all + zb.out, zb.out_len, t, factor);
        if (lane == 0u && c != zb.crc) status[b] = SPL_Z_BAD_CRC;
    }
}

// ---- BAM records ---------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// Does a record plausibly start at c?  (bam_reader.cpp, plausible_record: the same checks.)  len = the record's size with its
// length word.
__device__ __forceinline__ bool plausible_record(const uint8_t *c, const uint8_t *end, int32_t n_ref, uint64_t &len)
{
    if (end - c < 36) return false;
    const uint32_t bs = ld32(c);
    if (bs < 33u || bs > (1u << 29)) return false;
    const uint8_t *r = c + 4;
    const int32_t tid = (int32_t)ld32(r), pos0 = (int32_t)ld32(r + 4);
    const uint32_t l_name = r[8], n_cig = ld16(r + 12);
    const int32_t l_seq = (int32_t)ld32(r + 16), next_tid = (int32_t)ld32(r + 20), next_pos = (int32_t)ld32(r + 24);
    if (tid < -1 || tid >= n_ref || next_tid < -1 || next_tid >= n_ref || pos0 < -1 || next_pos < -1 || l_seq < 0 || l_name < 1u) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1ull) / 2ull + (uint64_t)l_seq;
    if (need > bs) return false;
    if ((uint64_t)(end - c) >= 4ull + 32ull + l_name) {
        const uint8_t *name = r + 32;
        if (name[l_name - 1u] != 0) return false;
        for (uint32_t i = 0; i + 1u < l_name; ++i)
            if (name[i] < 33 || name[i] > 126) return false;
    }
    len = 4ull + bs;
    return true;
}

} // namespace

// `stream_len` = where the inflated bytes end: the stream's end, or (more != 0) the end of the window that is inflated at the
// moment -- a record that runs past it is then no damage but something for the next window (SPL_BS_INCOMPLETE).
__global__ __launch_bounds__(64) void spl_bam_scan_kernel(const uint8_t *stream, uint64_t stream_len, uint64_t header_end, int32_t n_ref, int32_t tid_lo, int32_t tid_hi,
                                                           const spl_zblock *blocks, uint32_t n_blocks, spl_bscan *scan, uint32_t more, uint16_t *recs)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t u0 = blocks[b].out, u1 = u0 + blocks[b].out_len;
    uint16_t *const mine = recs ? recs + (size_t)b * SPL_BS_REC_CAP : nullptr; // where this block's placed records begin, for the extraction
    const uint8_t *const end = stream + stream_len;
    spl_bscan out;
    out.start = out.reached = u1;
    out.n_all = out.n_placed = out.n_ops = 0;
    out.n_foreign = out.n_foreign_hi = 0;
    out.flags = 0;
    out.tid_first = out.tid_last = -1;
    if (u1 <= header_end && !(u1 == header_end && u0 == u1)) { // BAM header bytes only (or an empty block inside them)
        out.start = out.reached = u1 < header_end ? u1 : header_end;
        scan[b] = out;
        return;
    }
    uint64_t at = u0;
    if (u0 <= header_end) {
        at = header_end; // the first record of the file: known, not guessed
    } else {
        // the first plausible start at or after u0 whose next three records chain; the search may run past the block (a record
        // larger than a block) but not for ever
        const uint64_t limit = u0 + (1ull << 20) < stream_len ? u0 + (1ull << 20) : stream_len;
        bool found = false;
        for (; at + 36 <= limit; ++at) {
            uint64_t len = 0;
            if (!plausible_record(stream + at, end, n_ref, len)) continue;
            uint64_t q = at + len;
            bool ok = true;
            for (int k = 0; k < 3 && ok && q + 36 <= stream_len; ++k) {
                uint64_t l2 = 0;
                ok = plausible_record(stream + q, end, n_ref, l2);
                q += l2;
            }
            if (ok) { found = true; break; }
        }
        if (!found) {
            if (limit == stream_len && more) out.flags |= SPL_BS_INCOMPLETE | SPL_BS_NO_START; // (whatever starts here ends beyond the window)
            else if (limit == stream_len) at = stream_len; // (the tail of the last record of the file)
            else out.flags |= SPL_BS_NO_START;
        }
    }
    out.start = at;
    // the records that start before the block's end
    int32_t last_tid = -1;
    while (at < u1 && !(out.flags & (SPL_BS_CORRUPT | SPL_BS_NO_START | SPL_BS_INCOMPLETE))) {
        if (stream_len - at < 4) { out.flags |= more ? SPL_BS_INCOMPLETE : SPL_BS_CORRUPT; break; }
        const uint32_t bs = ld32(stream + at);
        if (bs < 32u) { out.flags |= SPL_BS_CORRUPT; break; }
        if (stream_len - at < 4ull + bs) { out.flags |= more ? SPL_BS_INCOMPLETE : SPL_BS_CORRUPT; break; }
        const uint8_t *r = stream + at + 4;
        const int32_t tid = (int32_t)ld32(r), pos0 = (int32_t)ld32(r + 4);
        const uint32_t l_name = r[8], n_cig = ld16(r + 12), l_seq = ld32(r + 16);
        const uint64_t need = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1ull) / 2ull + (uint64_t)l_seq;
        if (need > bs) { out.flags |= SPL_BS_CORRUPT; break; }
        const int32_t tid_eff = tid < 0 || tid >= n_ref ? n_ref : tid; // (records without a reference: behind all others)
        if (tid_eff < last_tid) out.flags |= SPL_BS_UNSORTED; // (every record counts here, a neighbour's too: the shares are cut on this order)
        last_tid = tid_eff;
        if (tid_eff < tid_lo || tid_eff >= tid_hi) { out.n_foreign++; out.n_foreign_hi += tid_eff >= tid_hi ? 1u : 0u; at += 4ull + bs; continue; }
        out.n_all++;
        if (tid >= 0 && tid < n_ref && pos0 >= 0) {
            if (n_cig > 0) {
                const uint32_t op0 = ld32(r + 32 + l_name);
                if ((op0 & 15u) == 4u && (op0 >> 4) == l_seq && bs > need) out.flags |= SPL_BS_NEEDS_HOST; // maybe a CG tag behind it
            }
            if (out.tid_first < 0) out.tid_first = tid;
            out.tid_last = tid;
            if (mine && out.n_placed < SPL_BS_REC_CAP) mine[out.n_placed] = (uint16_t)(at - u0);
            out.n_placed++;
            out.n_ops += n_cig;
        }
        at += 4ull + bs;
    }
    out.reached = at;
    scan[b] = out;
}

__global__ __launch_bounds__(64) void spl_bam_extract_kernel(const uint8_t *stream, uint64_t stream_len, int32_t n_ref, int32_t tid_lo, int32_t tid_hi, const spl_zblock *blocks, uint32_t n_blocks,
                                                              const spl_bscan *scan, const uint64_t *rec_off, const uint64_t *op_off, int32_t *pos_out,
                                                              uint16_t *flag_out, uint32_t *cig_off, uint32_t *cigar, int32_t *tid_out,
                                                              unsigned long long *ref_max_end)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const spl_bscan sc = scan[b];
    if (sc.n_placed == 0) return;
    const uint64_t u1 = blocks[b].out + blocks[b].out_len;
    uint64_t at = sc.start, i = rec_off[b], o = op_off[b];
    int32_t run_tid = -1;     // the reference of the records seen last and the largest end among them: one atomic per run
    long long run_end = 0;    // (one per record was 20 M atomics on five words: most of the kernel's time)
    // The fields wanted from a record's fixed part lie in its first 20 bytes: two 16-byte loads, and those of the NEXT record are
    // asked for as soon as this one's size is known, before its CIGAR is fetched -- one trip to memory per record instead of
    // one for the size, one for the fields and one for the CIGAR.  (The stream is padded: 32 bytes can be read at any at < u1.)
    u32x4 h0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0};
    if (at < u1) { __builtin_memcpy(&h0, stream + at, 16); __builtin_memcpy(&h1, stream + at + 16, 16); }
    while (at < u1) {
        const uint32_t bs = h0.x;
        const int32_t tid = (int32_t)h0.y, pos0 = (int32_t)h0.z;
        const uint32_t l_name = h0.w & 0xffu, n_cig = h1.x & 0xffffu, flag = h1.x >> 16;
        const uint8_t *cig = stream + at + 36 + l_name;
        at += 4ull + bs;
        if (at < u1) { __builtin_memcpy(&h0, stream + at, 16); __builtin_memcpy(&h1, stream + at + 16, 16); }
        if (tid >= 0 && tid < n_ref && pos0 >= 0 && tid >= tid_lo && tid < tid_hi) {
            long long ref_len = 0;
            for (uint32_t k = 0; k < n_cig; ++k) {
                const uint32_t op = ld32(cig + 4ull * k);
                cigar[o + k] = op;
                const uint32_t code = op & 15u;
                if (code == 0u || code == 2u || code == 3u || code == 7u || code == 8u) ref_len += (long long)(op >> 4);
            }
            o += n_cig;
            pos_out[i] = pos0 + 1;
            flag_out[i] = (uint16_t)flag;
            tid_out[i] = tid;
            cig_off[i + 1] = (uint32_t)o;
            const long long e = (long long)pos0 + 1 + (ref_len > 0 ? ref_len : 1) - 1;
            if (tid != run_tid) {
                if (run_tid >= 0) atomicMax(&ref_max_end[run_tid], (unsigned long long)run_end);
                run_tid = tid;
                run_end = e;
            } else if (e > run_end) {
                run_end = e;
            }
            ++i;
        }
    }
    if (run_tid >= 0) atomicMax(&ref_max_end[run_tid], (unsigned long long)run_end);
}

// Where every reference's records begin: (first record, tid | first CIGAR op << 32) per run of equal tids, in no particular order.
// The same extraction with a WAVE per block: the scan has left where every placed record of the block begins (16 bits each), so
// 64 records are read at once -- each lane its record's fixed fields and CIGAR, none waiting for the one before as the walk
// above has to -- and written side by side: positions, flags and CIGAR offsets of consecutive records by consecutive lanes.
// Where a record's ops go follows from a prefix sum over the lanes' op counts.
__global__ __launch_bounds__(64) void spl_bam_extract_wave_kernel(const uint8_t *stream, int32_t n_ref, const spl_zblock *blocks, uint32_t n_blocks, const spl_bscan *scan,
                                                                   const uint16_t *recs, const uint64_t *rec_off, const uint64_t *op_off, int32_t *pos_out, uint16_t *flag_out,
                                                                   uint32_t *cig_off, uint32_t *cigar, int32_t *tid_out, unsigned long long *ref_max_end)
{
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint32_t n = scan[b].n_placed;
    if (n == 0) return;
    const uint32_t l = threadIdx.x;
    const uint64_t u0 = blocks[b].out, i0 = rec_off[b];
    const uint16_t *const mine = recs + (size_t)b * SPL_BS_REC_CAP;
    uint64_t o_base = op_off[b]; // (where the ops of this round's first record go)
    int32_t run_tid = -1;
    long long run_end = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
        const uint32_t j = j0 + l;
        const bool have = j < n;
        uint32_t n_cig = 0, l_name = 0, flag = 0;
        int32_t tid = -1, pos0 = 0;
        const uint8_t *r = stream;
        if (have) {
            r = stream + u0 + mine[j];
            u32x4 h0, h1;
            __builtin_memcpy(&h0, r, 16);
            __builtin_memcpy(&h1, r + 16, 16);
            tid = (int32_t)h0.y; pos0 = (int32_t)h0.z;
            l_name = h0.w & 0xffu; n_cig = h1.x & 0xffffu; flag = h1.x >> 16;
        }
        // ops of the records before mine in this round
        uint32_t incl = n_cig;
#pragma unroll
        for (uint32_t s = 1; s < 64u; s <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, s, 64);
            if (l >= s) incl += up;
        }
        const uint64_t o = o_base + incl - n_cig;
        o_base += (uint32_t)__shfl((int)incl, 63, 64);
        if (have) {
            const uint8_t *cig = r + 36 + l_name;
            long long ref_len = 0;
            for (uint32_t k = 0; k < n_cig; ++k) {
                const uint32_t op = ld32(cig + 4ull * k);
                cigar[o + k] = op;
                const uint32_t code = op & 15u;
                if (code == 0u || code == 2u || code == 3u || code == 7u || code == 8u) ref_len += (long long)(op >> 4);
            }
            const uint64_t i = i0 + j;
            pos_out[i] = pos0 + 1;
            flag_out[i] = (uint16_t)flag;
            tid_out[i] = tid;
            cig_off[i + 1] = (uint32_t)(o + n_cig);
            const long long e = (long long)pos0 + 1 + (ref_len > 0 ? ref_len : 1) - 1;
            if (tid != run_tid) {
                if (run_tid >= 0) atomicMax(&ref_max_end[run_tid], (unsigned long long)run_end); // (a block that holds the end of one reference and the beginning of the next)
                run_tid = tid;
                run_end = e;
            } else if (e > run_end) {
                run_end = e;
            }
        }
    }
    // the largest end per reference: one atomic for all lanes that hold the same one
    for (;;) {
        const unsigned long long pending = __ballot(run_tid >= 0);
        if (!pending) break;
        const int first = __ffsll((long long)pending) - 1;
        const int32_t t = (int32_t)__shfl((int)run_tid, first, 64);
        long long m = run_tid == t ? run_end : 0;
#pragma unroll
        for (uint32_t s = 32; s; s >>= 1) {
            const long long other = __shfl_xor(m, (int)s, 64);
            m = other > m ? other : m;
        }
        if ((int)l == first) atomicMax(&ref_max_end[t], (unsigned long long)m);
        if (run_tid == t) run_tid = -1;
    }
}

__global__ __launch_bounds__(256) void spl_bam_bounds_kernel(const int32_t *tid, const uint32_t *cig_off, uint64_t n, uint64_t *bounds, uint32_t *n_bounds, uint32_t cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || tid[i] != tid[i - 1]) {
        const uint32_t k = atomicAdd(n_bounds, 1u);
        if (k < cap) { bounds[2 * k] = i; bounds[2 * k + 1] = (uint64_t)(uint32_t)tid[i] | ((uint64_t)cig_off[i] << 32); }
    }
}

extern "C" int spl_dev_launch_bam_scan(const uint8_t *stream, uint64_t stream_len, uint64_t header_end, int32_t n_ref, int32_t tid_lo, int32_t tid_hi, const spl_zblock *blocks,
                                       uint32_t n_blocks, spl_bscan *scan, int more, uint16_t *recs, void *st)
{
    if (n_blocks == 0) return 0;
    const uint32_t L = 8; // (blocks per wave: the kernel is lanes waiting for memory, a wave as slow as its slowest lane -- 1.1 ms per window of 49 152 blocks with 8, 1.7 with 64, 3.9 with 1)
    hipLaunchKernelGGL(spl_bam_scan_kernel, dim3((n_blocks + L - 1u) / L), dim3(L), 0, (hipStream_t)st, stream, stream_len, header_end, n_ref, tid_lo, tid_hi, blocks, n_blocks, scan, more ? 1u : 0u, recs);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_bam_extract(const uint8_t *stream, uint64_t stream_len, int32_t n_ref, int32_t tid_lo, int32_t tid_hi, const spl_zblock *blocks, uint32_t n_blocks, const spl_bscan *scan,
                                          const uint64_t *rec_off, const uint64_t *op_off, int32_t *pos, uint16_t *flag, uint32_t *cig_off, uint32_t *cigar,
                                          int32_t *tid, unsigned long long *ref_max_end, const uint16_t *recs, void *st)
{
    if (n_blocks == 0) return 0;
    if (recs) { // (the scan of these very blocks has left the records' places: a wave per block)
        hipLaunchKernelGGL(spl_bam_extract_wave_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)st, stream, n_ref, blocks, n_blocks, scan, recs, rec_off, op_off, pos, flag, cig_off, cigar,
                           tid, ref_max_end);
        return (int)hipGetLastError();
    }
    const uint32_t L = 64;
    hipLaunchKernelGGL(spl_bam_extract_kernel, dim3((n_blocks + L - 1u) / L), dim3(L), 0, (hipStream_t)st, stream, stream_len, n_ref, tid_lo, tid_hi, blocks, n_blocks, scan, rec_off,
                       op_off, pos, flag, cig_off, cigar, tid, ref_max_end);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_bam_bounds(const int32_t *tid, const uint32_t *cig_off, uint64_t n, uint64_t *bounds, uint32_t *n_bounds, uint32_t cap, void *st)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(spl_bam_bounds_kernel, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, (hipStream_t)st, tid, cig_off, n, bounds, n_bounds, cap);
    return (int)hipGetLastError();
}

static size_t tokens_at(uint32_t n_blocks) { return ((size_t)n_blocks * 4 + 255) / 256 * 256; }
extern "C" size_t spl_dev_inflate_work_bytes2(uint32_t n_blocks, uint32_t stride) { return 256 + tokens_at(n_blocks) + (size_t)n_blocks * stride; }
extern "C" size_t spl_dev_inflate_work_bytes(uint32_t n_blocks) { return spl_dev_inflate_work_bytes2(n_blocks, SPL_Z_TOKEN_STRIDE); }

extern "C" int spl_dev_launch_inflate_decode2(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, uint32_t stride, void *stream)
{
    // (by the stride alone: the token room a caller gives says how well its blocks deflate -- 2.5 x a block + 4 KB is the rule of thumb)
    return spl_dev_launch_inflate_decode3(image, blocks, n_blocks, status, work, stride, stride <= 34816u ? SPL_Z_LAUNCH_DENSE : 0u, stream);
}

extern "C" int spl_dev_launch_inflate_decode3(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, uint32_t stride, uint32_t flags, void *stream)
{
    if (n_blocks == 0) return 0;
    if (!work) return (int)hipErrorInvalidValue; // (nothing would be launched and the blocks' status words left as they were: an error, not a success)
    if (stride < 256u || stride > SPL_Z_TOKEN_STRIDE || (stride & 15u)) return (int)hipErrorInvalidValue;
    // SPL_Z_WRITING_PASS=1 (A/B): every tile's tokens by a writing pass of their own, as until round 5 (now: written while the last count is taken)
    static const uint32_t opts = getenv("SPL_Z_WRITING_PASS") ? splz::OPT_WRITING_PASS : 0u;
    // (measured, round 6: fewer decoding waves a CU -- 12, 9, 8 instead of the 16 that fill its LDS, by padding -- make room for the
    //  copying kernel's waves and only slow the decoder, 294 -> 345 / 412 / 468 ms a human file, the copying kernel 234 -> 215:
    //  profiles/r06n_decode_lds_pad_q2.txt)
    // Which kernel: the caller's word (spl_capi.cpp: files whose blocks deflate to 12 KB or less on average get the denser one).
    // SPL_Z_DENSE=1 / 0: always / never (tests, A/B).
    const char *const de = getenv("SPL_Z_DENSE"); // (looked up at every launch: the tests switch it)
    const int dense_env = de ? atoi(de) : -1;
    const bool dense = dense_env >= 0 ? dense_env != 0 : (flags & SPL_Z_LAUNCH_DENSE) != 0u;
    if (dense)
        hipLaunchKernelGGL(spl_inflate_decode_dense_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, image, blocks, n_blocks, status, (uint8_t *)work + tokens_at(n_blocks), (uint32_t *)work, stride, opts);
    else
        hipLaunchKernelGGL(spl_inflate_decode_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, image, blocks, n_blocks, status, (uint8_t *)work + tokens_at(n_blocks), (uint32_t *)work, stride, opts);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_inflate_copy2(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, uint32_t stride, void *stream)
{
    if (n_blocks == 0) return 0;
    if (!work) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(spl_inflate_copy_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, blocks, n_blocks, out, status, (const uint8_t *)work + tokens_at(n_blocks),
                       (const uint32_t *)work, stride);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_inflate_decode(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, void *stream)
{
    return spl_dev_launch_inflate_decode2(image, blocks, n_blocks, status, work, SPL_Z_TOKEN_STRIDE, stream);
}

extern "C" int spl_dev_launch_inflate_copy(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream)
{
    return spl_dev_launch_inflate_copy2(blocks, n_blocks, out, status, work, SPL_Z_TOKEN_STRIDE, stream);
}

extern "C" int spl_dev_launch_inflate(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream)
{
    if (n_blocks == 0) return 0;
    if (!work) return (int)hipErrorInvalidValue; // (the two kernels exchange the blocks' token streams there: spl_dev_inflate_work_bytes)
    const int rc = spl_dev_launch_inflate_decode(image, blocks, n_blocks, status, work, stream);
    return rc ? rc : spl_dev_launch_inflate_copy(blocks, n_blocks, out, status, work, stream);
}

extern "C" int spl_dev_launch_crc32(const uint8_t *out, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *stream)
{
    if (n_blocks == 0) return 0;
    // SPL_CRC_LANES=1 (A/B): round 4's kernel, a block per lane as four streams (1.6 ms a window of 43 169 blocks alone, 2-4 in the pipeline)
    static const bool per_lane = getenv("SPL_CRC_LANES") != nullptr;
    if (per_lane) {
        hipLaunchKernelGGL(spl_crc32_kernel<4>, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, out, blocks, n_blocks, status);
        return (int)hipGetLastError();
    }
    // a block per wave, four waves a workgroup, at most six workgroups per CU's worth of grid: the waves take block after block
    const uint32_t groups = std::min<uint32_t>((n_blocks + 3u) / 4u, 256u * 6u);
    hipLaunchKernelGGL(spl_crc32_wave_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, out, blocks, n_blocks, status);
    return (int)hipGetLastError();
}
