// spl_inflate.hip -- see spl_inflate.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spl_inflate.h"

namespace {

struct BitReader {
    const uint32_t *p;   // next aligned word of the image
    const uint8_t *end;  // one past the block's DEFLATE data
    uint64_t buf;
    uint32_t cnt;        // valid bits in buf
    __device__ __forceinline__ void init(const uint8_t *at, const uint8_t *e)
    {
        end = e;
        const uintptr_t a = (uintptr_t)at;
        p = (const uint32_t *)(a & ~(uintptr_t)3);
        const uint32_t skip = (uint32_t)(a & 3u) * 8u;
        buf = (uint64_t)(*p++) >> skip;
        cnt = 32u - skip;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32u) { buf |= (uint64_t)(*p++) << cnt; cnt += 32u; }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    // bytes of the block consumed so far (whole bytes still in the buffer given back)
    __device__ __forceinline__ const uint8_t *pos() const { return (const uint8_t *)p - (cnt >> 3); }
};

// A canonical Huffman table: how many codes of each length (1..15), packed two per word so that the decode loop's fixed
// indices keep them in registers, and the symbols in order of code (the caller's scratch array).
struct Counts {
    uint32_t w[8]; // count[len] = (w[len >> 1] >> (16 * (len & 1))) & 0xffff
};

template <int MAXSYM>
__device__ __forceinline__ bool build_table(const uint8_t *lengths, int n, Counts &c, uint16_t *symbol)
{
    uint16_t count[16], offs[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lengths[s]]++;
    // over-subscribed sets are an error; incomplete ones are legal only in the cases RFC 1951 allows, which the decode loop
    // handles by failing to find a code
    int left = 1;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= (int)count[l];
        if (left < 0) return false;
    }
    offs[1] = 0;
#pragma unroll
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (lengths[s]) symbol[offs[lengths[s]]++] = (uint16_t)s;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.w[k] = (uint32_t)count[2 * k] | ((uint32_t)count[2 * k + 1] << 16);
    c.w[0] &= 0xffff0000u; // (codes of length 0 do not exist)
    return true;
}

// One symbol.  -1: no code matches (corrupt data, or an incomplete table was asked for a code it does not have).
__device__ __forceinline__ int decode_symbol(BitReader &br, const Counts &c, const uint16_t *symbol)
{
    br.refill();
    uint32_t bits = (uint32_t)br.buf;
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)((c.w[len >> 1] >> (16 * (len & 1))) & 0xffffu);
        if (code - count < first) {
            br.drop((uint32_t)len);
            return (int)symbol[index + (code - first)];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__constant__ uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t k_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

} // namespace

__global__ __launch_bounds__(64) void spl_inflate_kernel(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out_all, uint32_t *status)
{
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    const spl_zblock zb = blocks[b];
    uint8_t *const out = out_all + zb.out;
    const uint32_t out_len = zb.out_len;
    uint32_t at = 0; // bytes written
    uint32_t err = SPL_Z_OK;
    if (out_len == 0) { status[b] = SPL_Z_OK; return; } // (the EOF marker and other empty blocks: nothing to decode into)
    BitReader br;
    br.init(image + zb.in, image + zb.in + zb.in_len);
    uint16_t lsym[288], dsym[32];
    uint8_t lengths[320];
    Counts lc, dc;
    for (int last = 0; !last && err == SPL_Z_OK;) {
        br.refill();
        last = (int)br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0u) { // stored: skip to a byte boundary, LEN, NLEN, bytes
            br.drop(br.cnt & 7u);
            br.refill();
            const uint32_t len = br.take(16);
            br.refill();
            const uint32_t nlen = br.take(16);
            if ((len ^ 0xffffu) != nlen) { err = SPL_Z_BAD_STORED; break; }
            const uint8_t *src = br.pos();
            if (src + len > br.end || at + len > out_len) { err = SPL_Z_OVERRUN; break; }
            for (uint32_t i = 0; i < len; ++i) out[at + i] = src[i];
            at += len;
            br.init(src + len, br.end);
            continue;
        }
        if (type == 3u) { err = SPL_Z_BAD_BLOCK_TYPE; break; }
        if (type == 1u) { // fixed codes
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            build_table<288>(lengths, 288, lc, lsym);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            build_table<32>(lengths, 30, dc, dsym);
        } else { // dynamic codes
            br.refill();
            const int nlen = (int)br.take(5) + 257, ndist = (int)br.take(5) + 1, ncode = (int)br.take(4) + 4;
            if (nlen > 286 || ndist > 30) { err = SPL_Z_BAD_LENGTHS; break; }
            uint8_t cl[19];
#pragma unroll
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (int i = 0; i < ncode; ++i) { br.refill(); cl[k_clen_order[i]] = (uint8_t)br.take(3); }
            Counts cc;
            uint16_t csym[19];
            if (!build_table<19>(cl, 19, cc, csym)) { err = SPL_Z_BAD_LENGTHS; break; }
            int idx = 0;
            while (idx < nlen + ndist) {
                const int sym = decode_symbol(br, cc, csym);
                if (sym < 0) { err = SPL_Z_BAD_CODE; break; }
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int prev = 0, rep;
                br.refill();
                if (sym == 16) {
                    if (idx == 0) { err = SPL_Z_BAD_LENGTHS; break; }
                    prev = lengths[idx - 1];
                    rep = 3 + (int)br.take(2);
                } else if (sym == 17) {
                    rep = 3 + (int)br.take(3);
                } else {
                    rep = 11 + (int)br.take(7);
                }
                if (idx + rep > nlen + ndist) { err = SPL_Z_BAD_LENGTHS; break; }
                while (rep--) lengths[idx++] = (uint8_t)prev;
            }
            if (err != SPL_Z_OK) break;
            if (lengths[256] == 0) { err = SPL_Z_BAD_LENGTHS; break; } // no end-of-block code
            if (!build_table<288>(lengths, nlen, lc, lsym)) { err = SPL_Z_BAD_LENGTHS; break; }
            if (!build_table<32>(lengths + nlen, ndist, dc, dsym)) { err = SPL_Z_BAD_LENGTHS; break; }
        }
        // the symbols of the block
        for (;;) {
            if (br.pos() > br.end + 8) { err = SPL_Z_OVERRUN; break; }
            int sym = decode_symbol(br, lc, lsym);
            if (sym < 0) { err = SPL_Z_BAD_CODE; break; }
            if (sym < 256) {
                if (at >= out_len) { err = SPL_Z_OVERRUN; break; }
                out[at++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = SPL_Z_BAD_CODE; break; }
            br.refill();
            const uint32_t len = (uint32_t)k_len_base[sym] + br.take(k_len_extra[sym]);
            const int ds = decode_symbol(br, dc, dsym);
            if (ds < 0 || ds >= 30) { err = SPL_Z_BAD_CODE; break; }
            br.refill();
            const uint32_t dist = (uint32_t)k_dist_base[ds] + br.take(k_dist_extra[ds]);
            if (dist > at) { err = SPL_Z_BAD_DISTANCE; break; }
            if (at + len > out_len) { err = SPL_Z_OVERRUN; break; }
            for (uint32_t i = 0; i < len; ++i) out[at + i] = out[at + i - dist];
            at += len;
        }
    }
    if (err == SPL_Z_OK && at != out_len) err = SPL_Z_SHORT;
    status[b] = err;
}

// CRC32 (IEEE, reflected) of every block's payload against the value in its trailer: one lane per block, a byte at a time
// through the 256-entry table in LDS.  Blocks that already failed keep their status.
__global__ __launch_bounds__(64) void spl_crc32_kernel(const uint8_t *out_all, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status)
{
    __shared__ uint32_t table[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        table[i] = c;
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != SPL_Z_OK) return;
    const spl_zblock zb = blocks[b];
    const uint8_t *p = out_all + zb.out;
    uint32_t c = 0xffffffffu;
    for (uint32_t i = 0; i < zb.out_len; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    if ((c ^ 0xffffffffu) != zb.crc) status[b] = SPL_Z_BAD_CRC;
}

extern "C" int spl_dev_launch_inflate(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *stream)
{
    if (n_blocks == 0) return 0;
    hipLaunchKernelGGL(spl_inflate_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, image, blocks, n_blocks, out, status);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_crc32(const uint8_t *out, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *stream)
{
    if (n_blocks == 0) return 0;
    hipLaunchKernelGGL(spl_crc32_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, out, blocks, n_blocks, status);
    return (int)hipGetLastError();
}
