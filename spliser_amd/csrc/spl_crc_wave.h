// spl_crc_wave.h -- CRC32 (IEEE 802.3, reflected: a BGZF block's trailer, RFC 1952 section 8) of one block's payload by ONE WAVE.
//
// Round 4's kernel gave a block to a lane (spl_crc.h): 768 waves for a window of 49 152 blocks, every lane of a wave on a 64 KiB
// stretch of its own -- 64 (x 4 streams) different lines per load instruction, a serial chain of table look-ups per stream -- and
// ran at a tenth of the memory's rate, 47-88 ms of a human file's decode.  Here the wave walks the block in ROWS of 1024 bytes,
// lane i bytes [16 i, 16 i + 16) of every row: one coalesced 16-byte load per lane and row.  The CRC register is linear in the data
// over GF(2), so a lane keeps a register of its own for "my 16-byte chunks, 1008 bytes apart":
//
//     a  <-  a * x^8192  +  S(chunk)            per row        (x^8192: the register moved past the 1024 bytes of a row)
//
// with S(chunk) = the register a chunk leaves when it starts from zero.  Both terms are look-ups that do NOT depend on each
// other: S(chunk) = sum over the chunk's 16 bytes of D[j][byte j] (D[j][b]: byte b followed by 15 - j zero bytes), and
// a * x^8192 = sum over the register's four bytes of M[k][byte k] -- twenty tables of 256 words in LDS, twenty independent
// look-ups a row, no chain longer than one look-up and an xor.  At the end lane i's register stands 16 (63 - i) bytes in front
// of the block's end: times x^(128 (63 - i)) (one multiplication modulo the polynomial per lane and block, the factor a
// constant of the lane), and the 64 products are xor-ed together.
//
// Two details.  The rows are laid against the block's END (its length is anything): the first row begins before the block, and
// what lies there counts as zero bytes -- leading zeros do not change a register that starts from zero.  And the register does
// start from zero: the standard's initial value 0xffffffff is the same as the block's first four bytes complemented, which is
// done to the data of the first row (blocks of fewer than 16 bytes are walked byte by byte).
//
// Written against spl_wave.h, so that the very same source runs on the CPU under the wave emulator (tests/hostsim/crc_wave_host.cpp,
// tests/test_crc_host.py: against zlib).
#pragma once
#include "spl_crc.h"
#include "spl_wave.h"

namespace splcrc {

constexpr uint32_t W_ROW = 1024;          // bytes a wave takes per step
constexpr uint32_t W_TABLES = 20;         // D[0..15], M[0..3]: 256 words each
constexpr uint32_t W_TABLE_WORDS = W_TABLES * 256u;
constexpr uint32_t W_SCRATCH_WORDS = 32;  // behind the tables: the 32 basis values M is made from

// x^(2^k) modulo the polynomial, by hand-rolled squaring at compile time (mulmod is a plain loop)
constexpr uint32_t c_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - (a >> 31));
        a <<= 1;
        b = (b >> 1) ^ (POLY & (0u - (b & 1u)));
    }
    return p;
}
constexpr uint32_t c_x2n(int k)
{
    uint32_t v = 0x40000000u;
    for (int j = 0; j < k; ++j) v = c_mulmod(v, v);
    return v;
}
constexpr uint32_t X_ROW = c_x2n(13);     // x^8192: a register moved past one row

// The lane's factor x^(128 (63 - lane)): what its register is multiplied by when the lanes' registers are put together.
WV_DEV uint32_t wave_lane_factor(uint32_t lane)
{
    constexpr uint32_t x2n[6] = {c_x2n(7), c_x2n(8), c_x2n(9), c_x2n(10), c_x2n(11), c_x2n(12)}; // x^128 ... x^4096
    const uint32_t e = 63u - lane; // in units of 128 bits
    uint32_t r = 0x80000000u;      // x^0
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k)
        if ((e >> k) & 1u) r = mulmod(r, x2n[k]);
    return r;
}

// The twenty tables, by n_threads threads of which this is number tid (a workgroup on the device, a wave under the emulator):
// t[W_TABLE_WORDS + W_SCRATCH_WORDS].  wv::sync() between the rounds: every thread of the workgroup must call this.
WV_DEV void wave_tables(uint32_t *t, uint32_t tid, uint32_t n_threads)
{
    for (uint32_t b = tid; b < 256u; b += n_threads) t[15u * 256u + b] = byte_entry(b);
    if (tid < 32u) t[W_TABLE_WORDS + tid] = mulmod(1u << tid, X_ROW);
    wv::sync();
    for (int j = 14; j >= 0; --j) { // one more zero byte behind the byte: D[j] from D[j + 1]
        for (uint32_t b = tid; b < 256u; b += n_threads) {
            const uint32_t c = t[(uint32_t)(j + 1) * 256u + b];
            t[(uint32_t)j * 256u + b] = (c >> 8) ^ t[15u * 256u + (c & 0xffu)];
        }
        wv::sync();
    }
    for (uint32_t e = tid; e < 1024u; e += n_threads) { // M[k][b] = (b << 8 k) * x^8192: the xor of its bits' products
        const uint32_t k = e >> 8, b = e & 0xffu;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) v ^= ((b >> q) & 1u) ? t[W_TABLE_WORDS + 8u * k + q] : 0u;
        t[(16u + k) * 256u + b] = v;
    }
    wv::sync();
}

// S(chunk): the register sixteen bytes leave, starting from zero
WV_DEV uint32_t wave_chunk(const uint32_t *t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    const uint32_t a = t[0u * 256u + (w0 & 0xffu)] ^ t[1u * 256u + ((w0 >> 8) & 0xffu)] ^ t[2u * 256u + ((w0 >> 16) & 0xffu)] ^ t[3u * 256u + (w0 >> 24)];
    const uint32_t b = t[4u * 256u + (w1 & 0xffu)] ^ t[5u * 256u + ((w1 >> 8) & 0xffu)] ^ t[6u * 256u + ((w1 >> 16) & 0xffu)] ^ t[7u * 256u + (w1 >> 24)];
    const uint32_t c = t[8u * 256u + (w2 & 0xffu)] ^ t[9u * 256u + ((w2 >> 8) & 0xffu)] ^ t[10u * 256u + ((w2 >> 16) & 0xffu)] ^ t[11u * 256u + (w2 >> 24)];
    const uint32_t d = t[12u * 256u + (w3 & 0xffu)] ^ t[13u * 256u + ((w3 >> 8) & 0xffu)] ^ t[14u * 256u + ((w3 >> 16) & 0xffu)] ^ t[15u * 256u + (w3 >> 24)];
    return (a ^ b) ^ (c ^ d);
}
// a * x^8192
WV_DEV uint32_t wave_advance(const uint32_t *t, uint32_t a)
{
    return t[16u * 256u + (a & 0xffu)] ^ t[17u * 256u + ((a >> 8) & 0xffu)] ^ t[18u * 256u + ((a >> 16) & 0xffu)] ^ t[19u * 256u + (a >> 24)];
}

// One word of the block's first row: the four bytes at block offset wo (which may lie in front of the block: zeros), the
// block's first four bytes complemented.  Never reads in front of p.
WV_DEV uint32_t wave_first_row_word(const uint8_t *p, int32_t wo)
{
    if (wo >= 4) return wv::ld32(p + wo);
    if (wo >= 0) return wv::ld32(p + wo) ^ (0xffffffffu >> (8u * (uint32_t)wo));
    if (wo <= -4) return 0u;
    uint32_t w = 0;
    for (int32_t b = -wo; b < 4; ++b) w |= ((uint32_t)p[wo + b] ^ 0xffu) << (8 * b); // (offsets 0 .. 3 + wo < 4: all complemented)
    return w;
}

// The CRC32 of p[0 .. n) by the calling wave; every lane gets the value.  t: the tables (wave_tables), factor: wave_lane_factor
// of the lane.  Never reads in front of p or behind p + n.
WV_DEV uint32_t wave_block(const uint8_t *p, uint32_t n, const uint32_t *t, uint32_t factor)
{
    const uint32_t lane = wv::lane();
    if (n < 16u) { // (wave-uniform) the BGZF end-of-file block, mostly: byte by byte, every lane the same
        uint32_t c = 0xffffffffu;
        for (uint32_t i = 0; i < n; ++i) c = t[15u * 256u + ((c ^ p[i]) & 0xffu)] ^ (c >> 8);
        return c ^ 0xffffffffu;
    }
    const uint32_t rows = (n + W_ROW - 1u) / W_ROW;
    int32_t o = (int32_t)n - (int32_t)(rows * W_ROW) + 16 * (int32_t)lane; // of the lane's chunk in the first row: negative = in front of the block
    uint32_t a = wave_chunk(t, wave_first_row_word(p, o), wave_first_row_word(p, o + 4), wave_first_row_word(p, o + 8), wave_first_row_word(p, o + 12));
    // the rows behind the first lie inside the block (and all but the second's first bytes behind the complemented four): one 16-byte
    // load a lane, the next row's asked for before this one's look-ups
    uint64_t lo = 0, hi = 0;
    if (rows > 1u) {
        wv::ld128(p + (o + (int32_t)W_ROW), lo, hi);
        const int32_t o1 = o + (int32_t)W_ROW; // >= 1: the block's first four bytes reach into the second row when its length is 1 .. 3 modulo the row
        if (o1 < 4) lo ^= (uint64_t)(0xffffffffu >> (8u * (uint32_t)o1));
    }
    for (uint32_t r = 1; r < rows; ++r) {
        o += (int32_t)W_ROW;
        const uint64_t c_lo = lo, c_hi = hi;
        if (r + 1u < rows) wv::ld128(p + (o + (int32_t)W_ROW), lo, hi);
        a = wave_advance(t, a) ^ wave_chunk(t, (uint32_t)c_lo, (uint32_t)(c_lo >> 32), (uint32_t)c_hi, (uint32_t)(c_hi >> 32));
    }
    // lane i's register stands 16 (63 - i) bytes in front of the end
    uint32_t v = mulmod(a, factor);
#pragma unroll
    for (uint32_t s = 32u; s; s >>= 1) v ^= wv::shfl(v, lane ^ s);
    return v ^ 0xffffffffu;
}

} // namespace splcrc
