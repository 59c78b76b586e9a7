// CRC32 (IEEE 802.3, reflected: what a BGZF block's trailer holds, RFC 1952 §8) of one block's payload by ONE lane, as S streams.
//
// A lane that walks its block a word after the other waits for every step: the register after a word depends on the register
// before it (four table look-ups, each a trip to LDS), and the 16 bytes it works on next have to have arrived.  The CRC register
// is linear in the data over GF(2), so the payload is cut into S parts of the same length (a multiple of 16 bytes) that are walked
// side by side with a register each -- S independent chains of look-ups, S loads in flight -- and the registers are put together
// afterwards: the register of a part followed by m zero bytes is the register times x^(8m) modulo the polynomial (mulmod, xpow:
// zlib's crc32_combine does the same on the host).  What is left behind the S parts (less than 16·S + 16 bytes) is walked alone.
//
// Shared by the kernel (spl_inflate.hip) and by the host build the CPU tests check against zlib (tests/hostsim/crc_host.cpp).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SPL_CRC_FN __host__ __device__ inline
#else
#define SPL_CRC_FN inline
#endif

namespace splcrc {

constexpr uint32_t POLY = 0xEDB88320u;
constexpr int N_X2N = 24; // x^(2^k), k < 24: exponents (bits of payload) below 2^24 -- a BGZF payload has 2^19 at most

// table[0][b] of the byte-wise method
SPL_CRC_FN uint32_t byte_entry(uint32_t b)
{
    uint32_t c = b;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (POLY ^ (c >> 1)) : (c >> 1);
    return c;
}

// a · b modulo the polynomial; bit 31 is the coefficient of x^0 (the reflected order the register is kept in)
SPL_CRC_FN uint32_t mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - (a >> 31));
        a <<= 1;
        b = (b >> 1) ^ (POLY & (0u - (b & 1u)));
    }
    return p;
}

// x^(2^k): k squarings of x
SPL_CRC_FN uint32_t x2n_entry(uint32_t k)
{
    uint32_t v = 0x40000000u; // x^1
    for (uint32_t j = 0; j < k; ++j) v = mulmod(v, v);
    return v;
}

// x^e from the table of x^(2^k)
SPL_CRC_FN uint32_t xpow(const uint32_t *x2n, uint32_t e)
{
    uint32_t r = 0x80000000u; // x^0
    for (uint32_t k = 0; e != 0u; ++k, e >>= 1)
        if (e & 1u) r = mulmod(r, x2n[k]);
    return r;
}

typedef uint32_t Word16 __attribute__((vector_size(16))); // (a vector, so that the 16 bytes stay ONE load of unknown alignment)

// the four tables of slicing by four as one array: t[k * 256 + b] = the register after byte b and k zero bytes
SPL_CRC_FN uint32_t step_word(const uint32_t *t, uint32_t c, uint32_t w)
{
    c ^= w;
    return t[768 + (c & 0xffu)] ^ t[512 + ((c >> 8) & 0xffu)] ^ t[256 + ((c >> 16) & 0xffu)] ^ t[c >> 24];
}

// The CRC32 of p[0 .. n) (final complement applied).  Never reads behind p + n.
template <int S>
SPL_CRC_FN uint32_t block(const uint8_t *p, uint32_t n, const uint32_t *t, const uint32_t *x2n)
{
    const uint32_t part = (n / (16u * S)) * 16u;
    uint32_t c[S];
    c[0] = 0xffffffffu;
#pragma unroll
    for (int s = 1; s < S; ++s) c[s] = 0u;
    if (part != 0u) {
        Word16 cur[S], next[S];
#pragma unroll
        for (int s = 0; s < S; ++s) memcpy(&cur[s], p + (uint32_t)s * part, 16);
        for (uint32_t i = 16u; i < part; i += 16u) {
#pragma unroll
            for (int s = 0; s < S; ++s) memcpy(&next[s], p + (uint32_t)s * part + i, 16);
#pragma unroll
            for (int s = 0; s < S; ++s) c[s] = step_word(t, c[s], cur[s][0]);
#pragma unroll
            for (int s = 0; s < S; ++s) c[s] = step_word(t, c[s], cur[s][1]);
#pragma unroll
            for (int s = 0; s < S; ++s) c[s] = step_word(t, c[s], cur[s][2]);
#pragma unroll
            for (int s = 0; s < S; ++s) c[s] = step_word(t, c[s], cur[s][3]);
#pragma unroll
            for (int s = 0; s < S; ++s) cur[s] = next[s];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            c[s] = step_word(t, c[s], cur[s][0]);
            c[s] = step_word(t, c[s], cur[s][1]);
            c[s] = step_word(t, c[s], cur[s][2]);
            c[s] = step_word(t, c[s], cur[s][3]);
        }
    }
    uint32_t acc = c[0];
    if (S > 1 && part != 0u) {
        const uint32_t shift = xpow(x2n, part * 8u);
#pragma unroll
        for (int s = 1; s < S; ++s) acc = mulmod(acc, shift) ^ c[s];
    }
    uint32_t i = part * (uint32_t)S;
    for (; i + 4u <= n; i += 4u) {
        uint32_t w;
        memcpy(&w, p + i, 4);
        acc = step_word(t, acc, w);
    }
    for (; i < n; ++i) acc = t[(acc ^ p[i]) & 0xffu] ^ (acc >> 8);
    return acc ^ 0xffffffffu;
}

} // namespace splcrc
