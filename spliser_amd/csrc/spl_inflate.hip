// spl_inflate.hip -- see spl_inflate.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "spl_inflate.h"
#include "spl_wave.h"
#include "spl_inflate_wave.h"
#include "spl_crc.h"

// The Huffman decoding, one BGZF block per WAVE (the block's symbols as a stream of tokens), and the block's bytes made from
// that stream, one block per LANE (spl_inflate_wave.h has the method, and is what the host tests run through the wave emulator).
__global__ __launch_bounds__(64) void spl_inflate_decode_kernel(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status,
                                                                uint8_t *tokens_all, uint32_t *n_tok)
{
    __shared__ splz::Shared sh;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const spl_zblock zb = blocks[b];
    uint32_t n = 0;
    const uint32_t st = splz::decode_block(sh, image, zb, tokens_all + (size_t)b * SPL_Z_TOKEN_STRIDE, n);
    if (threadIdx.x == 0) { status[b] = st; n_tok[b] = st == SPL_Z_OK ? n : 0u; }
}

__global__ __launch_bounds__(64) void spl_inflate_copy_kernel(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out_all, uint32_t *status, const uint8_t *tokens_all,
                                                              const uint32_t *n_tok)
{
    __shared__ uint32_t lds[64u * splz::COPY_LANE_BYTES / 4u];
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks) return;
    if (status[b] != SPL_Z_OK) return;
    uint8_t *const mine = (uint8_t *)lds + threadIdx.x * splz::COPY_LANE_BYTES;
    const uint32_t made = splz::copy_block(out_all + blocks[b].out, blocks[b].out_len, tokens_all + (size_t)b * SPL_Z_TOKEN_STRIDE, n_tok[b], mine, mine + splz::RING_BYTES);
    if (made != blocks[b].out_len) status[b] = SPL_Z_SHORT;
}

namespace {

// (pointers into the file image say "global memory" in their type: the image's address goes through an integer to be aligned,
// after which the compiler no longer knows, and a FLAT load counts as an LDS operation too -- every wait for a table look-up
// would then wait for the read-ahead word as well, a trip to memory on the dependent chain of the symbols)
typedef const __attribute__((address_space(1))) uint32_t *gptr32;
typedef const __attribute__((address_space(1))) uint8_t *gptr8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct BitReader {
    gptr32 p;            // the word after `ahead`
    const uint8_t *end;  // one past the block's DEFLATE data
    uint64_t buf;
    uint32_t cnt;        // valid bits in buf
    uint32_t ahead;      // the next word of the image, asked for when the previous one was taken: a refill never waits for memory
    __device__ __forceinline__ void init(const uint8_t *at, const uint8_t *e)
    {
        end = e;
        const uintptr_t a = (uintptr_t)at;
        p = (gptr32)(a & ~(uintptr_t)3);
        const uint32_t skip = (uint32_t)(a & 3u) * 8u;
        buf = (uint64_t)(*p++) >> skip;
        cnt = 32u - skip;
        ahead = *p++;
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32u) { buf |= (uint64_t)ahead << cnt; cnt += 32u; ahead = *p++; }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    // bytes of the block consumed so far (whole bytes still in the buffer, and the word read ahead, given back)
    __device__ __forceinline__ const uint8_t *pos() const { return (const uint8_t *)(p - 1) - (cnt >> 3); }
};

// A canonical Huffman table, decoded without a loop over code lengths: limit[len] = the first 15-bit left-justified code value
// that is NOT a code of `len` bits or fewer (non-decreasing in len, limit[0] = 0), so the length of the code at the head of the
// bit buffer is the number of limits its left-justified value has reached.  Two limits to a word, both compared by ONE
// subtraction: with v < 0x8000 and limits <= 0x8000, (0x8000 + v - limit) has bit 15 set exactly when v >= limit and never
// borrows from its neighbour -- eight subtractions, eight masked population counts.  The symbol's place in the table's array
// is base[len] + (v >> (15 - len)), base[len] = (symbols with shorter codes) - (first code of that length): one value per
// length, picked from eight more registers.
struct Table {
    uint32_t w[8]; // limit[len] = (w[len >> 1] >> (16 * (len & 1))) & 0xffff, len = 0..15
    uint32_t b[8]; // base[len] as int16, packed the same way
};

// The symbols of a lane's tables live in LDS, entry-major (entry e of lane l at [e * 64 + l]): a look-up is a ds_read instead
// of a trip to the lane's scratch -- the look-up sits on the dependent chain of every symbol.  A byte per symbol plus, for the
// literal/length table, a bit per entry for "256 and above": 354 bytes per lane, seven workgroups in a CU's 160 KB.
struct LdsSyms {
    uint8_t *lo;   // &bytes[lane]
    uint32_t *hi;  // &bits[lane] (word w of lane l at [w * 64 + l]) or nullptr for tables whose symbols fit a byte
    __device__ __forceinline__ int get(int e) const
    {
        int v = lo[e * 64];
        if (hi) v |= (int)((hi[(e >> 5) * 64] >> (e & 31)) & 1u) << 8;
        return v;
    }
    __device__ __forceinline__ void set(int e, int v) const
    {
        lo[e * 64] = (uint8_t)v;
        if (hi) {
            uint32_t &w = hi[(e >> 5) * 64];
            w = (w & ~(1u << (e & 31))) | ((uint32_t)(v >> 8) << (e & 31));
        }
    }
};

__device__ __forceinline__ bool build_table(const uint8_t *lengths, int n, Table &c, const LdsSyms &symbol)
{
    uint16_t count[16], offs[16], limit[16], base[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lengths[s]]++;
    count[0] = 0;
    // over-subscribed sets are an error; incomplete ones are legal only in the cases RFC 1951 allows, which the decode
    // handles by finding no code
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
    int code = 0;
    limit[0] = 0;
    base[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        base[l] = (uint16_t)((int)offs[l] - code); // (code = the first code of length l)
        code += (int)count[l];
        limit[l] = (uint16_t)(code << (15 - l)); // (<= 0x8000: the set is not over-subscribed)
        code <<= 1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c.b[k] = (uint32_t)base[2 * k] | ((uint32_t)base[2 * k + 1] << 16);
    for (int s = 0; s < n; ++s)
        if (lengths[s]) symbol.set(offs[lengths[s]]++, s);
#pragma unroll
    for (int k = 0; k < 8; ++k) c.w[k] = (uint32_t)limit[2 * k] | ((uint32_t)limit[2 * k + 1] << 16);
    return true;
}

// One symbol from the bits in the buffer (the caller has refilled it: 15 bits at most).  -1: no code matches (corrupt data,
// or an incomplete table was asked for a code it does not have).
__device__ __forceinline__ int decode_symbol(BitReader &br, const Table &c, const LdsSyms &symbol)
{
    const uint32_t v = __brev((uint32_t)br.buf) >> 17; // the next 15 bits, first bit on top: codes are packed from their top bit
    const uint32_t vv = v * 0x10001u + 0x80008000u;
    uint32_t len = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) len += (uint32_t)__popc((vv - c.w[k]) & 0x80008000u);
    if (len > 15u) return -1; // (v >= limit[15])
    const uint32_t i = len >> 1;
    const uint32_t x0 = (i & 1u) ? c.b[1] : c.b[0], x1 = (i & 1u) ? c.b[3] : c.b[2], x2 = (i & 1u) ? c.b[5] : c.b[4], x3 = (i & 1u) ? c.b[7] : c.b[6];
    const uint32_t y0 = (i & 2u) ? x1 : x0, y1 = (i & 2u) ? x3 : x2;
    const uint32_t z = (i & 4u) ? y1 : y0;
    const int base = (int)(int16_t)(uint16_t)(z >> (16u * (len & 1u)));
    br.drop(len);
    return symbol.get(base + (int)(v >> (15u - len)));
}

// Length symbol 257 + i -> (base, extra bits); distance symbol -> the same (RFC 1951, 3.2.5), by arithmetic: the tables would be
// a trip to LDS on the symbol's dependent chain.
__device__ __forceinline__ void length_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 8u || i == 28u ? 0u : (i >> 2) - 1u;
    base = i < 4u ? 3u + i : (i == 28u ? 258u : 3u + ((4u + (i & 3u)) << extra));
}
__device__ __forceinline__ void distance_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 4u ? 0u : (i >> 1) - 1u;
    base = i < 2u ? 1u + i : 1u + ((2u + (i & 1u)) << extra);
}

__constant__ uint8_t k_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

} // namespace

__global__ __launch_bounds__(64) void spl_inflate_kernel(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out_all, uint32_t *status)
{
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    // lane-interleaved symbol tables: literal/length (288), distance (32); the code-length code's 19 symbols borrow the
    // distance table's place while the lengths are being read
    __shared__ uint8_t s_sym[(288 + 32) * 64];
    __shared__ uint32_t s_hi[9 * 64];
    const LdsSyms lsym{s_sym + threadIdx.x, s_hi + threadIdx.x}, dsym{s_sym + 288 * 64 + threadIdx.x, nullptr};
    if (b >= n_blocks) return;
    const spl_zblock zb = blocks[b];
    uint8_t *const out = out_all + zb.out;
    const uint32_t out_len = zb.out_len;
    uint32_t at = 0; // bytes written
    uint32_t err = SPL_Z_OK;
    if (out_len == 0) { status[b] = SPL_Z_OK; return; } // (the EOF marker and other empty blocks: nothing to decode into)
    BitReader br;
    br.init(image + zb.in, image + zb.in + zb.in_len);
    uint8_t lengths[320];
    Table lc, dc;
    uint64_t window = 0; // the last eight bytes of the output, the most recent one on top
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
            for (uint32_t i = 0; i < len; ++i) { out[at + i] = src[i]; window = (window >> 8) | ((uint64_t)src[i] << 56); }
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
            build_table(lengths, 288, lc, lsym);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            build_table(lengths, 30, dc, dsym);
        } else { // dynamic codes
            br.refill();
            const int nlen = (int)br.take(5) + 257, ndist = (int)br.take(5) + 1, ncode = (int)br.take(4) + 4;
            if (nlen > 286 || ndist > 30) { err = SPL_Z_BAD_LENGTHS; break; }
            uint8_t cl[19];
#pragma unroll
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            // (a header is bounded by the block's own bytes like everything else: the read-ahead runs 12 bytes past what has been
            //  consumed, the image is readable SPL_Z_IMAGE_PAD bytes past any block -- beyond end + 24 the data is corrupt)
            const gptr8 stop_h = (gptr8)(br.end + 24);
            for (int i = 0; i < ncode && (gptr8)br.p <= stop_h; ++i) { br.refill(); cl[k_clen_order[i]] = (uint8_t)br.take(3); }
            if ((gptr8)br.p > stop_h) { err = SPL_Z_OVERRUN; break; }
            Table cc;
            const LdsSyms csym = dsym;
            if (!build_table(cl, 19, cc, csym)) { err = SPL_Z_BAD_LENGTHS; break; }
            int idx = 0;
            while (idx < nlen + ndist) {
                if ((gptr8)br.p > stop_h) { err = SPL_Z_OVERRUN; break; }
                br.refill();
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
            if (!build_table(lengths, nlen, lc, lsym)) { err = SPL_Z_BAD_LENGTHS; break; }
            if (!build_table(lengths + nlen, ndist, dc, dsym)) { err = SPL_Z_BAD_LENGTHS; break; }
        }
        // The symbols of the block, as a state machine that does ONE small thing per turn -- a literal/length symbol, a distance
        // symbol, or a piece of a pending copy.  The 64 lanes of a wave take their turns together: a loop that finished a
        // 258-byte copy before looking at the next symbol would make 63 lanes wait for the longest copy among them at every
        // step (that version ran at a twelfth of this one's speed).
        //
        // A turn has a memory half and a decoding half, and nothing in a turn waits for memory it has asked for itself:
        // what the memory half asks for -- the next word of the block, the source bytes of a copy -- is used by the memory
        // half of the NEXT turn, and what the decoding half produces -- a literal -- is stored by the next turn's memory half.
        // The wave's one wait per turn is then for operations issued a whole symbol decode earlier, instead of a round trip to
        // the L2 after every load and an acknowledged store before every refill (62 % of the kernel's time, by SQ_WAIT_ANY).
        // `window` = the last eight bytes of the output: a copy at a distance of eight or less is made from it, no load at all.
        uint32_t copy_left = 0, copy_dist = 0, want_dist = 0;
        uint32_t lit = 0, n_lit = 0; // literals of the last turn (two at most), not yet stored
        bool eob = false, loaded = false;
        uint32_t wide = 0; // what has been asked for: 0 = eight bytes, 1 = 32, 2 = 64
        uint64_t w0 = 0; // eight source bytes of a copy, or
        u32x4 wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0}, wc = {0, 0, 0, 0}, wd = {0, 0, 0, 0}; // thirty-two (wa, wb) or sixty-four
        const gptr8 stop = (gptr8)(br.end + 24); // (the read-ahead runs 12 bytes past what has been consumed: beyond this, the data is corrupt)
        for (uint32_t turns = 0;; ++turns) {
            // the turn's one wait for memory: everything the previous turn asked for, asked for before its decoding half.
            // (Said to the compiler as a use of all of it, here: left to itself it waits where each value is first touched, for
            // everything in flight at that point -- this turn's stores and loads included.)
            asm volatile("" : "+v"(w0), "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd), "+v"(br.ahead));
            if (turns > 2u * out_len + 4096u || (gptr8)br.p > stop) { err = SPL_Z_OVERRUN; break; }
            br.refill(); // (33 bits or more after this: a symbol and its extra bits are 28 at most)
            // ---- the memory half: stores first (they may be what the loads after them read), then the loads
            if (n_lit) {
                if (n_lit == 2u) { const uint16_t two = (uint16_t)lit; __builtin_memcpy(out + at - 2u, &two, 2); }
                else out[at - 1u] = (uint8_t)lit;
                n_lit = 0;
            }
            if (eob) break; // (seen behind a literal in the last turn)
            if (copy_left) {
                uint64_t bytes = 0;
                bool narrow = false;
                if (copy_dist <= 8u) {
                    // the next bytes repeat the last copy_dist ones: double the pattern until it covers eight bytes
                    uint64_t rep = copy_dist == 8u ? window : (window >> (8u * (8u - copy_dist)));
                    if (copy_dist < 8u) rep &= (1ull << (8u * copy_dist)) - 1ull;
                    if (copy_dist < 2u) rep |= rep << 8;
                    if (copy_dist < 3u) rep |= rep << 16;
                    else if (copy_dist == 3u) rep |= rep << 24;
                    if (copy_dist < 5u) { if (copy_dist == 3u) rep |= rep << 48; else rep |= rep << 32; }
                    else if (copy_dist < 8u) rep |= rep << (8u * copy_dist);
                    bytes = rep;
                    narrow = true;
                } else if (loaded) {
                    if (wide == 2u) {
                        __builtin_memcpy(out + at, &wa, 16);
                        __builtin_memcpy(out + at + 16, &wb, 16);
                        __builtin_memcpy(out + at + 32, &wc, 16);
                        __builtin_memcpy(out + at + 48, &wd, 16);
                        window = (uint64_t)wd.z | ((uint64_t)wd.w << 32);
                        at += 64u;
                        copy_left -= 64u;
                    } else if (wide == 1u) {
                        __builtin_memcpy(out + at, &wa, 16);
                        __builtin_memcpy(out + at + 16, &wb, 16);
                        window = (uint64_t)wb.z | ((uint64_t)wb.w << 32);
                        at += 32u;
                        copy_left -= 32u;
                    } else {
                        bytes = w0;
                        narrow = true;
                    }
                    loaded = false;
                }
                if (narrow) {
                    // eight bytes at out + at, the first n of them meant: stored whole when the ones beyond are this lane's to
                    // overwrite later (all but the last seven bytes of a block)
                    const uint32_t n = copy_left < 8u ? copy_left : 8u;
                    if (at + 8u <= out_len) {
                        __builtin_memcpy(out + at, &bytes, 8);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if ((uint32_t)k < n) out[at + (uint32_t)k] = (uint8_t)(bytes >> (8 * k));
                    }
                    window = n == 8u ? bytes : ((window >> (8u * n)) | (bytes << (8u * (8u - n))));
                    at += n;
                    copy_left -= n;
                }
                if (copy_left && copy_dist > 8u) {
                    // the next piece's source: behind `at` in full (the distance is more than its length), stored already
                    const uint8_t *src = out + at - copy_dist;
                    wide = copy_dist >= 64u && copy_left >= 64u ? 2u : (copy_dist >= 32u && copy_left >= 32u ? 1u : 0u);
                    if (wide) {
                        __builtin_memcpy(&wa, src, 16);
                        __builtin_memcpy(&wb, src + 16, 16);
                        if (wide == 2u) {
                            __builtin_memcpy(&wc, src + 32, 16);
                            __builtin_memcpy(&wd, src + 48, 16);
                        }
                    } else {
                        __builtin_memcpy(&w0, src, 8);
                    }
                    loaded = true;
                }
            }
            if (copy_left) continue;
            // ---- the decoding half.  One decode per turn whichever table the lane is at (a length is followed by a distance):
            // the lanes that want a distance symbol and those that want a literal/length symbol go through the same
            // instructions with their own limits, bases and symbol arrays -- two decode blocks in a row cost every turn twice
            const bool is_dist = want_dist != 0u;
            Table tc;
#pragma unroll
            for (int k = 0; k < 8; ++k) { tc.w[k] = is_dist ? dc.w[k] : lc.w[k]; tc.b[k] = is_dist ? dc.b[k] : lc.b[k]; }
            const LdsSyms ts{is_dist ? dsym.lo : lsym.lo, lsym.hi};
            int sym = decode_symbol(br, tc, ts);
            if (sym < 0) { err = SPL_Z_BAD_CODE; break; }
            if (is_dist) {
                const uint32_t ds = (uint32_t)sym & 0xffu; // (the high bit belongs to the literal/length table: meaningless here)
                if (ds >= 30u) { err = SPL_Z_BAD_CODE; break; }
                uint32_t base, extra;
                distance_code(ds, base, extra);
                copy_dist = base + br.take(extra);
                if (copy_dist > at) { err = SPL_Z_BAD_DISTANCE; break; }
                copy_left = want_dist;
                if (at + copy_left > out_len) { err = SPL_Z_OVERRUN; break; }
                want_dist = 0;
                continue;
            }
            if (sym < 256) {
                if (at >= out_len) { err = SPL_Z_OVERRUN; break; }
                lit = (uint32_t)sym;
                n_lit = 1; // (stored by the next turn, behind `at`)
                ++at;
                window = (window >> 8) | ((uint64_t)sym << 56);
                // a second symbol in the same turn when the bits are there (a code and a length's extra bits: 20 at most):
                // most symbols are literals, and the turn's memory half, its wait and its bookkeeping are then paid once for two
                if (br.cnt >= 20u) {
                    const int s2 = decode_symbol(br, lc, lsym);
                    if (s2 < 0) { err = SPL_Z_BAD_CODE; break; }
                    if (s2 < 256) {
                        if (at >= out_len) { err = SPL_Z_OVERRUN; break; }
                        lit |= (uint32_t)s2 << 8;
                        n_lit = 2;
                        ++at;
                        window = (window >> 8) | ((uint64_t)s2 << 56);
                    } else if (s2 == 256) {
                        eob = true; // (the literal is still to be stored: the next turn does that and leaves)
                    } else {
                        const uint32_t l2 = (uint32_t)s2 - 257u;
                        if (l2 >= 29u) { err = SPL_Z_BAD_CODE; break; }
                        uint32_t base, extra;
                        length_code(l2, base, extra);
                        want_dist = base + br.take(extra);
                    }
                }
                continue;
            }
            if (sym == 256) break;
            const uint32_t ls = (uint32_t)sym - 257u;
            if (ls >= 29u) { err = SPL_Z_BAD_CODE; break; }
            uint32_t base, extra;
            length_code(ls, base, extra);
            want_dist = base + br.take(extra); // (the length, 3..258: a distance symbol follows)
        }
    }
    if (err == SPL_Z_OK && at != out_len) err = SPL_Z_SHORT;
    status[b] = err;
}

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
extern "C" size_t spl_dev_inflate_work_bytes(uint32_t n_blocks) { return 256 + tokens_at(n_blocks) + (size_t)n_blocks * SPL_Z_TOKEN_STRIDE; }

extern "C" int spl_dev_launch_inflate_decode(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *work, void *stream)
{
    if (n_blocks == 0 || !work) return 0;
    hipLaunchKernelGGL(spl_inflate_decode_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, image, blocks, n_blocks, status, (uint8_t *)work + tokens_at(n_blocks), (uint32_t *)work);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_inflate_copy(const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream)
{
    if (n_blocks == 0 || !work) return 0;
    hipLaunchKernelGGL(spl_inflate_copy_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, blocks, n_blocks, out, status, (const uint8_t *)work + tokens_at(n_blocks),
                       (const uint32_t *)work);
    return (int)hipGetLastError();
}

// (round 2's decoder needs the file image where the others need the tokens: callers that may get either pass it here)
extern "C" int spl_dev_launch_inflate(const uint8_t *image, const spl_zblock *blocks, uint32_t n_blocks, uint8_t *out, uint32_t *status, void *work, void *stream)
{
    if (n_blocks == 0) return 0;
    if (!work) {
        hipLaunchKernelGGL(spl_inflate_kernel, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, image, blocks, n_blocks, out, status);
        return (int)hipGetLastError();
    }
    const int rc = spl_dev_launch_inflate_decode(image, blocks, n_blocks, status, work, stream);
    return rc ? rc : spl_dev_launch_inflate_copy(blocks, n_blocks, out, status, work, stream);
}

extern "C" int spl_dev_launch_crc32(const uint8_t *out, const spl_zblock *blocks, uint32_t n_blocks, uint32_t *status, void *stream)
{
    if (n_blocks == 0) return 0;
    // (four streams a lane: 1.6 ms a window of 43 169 blocks against 2.0 with one and 1.5 with two; 3.5 with eight, whose 64 x 8 lines
    //  the L1 cannot hold -- profiles/r04v_crc_streams.txt)
    hipLaunchKernelGGL(spl_crc32_kernel<4>, dim3((n_blocks + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, out, blocks, n_blocks, status);
    return (int)hipGetLastError();
}
