// spl_inflate_wave.h -- DEFLATE (RFC 1951) on the device in two kernels (spl_inflate.hip launches them; the bodies are here,
// written against the primitives of spl_wave.h so that the same source runs on the host under tests/hostsim/wave_emul.h):
//
//   decode_block   one BGZF block per WAVE: the Huffman decoding.  What the block's symbols say is written down as a STREAM OF
//                  TOKENS, bytes: a run of n literals is n - 1 (0..127) and the n bytes; a match is three bytes,
//                  0x80 | L & 0x7f, L >> 7 | (D & 0x7f) << 1, D >> 7 with L = length - 3, D = distance - 1.  Nothing of the
//                  inflated stream is written here.
//   copy_block     one BGZF block per LANE: the token stream read front to back and the block's bytes made from it -- literals
//                  copied, matches copied from what is there by then -- through a small window of the newest output in LDS
//                  (RING bytes a lane): whole 16-byte pieces of the output go to memory once, a match's source is read from the
//                  window when it lies that close and from memory otherwise -- asked for AHEAD: the next piece of a match that
//                  goes on a turn before it is needed, the first piece of the next far match by a second reader that walks the
//                  tokens in front of the first --, the token stream comes in 64 bytes at a time on a fixed beat, asked for a
//                  beat before it is needed.  What this is made to avoid: the first version kept lists of matches and copied
//                  in place, three to four scattered 16-byte accesses per match and 9 000 matches a block (22 ms per 49 152
//                  blocks); this one stores every output byte once and loads every token byte once (10.5 ms).  The window is
//                  small on purpose: with 512 bytes a lane (53 KB a wave) the kernel is another millisecond faster alone, and
//                  its waves leave the decoding kernel, which runs beside it, no LDS to be resident with
//                  (profiles/r03_inflate_wave_account.md).
//
// Why the split.  Round 2 gave every lane a block of its own for everything: a lane's Huffman tables were 356 bytes of LDS and
// thirty-odd registers (1.75 waves per SIMD), every turn of the 64 lanes ran through all the decoder's branches (330 vector
// instructions a turn).  Decoding is what a wave can share: here 64 lanes use ONE pair of tables (two-level look-up tables in
// LDS, 2.9 KB) and read the compressed bytes from a tile of them staged in LDS by coalesced loads.  Copying is what it cannot: a
// BAM record's matches copy from the record before, whose matches copy from the one before that -- measured with
// tools/inflate_sim.cpp: about 2 000 dependent steps per block, a dozen copies ready at any time -- so 64 lanes on one block's
// copies idle (built and measured: 33 of the kernel's 58 ms, same account), while 64 lanes on 64 blocks'
// copies are all busy and need no tables.
//
// Huffman codes have no markers, so where a lane should start decoding is not known: it is FOUND.  The data of a DEFLATE block
// is worked off in tiles of 64 subsequences of SUB_BITS bits, one per lane.  Every lane decodes from a guessed start (the beginning
// of its subsequence; lane 0 from the true position) to the first symbol boundary at or past its subsequence's end, counting what
// it would produce; then every lane takes its predecessor's end as its start and decodes again if that differs, until nothing
// changes.  Wrongly started decoders fall into step with the true sequence of symbols sooner or later, so a few passes do; lane k
// is right after pass k + 1 whatever the data.  A prefix sum over the lanes' byte counts places every lane in the output, and a last
// pass decodes once more for good.  A turn of any pass takes one symbol, or two when both are literals (decode).
//
// Replaces what SpliSER_v0_1_8.py:422 (samtools view) does to every BGZF block it touches.
#ifndef SPL_INFLATE_WAVE_H
#define SPL_INFLATE_WAVE_H

#include "spl_inflate.h"

namespace splz {

constexpr uint32_t ROOT_L = 9, ROOT_D = 6, ROOT_C = 7;
constexpr uint32_t LUT_L = 852, LUT_D = 592; // entries: root table + the most sub-tables a valid code can need (zlib's ENOUGH_LENS / ENOUGH_DISTS for these roots)
#ifndef SPLZ_SUB_BITS
#define SPLZ_SUB_BITS 256
#endif
#ifndef SPLZ_TOKCAP
#define SPLZ_TOKCAP 5120
#endif
constexpr uint32_t SUB_BITS = SPLZ_SUB_BITS; // bits of DEFLATE data per lane and tile (256; 512 measured: profiles/r04ae_k1_occupancy.txt)
constexpr uint32_t TILE_WORDS = 64u * SUB_BITS / 32u;
constexpr uint32_t TILE_PAD = 16;            // words behind the tile: a symbol that begins in the last subsequence ends there
constexpr uint32_t TOKCAP = SPLZ_TOKCAP;     // bytes of token stream per tile (a tile with more is cut short)
#ifndef SPLZ_TOKCAP_SMALL
#define SPLZ_TOKCAP_SMALL 3072
#endif
constexpr uint32_t TOKCAP_SMALL = SPLZ_TOKCAP_SMALL; // ... in the kernel for blocks that deflate well (spl_inflate.hip: five waves a SIMD instead of four)
constexpr uint32_t FL_OK = 0, FL_EOB = 1, FL_ERR = 2;
#ifndef SPLZ_EMIT_ROUNDS
#define SPLZ_EMIT_ROUNDS 2
#endif
constexpr uint32_t EMIT_ROUNDS = SPLZ_EMIT_ROUNDS; // rounds of correction in which a corrected lane writes its tokens down at once (decode_block)
constexpr uint32_t OPT_WRITING_PASS = 1; // decode_block: every tile's tokens by a writing pass of their own (rounds 3-5's way; A/B and tests)
constexpr uint32_t SYM_EOB = 256, SYM_MATCH = 257, SYM_BAD = 0xffffffffu;

// One wave's shared memory: 10120 bytes (16 waves on a CU's 160 KB).
struct Shared {
    uint16_t lut_l[LUT_L];                // literal/length code, entries that carry what a symbol MEANS (leaf_l below); 0 = no such code
    uint16_t lut_d[LUT_D];                // distance code (leaf_d)
    uint32_t tile[TILE_WORDS + TILE_PAD]; // the compressed bytes being worked on (while tables are built: work space)
    uint32_t tok[TOKCAP / 4];             // the tile's stretch of the token stream (while a header is read: code lengths, the code-length code's table)
};
static_assert(SUB_BITS % 256u == 0u, "the tile is staged thirty-two bytes a lane and round");
static_assert(TOKCAP >= 352u + 256u + 64u, "code lengths and the code-length code's table lie in tok while a header is read");

// the order in which a dynamic header lists the lengths of the code-length code (RFC 1951, 3.2.7), five bits a place
constexpr uint64_t pack5(const int *v, int n) { uint64_t r = 0; for (int i = 0; i < n; ++i) r |= (uint64_t)v[i] << (5 * i); return r; }
constexpr int k_clen_a[12] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4}, k_clen_b[7] = {12, 3, 13, 2, 14, 1, 15};
constexpr uint64_t k_clen_lo = pack5(k_clen_a, 12), k_clen_hi = pack5(k_clen_b, 7);
WV_DEV uint32_t clen_order(uint32_t i) { return (uint32_t)((i < 12u ? k_clen_lo >> (5u * i) : k_clen_hi >> (5u * (i - 12u))) & 31u); }

WV_DEV uint32_t bitrev(uint32_t v, uint32_t n) { return wv::brev32(v) >> (32u - n); }

// 32 bits of the block's data from bit `pos`, straight from memory (headers: every lane asks for the same bytes)
WV_DEV uint32_t gbits(const uint8_t *in, uint32_t pos) { return (uint32_t)(wv::ld64(in + (pos >> 3)) >> (pos & 7u)); }

// length symbol 257 + i -> (base, extra bits); distance symbol -> the same (RFC 1951, 3.2.5), by arithmetic
WV_DEV void length_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 8u || i == 28u ? 0u : (i >> 2) - 1u;
    base = i < 4u ? 3u + i : (i == 28u ? 258u : 3u + ((4u + (i & 3u)) << extra));
}
WV_DEV void distance_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 4u ? 0u : (i >> 1) - 1u;
    base = i < 2u ? 1u + i : 1u + ((2u + (i & 1u)) << extra);
}

// What a table entry says, twelve bits above the four that hold the code's length (an entry of 0: no such code).  The decoder's
// turn is a chain of dependent vector instructions that every lane of the wave runs through, whichever symbol its own lane
// has: what a symbol MEANS is worked out once per table, not once per decoded symbol (18 of a turn's 70 instructions were the
// arithmetic of length_code / distance_code).
//   KIND_RAW   the symbol itself (the code-length code)
//   KIND_LIT   literal or end of block: the symbol, 0..256.  Length symbol 257 + i: 0x800 | extra << 8 | base - 3 (base - 3 is
//              0..255).  286 and 287 have no entry.
//   KIND_DIST  distance symbol i: extra << 2 | m with base = 1 + (m << extra), m = i for i < 4, 2 + (i & 1) beyond.  30 and 31
//              have no entry.
// A root entry that stands for a sub-table: KIND_LIT has no bit to spare for a flag, so it is the entry whose length field
// is 0 and which is not 0: (offset - root entries) << 7 | sub-table bits << 4 (1..6, 9 bits of offset: LUT_L - 512 < 512).  The
// others: 0x8000 | offset << 4 | sub-table bits.
constexpr int KIND_RAW = 0, KIND_LIT = 1, KIND_DIST = 2;
template <int KIND>
WV_DEV uint32_t leaf(uint32_t s)
{
    if (KIND == KIND_LIT) {
        if (s <= 256u) return s;
        if (s > 285u) return 0xfffu; // (marks "no entry": see build_lut)
        uint32_t b, x;
        length_code(s - 257u, b, x);
        return 0x800u | x << 8 | (b - 3u);
    }
    if (KIND == KIND_DIST) {
        if (s >= 30u) return 0xfffu;
        const uint32_t x = s < 4u ? 0u : (s >> 1) - 1u, m = s < 4u ? s : 2u + (s & 1u);
        return x << 2 | m;
    }
    return s;
}

// Canonical code -> look-up table, by the whole wave.  lens[0..n): code lengths (bytes in shared memory); work: 1 << root words.
// false: the lengths over-subscribe the code space, or need more sub-tables than any valid code does.
template <int KIND>
WV_DEV bool build_lut(const uint8_t *lens, uint32_t n, uint32_t root, uint16_t *lut, uint32_t cap, uint32_t *work)
{
    const uint32_t l = wv::lane();
    const uint64_t below = (1ull << l) - 1ull;
    uint32_t cnt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cnt[k] = 0;
    uint32_t L[5], rank[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const uint32_t s = (uint32_t)c * 64u + l;
        L[c] = s < n ? lens[s] : 0u;
        rank[c] = 0;
        if ((uint32_t)c * 64u < n) {
#pragma unroll
            for (uint32_t len = 1; len < 16u; ++len) {
                const uint64_t m = wv::ballot(L[c] == len);
                if (L[c] == len) rank[c] = cnt[len] + wv::popc64(m & below);
                cnt[len] += wv::popc64(m);
            }
        }
    }
    int left = 1;
    uint32_t first[16], code = 0;
    first[0] = 0;
    bool over = false;
#pragma unroll
    for (uint32_t len = 1; len < 16u; ++len) {
        left = (left << 1) - (int)cnt[len];
        over = over || left < 0;
        first[len] = code;
        code = (code + cnt[len]) << 1;
    }
    if (over) return false;
    uint32_t cd[5]; // the symbols' codes (first code of the length + rank among the symbols of that length), top bit first
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        cd[c] = 0;
#pragma unroll
        for (uint32_t len = 1; len < 16u; ++len)
            if (L[c] == len) cd[c] = first[len] + rank[c];
    }
    const uint32_t n_root = 1u << root;
    for (uint32_t i = l; i < n_root; i += 64u) { lut[i] = 0; work[i] = 0; }
    wv::sync();
    // codes longer than the root share a sub-table with the codes that begin with the same `root` bits: its size by the longest
#pragma unroll
    for (int c = 0; c < 5; ++c)
        if (L[c] > root) wv::lds_max(&work[cd[c] >> (L[c] - root)], L[c] - root);
    wv::sync();
    const uint32_t per = n_root >= 64u ? n_root / 64u : 1u;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t P = l * per + j;
        const uint32_t b = P < n_root ? work[P] : 0u;
        mine += b ? 1u << b : 0u;
    }
    const uint32_t incl = wv::scan_add(mine);
    if (n_root + wv::readlane(incl, 63u) > cap) return false;
    uint32_t off = n_root + incl - mine;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t P = l * per + j;
        const uint32_t b = P < n_root ? work[P] : 0u;
        if (b) {
            work[P] = off << 4 | b;
            lut[bitrev(P, root)] = KIND == KIND_LIT ? (uint16_t)((off - n_root) << 7 | b << 4) : (uint16_t)(0x8000u | off << 4 | b);
            for (uint32_t k = 0; k < (1u << b); ++k) lut[off + k] = 0;
            off += 1u << b;
        }
    }
    wv::sync();
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const uint32_t len = L[c], s = (uint32_t)c * 64u + l;
        if (!len) continue;
        const uint32_t what = leaf<KIND>(s);
        if (len <= root) {
            const uint16_t e = what == 0xfffu ? (uint16_t)0 : (uint16_t)(what << 4 | len); // (a code for a symbol that does not exist decodes to "no such code")
            for (uint32_t k = bitrev(cd[c], len); k < n_root; k += 1u << len) lut[k] = e;
        } else {
            const uint32_t e = work[cd[c] >> (len - root)], sub = e >> 4, b = e & 15u, rest = len - root;
            const uint16_t e2 = what == 0xfffu ? (uint16_t)0 : (uint16_t)(what << 4 | rest);
            for (uint32_t k = bitrev(cd[c] & ((1u << rest) - 1u), rest); k < (1u << b); k += 1u << rest) lut[sub + k] = e2;
        }
    }
    wv::sync();
    return true;
}

// 32 bits of the tile from bit `pos` of the block's data (the tile begins at bit `base`)
// (Measured and not kept, round 4: the lane's next 64 bits held in registers and topped up from a word asked for a turn ahead,
// so that the tile's words are off the turn's chain of dependent LDS trips -- 20 vector instructions more per turn for the 64-bit
// shifts and the conditional top-up, and not a microsecond less: the kernel is bound by instruction issue, not by that chain.
// profiles/r04l_k1_reader_in_registers_q*.txt)
WV_DEV uint32_t tbits(const uint32_t *tile, uint32_t base, uint32_t pos)
{
    const uint32_t rel = pos - base, w = rel >> 5;
    const uint64_t two = (uint64_t)tile[w] | (uint64_t)tile[w + 1u] << 32;
    return (uint32_t)(two >> (rel & 31u));
}

// One symbol at bit `pos` of the tile: a literal (its value), SYM_EOB, SYM_MATCH (len, dist set) or SYM_BAD.  `pos` moves past it.
// Behind a literal that ends before `stop` a SECOND literal is taken from the bits already at hand (17 at least) if a literal
// with a code of the root table is what follows: lit2 is its value, or NO_LIT2.  A wave's turn costs what its slowest path costs,
// and the lane with the most turns in a tile is the one whose subsequence is all short literal codes: two to a turn, it has half of them.
constexpr uint32_t NO_LIT2 = 0xffffffffu;
WV_DEV uint32_t decode(const Shared &sh, uint32_t base, uint32_t &pos, uint32_t &len, uint32_t &dist, uint32_t stop, uint32_t &lit2)
{
    uint32_t w = tbits(sh.tile, base, pos), used = 0;
    uint32_t e = sh.lut_l[w & ((1u << ROOT_L) - 1u)];
    lit2 = NO_LIT2;
    if ((e & 15u) == 0u) { // a sub-table, or no such code
        if (e == 0u) return SYM_BAD;
        used = ROOT_L;
        e = sh.lut_l[(1u << ROOT_L) + (e >> 7) + ((w >> ROOT_L) & ((1u << ((e >> 4) & 7u)) - 1u))];
        if (e == 0u) return SYM_BAD;
    }
    used += e & 15u;
    if (!(e & 0x8000u)) { // a literal or the end of the block
        const uint32_t sym = e >> 4;
        pos += used;
        if (sym < 256u && pos < stop) { // a second literal, if its code is one of the root table's (9 bits at most of the 17 and more at hand)
            const uint32_t e2 = sh.lut_l[(w >> used) & ((1u << ROOT_L) - 1u)];
            if ((e2 & 15u) != 0u && e2 < (256u << 4)) { lit2 = e2 >> 4; pos += e2 & 15u; }
        }
        return sym;
    }
    const uint32_t xl = (e >> 12) & 7u;
    len = 3u + ((e >> 4) & 255u) + ((w >> used) & ((1u << xl) - 1u)); // (a code and its extra bits: 20 at most)
    pos += used + xl;
    w = tbits(sh.tile, base, pos);
    e = sh.lut_d[w & ((1u << ROOT_D) - 1u)];
    used = 0;
    if (e & 0x8000u) {
        used = ROOT_D;
        e = sh.lut_d[((e >> 4) & 0x3ffu) + ((w >> ROOT_D) & ((1u << (e & 15u)) - 1u))];
    }
    if ((e & 15u) == 0u) return SYM_BAD;
    used += e & 15u;
    const uint32_t xd = (e >> 6) & 15u;
    dist = 1u + (((e >> 4) & 3u) << xd) + ((w >> used) & ((1u << xd) - 1u)); // (28 at most)
    pos += used + xd;
    return SYM_MATCH;
}


// What a lane would produce from `start` to the first symbol boundary at or past `sub_end`: bytes of output, bytes of token stream
// (a lane's literal runs are its own: a run never goes on in the next lane's tokens).
struct Count { uint32_t end, n_out, n_tok, flag; };
WV_DEV Count count_from(const Shared &sh, uint32_t base, uint32_t start, uint32_t sub_end)
{
    // tokens = literals + a length byte for every 128 (or fewer) of a run + three bytes a match: the runs' length bytes are
    // counted where a run ENDS (a turn that takes literals then only adds to two sums)
    Count c{start, 0, 0, FL_OK};
    uint32_t run = 0, n_lit = 0; // literals in the run that is open; literals so far
    while (c.end < sub_end) {
        uint32_t len = 0, dist = 0, lit2;
        const uint32_t s = decode(sh, base, c.end, len, dist, sub_end, lit2);
        if (s < 256u) {
            const uint32_t k = lit2 != NO_LIT2 ? 2u : 1u;
            run += k;
            n_lit += k;
            continue;
        }
        c.n_tok += (run + 127u) >> 7;
        run = 0;
        if (s == SYM_MATCH) { c.n_out += len; c.n_tok += 3u; continue; }
        c.flag = s == SYM_EOB ? FL_EOB : FL_ERR;
        break;
    }
    c.n_tok += ((run + 127u) >> 7) + n_lit;
    c.n_out += n_lit;
    return c;
}

// The same walk WITH its tokens: written to tok[0 .. cap) as the writing pass of decode_block writes them (a lane's literal runs
// are its own), counted like count_from counts them.  fits: all of them found room (the counts are right either way).  need: the
// largest (distance - bytes this lane had made before the match) of its matches: the match reaches back past the block's first
// byte if that exceeds the output position the lane starts at, which is known only when all lanes' counts are (decode_block).
// Why: a tile's symbols were decoded once more after the last count, by a pass of all lanes that did nothing but write down what
// the count had seen -- one pass of the tile's four or five.  Here the LAST count is that pass.
WV_DEV Count emit_from(const Shared &sh, uint32_t base, uint32_t start, uint32_t sub_end, uint8_t *tok, uint32_t cap, uint32_t &need, bool &fits)
{
    Count c{start, 0, 0, FL_OK};
    uint32_t tp = 0, run = 0, hdr = 0, wr = 0;
    need = 0;
    auto put = [&](uint32_t at, uint32_t v) { if (at < cap) tok[at] = (uint8_t)v; };
    while (c.end < sub_end) {
        uint32_t len = 0, dist = 0, lit2;
        const uint32_t s = decode(sh, base, c.end, len, dist, sub_end, lit2);
        if (s < 256u) {
            if (run == 0u || run == 128u) { if (run) put(hdr, 127u); hdr = tp++; run = 0; }
            put(tp++, s);
            ++run; ++wr;
            if (lit2 != NO_LIT2) {
                if (run == 128u) { put(hdr, 127u); hdr = tp++; run = 0; }
                put(tp++, lit2);
                ++run; ++wr;
            }
            continue;
        }
        if (run) { put(hdr, run - 1u); run = 0; }
        if (s != SYM_MATCH) { c.flag = s == SYM_EOB ? FL_EOB : FL_ERR; break; }
        if (dist > wr && dist - wr > need) need = dist - wr;
        const uint32_t L = len - 3u, D = dist - 1u;
        put(tp, 0x80u | (L & 0x7fu)); put(tp + 1u, L >> 7 | (D & 0x7fu) << 1); put(tp + 2u, D >> 7);
        tp += 3u;
        wr += len;
    }
    if (run) put(hdr, run - 1u);
    c.n_tok = tp;
    c.n_out = wr;
    fits = tp <= cap;
    return c;
}

// n bytes (1..16) of (lo, hi) to p
WV_DEV void store_n(uint8_t *p, uint64_t lo, uint64_t hi, uint32_t n)
{
    if (n == 16u) { wv::st128(p, lo, hi); return; }
    if (n & 8u) { wv::st64(p, lo); p += 8; lo = hi; }
    if (n & 4u) { wv::st32(p, (uint32_t)lo); p += 4; lo >>= 32; }
    if (n & 2u) { wv::st16(p, (uint32_t)lo); p += 2; lo >>= 16; }
    if (n & 1u) *p = (uint8_t)lo;
}

// The block `zb` of the file image, by one wave.  stream: room for tok_cap bytes of tokens (16-byte aligned; SPL_Z_TOKEN_STRIDE
// holds any block's, a caller that gives less gets SPL_Z_OVERRUN for a block that needs more); n_tok_out: how many were written.
// opts: OPT_* bits.  Returns the block's status (every lane the same).
// CAP: the bytes of sh.tok the tiles may use (TOKCAP: all of it; TOKCAP_SMALL: what a wave of the denser kernel has, whose shared
// memory ends that much earlier -- tok is Shared's last member).  Same tokens whatever CAP: a tile with more is cut short.
template <uint32_t CAP = TOKCAP>
WV_DEV uint32_t decode_block(Shared &sh, const uint8_t *image, const spl_zblock &zb, uint8_t *stream, uint32_t &n_tok_out, uint32_t tok_cap = SPL_Z_TOKEN_STRIDE,
                             uint32_t opts = 0)
{
    static_assert(CAP <= TOKCAP && CAP >= 352u + 256u + 64u && CAP % 16u == 0u, "the tiles' share of Shared::tok");
    n_tok_out = 0;
    uint32_t n_tok = 0; // bytes of token stream so far
    const uint32_t l = wv::lane();
    const uint8_t *const in = image + zb.in;
    const uint32_t in_len = zb.in_len, out_len = zb.out_len, end_bits = in_len * 8u;
    if (out_len == 0u) return SPL_Z_OK; // (the EOF marker and other empty blocks: nothing to decode into)
    if (out_len > 65536u || in_len > 65536u) return SPL_Z_OVERRUN; // (not a BGZF block)
    uint8_t *const lens = (uint8_t *)sh.tok; // 352 code lengths while a header is read
    uint32_t *const work = sh.tile;
    uint32_t pos = 0, at = 0;
    for (uint32_t last = 0; !last;) {
        if (pos + 3u > end_bits) return SPL_Z_OVERRUN;
        uint32_t w = gbits(in, pos);
        last = w & 1u;
        const uint32_t type = (w >> 1) & 3u;
        pos += 3u;
        if (type == 3u) return SPL_Z_BAD_BLOCK_TYPE;
        if (type == 0u) { // stored: to a byte boundary, LEN, NLEN, the bytes
            const uint32_t byte = (pos + 7u) >> 3;
            if (byte + 4u > in_len) return SPL_Z_OVERRUN;
            const uint32_t ln = wv::ld32(in + byte);
            const uint32_t len = ln & 0xffffu;
            if ((len ^ 0xffffu) != ln >> 16) return SPL_Z_BAD_STORED;
            if (byte + 4u + len > in_len || at + len > out_len) return SPL_Z_OVERRUN;
            const uint8_t *src = in + byte + 4u;
            // as literal runs of 128 (the last one shorter): run r is stream bytes [129 r, 129 r + 129)
            const uint32_t n_runs = (len + 127u) / 128u;
            // (the same room the Huffman path leaves: sections of both kinds in one block can ask for more tokens than any BAM
            // writer's block does -- lone literals between 3-byte matches, then one-byte stored sections -- and what follows this
            // block's room is the next block's)
            if (n_tok + len + n_runs > tok_cap - 64u) return SPL_Z_TOKENS;
            for (uint32_t r = l; r < n_runs; r += 64u) {
                const uint32_t n = len - 128u * r < 128u ? len - 128u * r : 128u;
                uint8_t *t = stream + n_tok + 129u * r;
                t[0] = (uint8_t)(n - 1u);
                for (uint32_t i = 0; i < n; ++i) t[1u + i] = src[128u * r + i];
            }
            n_tok += len + n_runs;
            at += len;
            pos = (byte + 4u + len) * 8u;
            continue;
        }
        uint32_t n_lit, n_dist;
        if (type == 1u) { // the fixed code
            for (uint32_t s = l; s < 320u; s += 64u) lens[s] = s < 144u ? 8 : (s < 256u ? 9 : (s < 280u ? 7 : (s < 288u ? 8 : 5)));
            n_lit = 288u; n_dist = 30u;
            wv::sync();
        } else {
            if (pos + 14u > end_bits) return SPL_Z_OVERRUN;
            w = gbits(in, pos);
            n_lit = (w & 31u) + 257u; n_dist = ((w >> 5) & 31u) + 1u;
            const uint32_t n_code = ((w >> 10) & 15u) + 4u;
            pos += 14u;
            if (n_lit > 286u || n_dist > 30u) return SPL_Z_BAD_LENGTHS;
            if (pos + 3u * n_code > end_bits) return SPL_Z_OVERRUN;
            // the code-length code: its own lengths three bits each, lane by lane
            if (l < 19u) lens[l] = 0;
            wv::sync();
            if (l < n_code) lens[clen_order(l)] = (uint8_t)(gbits(in, pos + 3u * l) & 7u);
            pos += 3u * n_code;
            wv::sync();
            uint16_t *const lut_c = (uint16_t *)sh.tok + 176;
            if (!build_lut<KIND_RAW>(lens, 19u, ROOT_C, lut_c, 1u << ROOT_C, work)) return SPL_Z_BAD_LENGTHS;
            // the lengths of the two codes, a run-length code of its own: one after the other (every lane does the same)
            const uint32_t n_all = n_lit + n_dist;
            uint32_t idx = 0, err = SPL_Z_OK;
            while (idx < n_all) {
                if (pos > end_bits) { err = SPL_Z_OVERRUN; break; }
                w = gbits(in, pos);
                const uint32_t e = lut_c[w & ((1u << ROOT_C) - 1u)], nb = e & 15u, sym = e >> 4;
                if (nb == 0u) { err = SPL_Z_BAD_CODE; break; }
                pos += nb;
                w >>= nb;
                if (sym < 16u) { lens[32u + idx++] = (uint8_t)sym; continue; }
                uint32_t prev = 0, rep;
                if (sym == 16u) {
                    if (idx == 0u) { err = SPL_Z_BAD_LENGTHS; break; }
                    prev = lens[32u + idx - 1u];
                    rep = 3u + (w & 3u); pos += 2u;
                } else if (sym == 17u) {
                    rep = 3u + (w & 7u); pos += 3u;
                } else {
                    rep = 11u + (w & 127u); pos += 7u;
                }
                if (idx + rep > n_all) { err = SPL_Z_BAD_LENGTHS; break; }
                for (uint32_t k = 0; k < rep; ++k) lens[32u + idx + k] = (uint8_t)prev;
                idx += rep;
            }
            if (err != SPL_Z_OK) return err;
            if (pos > end_bits) return SPL_Z_OVERRUN;
            wv::sync();
            if (lens[32u + 256u] == 0u) return SPL_Z_BAD_LENGTHS; // no end-of-block code
        }
        const uint8_t *const code_lens = type == 1u ? lens : lens + 32u;
        if (!build_lut<KIND_LIT>(code_lens, n_lit, ROOT_L, sh.lut_l, LUT_L, work)) return SPL_Z_BAD_LENGTHS;
        if (!build_lut<KIND_DIST>(code_lens + n_lit, n_dist, ROOT_D, sh.lut_d, LUT_D, work)) return SPL_Z_BAD_LENGTHS;
        // ---- the symbols, a tile at a time
        for (bool eob = false; !eob;) {
            if (pos >= end_bits) return SPL_Z_OVERRUN;
            const uint32_t base = pos & ~31u, byte0 = base >> 3;
            // the tile: 32 bytes per lane, and the words behind it (what lies beyond the block's data is never used: zeros will do,
            // and the image is readable for SPL_Z_IMAGE_PAD bytes past any block)
            {
                for (uint32_t r = 0; r < SUB_BITS / 256u; ++r) {
                    const uint32_t o = byte0 + 32u * (l + 64u * r);
                    uint64_t a = 0, b = 0, c = 0, d = 0;
                    if (o + 16u <= in_len + SPL_Z_IMAGE_PAD) { a = wv::ld64(in + o); b = wv::ld64(in + o + 8u); }
                    if (o + 32u <= in_len + SPL_Z_IMAGE_PAD) { c = wv::ld64(in + o + 16u); d = wv::ld64(in + o + 24u); }
                    uint32_t *t = sh.tile + 8u * (l + 64u * r);
                    t[0] = (uint32_t)a; t[1] = (uint32_t)(a >> 32); t[2] = (uint32_t)b; t[3] = (uint32_t)(b >> 32);
                    t[4] = (uint32_t)c; t[5] = (uint32_t)(c >> 32); t[6] = (uint32_t)d; t[7] = (uint32_t)(d >> 32);
                }
                if (l < TILE_PAD / 4u) {
                    const uint32_t o2 = byte0 + TILE_WORDS * 4u + 16u * l;
                    uint64_t e = 0, f = 0;
                    if (o2 + 16u <= in_len + SPL_Z_IMAGE_PAD) { e = wv::ld64(in + o2); f = wv::ld64(in + o2 + 8u); }
                    uint32_t *t2 = sh.tile + TILE_WORDS + 4u * l;
                    t2[0] = (uint32_t)e; t2[1] = (uint32_t)(e >> 32); t2[2] = (uint32_t)f; t2[3] = (uint32_t)(f >> 32);
                }
            }
            wv::sync();
            // ---- where every lane starts: guessed, then corrected from the lane before until nothing changes
            const uint32_t sub_begin = base + SUB_BITS * l, sub_end = sub_begin + SUB_BITS;
            uint32_t start = l ? sub_begin : pos;
            bool dead = l != 0u && sub_begin >= end_bits;
            Count c{start, 0, 0, FL_OK};
            if (!dead) c = count_from(sh, base, start, sub_end);
            // From the first correction on a lane writes its tokens down WHILE it counts (emit_from), into a place of its own in the
            // token room sized by what its first, guessed decode counted and a little more (a corrected start moves a lane's count
            // by a few bytes): when the starts have settled the tokens are there, and the writing pass below -- all the tile's
            // symbols decoded once more -- is not needed; a tile whose places do not fit the room, or a lane whose tokens do not
            // fit its place, takes that pass as before.
            uint8_t *const tok = (uint8_t *)sh.tok;
            const uint32_t place_len = dead ? 0u : c.n_tok + 12u;
            const uint32_t place_incl = wv::scan_add(place_len);
            const bool places = !(opts & OPT_WRITING_PASS) && wv::readlane(place_incl, 63u) <= CAP;
            uint8_t *const place = tok + (place_incl - place_len);
            bool emitted = false, fits = true;
            uint32_t need = 0;
            for (uint32_t pass = 0;; ++pass) {
                if (pass > 66u) return SPL_Z_OVERRUN; // (cannot happen: lane k is settled after pass k + 1)
                const uint32_t p_end = wv::shfl_up(c.end, 1u), p_flag = wv::shfl_up(c.flag, 1u), p_dead = wv::shfl_up(dead ? 1u : 0u, 1u);
                const bool want_dead = l != 0u && (p_dead != 0u || p_flag != FL_OK || p_end >= end_bits);
                const bool redo = l != 0u && (want_dead != dead || (!want_dead && p_end != start));
                if (!wv::any(redo)) break;
                // (the first round is nearly every lane's: whoever is right already -- lane 0, a lucky guess -- writes its tokens with it)
                if (redo || (places && pass == 0u && !dead)) {
                    if (redo) { dead = want_dead; start = p_end; }
                    c = Count{start, 0, 0, FL_OK};
                    emitted = false;
                    // (the first two rounds settle all but a tile's stragglers -- where the codes at hand are all about as long as each
                    //  other the true start travels one lane a round, a dozen rounds and more: those rounds only count, and the lanes
                    //  they corrected write their tokens in one round of their own behind the loop)
                    if (!dead && places && pass < EMIT_ROUNDS) { c = emit_from(sh, base, start, sub_end, place, place_len, need, fits); emitted = true; }
                    else if (!dead) c = count_from(sh, base, start, sub_end);
                }
            }
            if (places && wv::any(!dead && !emitted)) { // (no round at all, or a lane that came alive again: its tokens now)
                if (!dead && !emitted) { c = emit_from(sh, base, start, sub_end, place, place_len, need, fits); emitted = true; }
            }
            // (a lane whose tokens did not fit its place -- its guessed decode had counted something else altogether -- writes them once
            //  more behind the others, into room of exactly their size; should even that not fit, the writing pass below takes the tile)
            const bool misfit = places && !dead && !fits;
            const uint32_t again_len = misfit ? c.n_tok : 0u, again_incl = wv::scan_add(again_len);
            const bool in_place = places && wv::readlane(again_incl, 63u) <= CAP;
            // the lanes that count: all that are alive (a suffix of the lanes is dead), short of the one whose tokens overflow the tile's room
            const uint32_t cum_t = wv::scan_add(dead ? 0u : c.n_tok), cum_o = wv::scan_add(dead ? 0u : c.n_out);
            const uint64_t m_ok = wv::ballot(!dead && (in_place || cum_t <= CAP));
            const uint32_t n_valid = ~m_ok ? wv::ffs64(~m_ok) : 64u; // (the low run of ones)
            if (n_valid == 0u) return SPL_Z_OVERRUN;
            const bool valid = l < n_valid;
            if (wv::any(valid && c.flag == FL_ERR)) return SPL_Z_BAD_CODE;
            eob = wv::any(valid && c.flag == FL_EOB);
            const uint32_t total = wv::readlane(cum_o, n_valid - 1u), n_t = wv::readlane(cum_t, n_valid - 1u);
            if (at + total > out_len) return SPL_Z_OVERRUN;
            if (n_tok + n_t > tok_cap - 64u) return SPL_Z_TOKENS;
            if (in_place) {
                // the tokens are written: each lane's to its stretch of the block's stream, sixteen bytes at a time, the last ones exactly
                if (wv::any(valid && need > at + cum_o - c.n_out)) return SPL_Z_BAD_DISTANCE;
                uint8_t *const dst = stream + n_tok + (cum_t - c.n_tok);
                auto copy_out = [&](const uint8_t *from) {
                    for (uint32_t o = 0; o < c.n_tok; o += 16u) {
                        uint64_t lo, hi;
                        wv::lds_ld128(from + o, lo, hi);
                        store_n(dst + o, lo, hi, c.n_tok - o < 16u ? c.n_tok - o : 16u);
                    }
                };
                if (valid && !misfit) copy_out(place);
                if (wv::any(valid && misfit)) {
                    wv::sync(); // (the places are read: the room is free)
                    if (valid && misfit) {
                        uint8_t *const again = tok + (again_incl - again_len);
                        uint32_t need2;
                        bool fits2;
                        (void)emit_from(sh, base, start, sub_end, again, again_len, need2, fits2);
                        copy_out(again);
                    }
                }
                n_tok += n_t;
                wv::sync();
                at += total;
                pos = wv::readlane(c.end, n_valid - 1u);
                continue;
            }
            // ---- the writing pass: every lane's symbols as tokens, at the lane's place in the tile's stretch of the stream
            bool bad_dist = false;
#ifndef SPL_EXP_NO_WRITE
            if (valid) {
                uint32_t p = start, wr = at + cum_o - c.n_out, tp = cum_t - c.n_tok, run = 0, hdr = 0;
                while (p < sub_end) {
                    uint32_t len = 0, dist = 0, lit2;
                    const uint32_t s = decode(sh, base, p, len, dist, sub_end, lit2);
                    if (s < 256u) {
                        if (run == 0u || run == 128u) { if (run) tok[hdr] = 127; hdr = tp++; run = 0; }
                        tok[tp++] = (uint8_t)s;
                        ++run; ++wr;
                        if (lit2 != NO_LIT2) {
                            if (run == 128u) { tok[hdr] = 127; hdr = tp++; run = 0; }
                            tok[tp++] = (uint8_t)lit2;
                            ++run; ++wr;
                        }
                        continue;
                    }
                    if (run) { tok[hdr] = (uint8_t)(run - 1u); run = 0; }
                    if (s != SYM_MATCH) break; // (the end of the block; errors were seen by the counting pass)
                    if (dist > wr) { bad_dist = true; break; }
                    const uint32_t L = len - 3u, D = dist - 1u;
                    tok[tp] = (uint8_t)(0x80u | (L & 0x7fu)); tok[tp + 1u] = (uint8_t)(L >> 7 | (D & 0x7fu) << 1); tok[tp + 2u] = (uint8_t)(D >> 7);
                    tp += 3u;
                    wr += len;
                }
                if (run) tok[hdr] = (uint8_t)(run - 1u);
            }
#endif
            if (wv::any(bad_dist)) return SPL_Z_BAD_DISTANCE;
            wv::sync();
            // the tile's tokens to their place in the block's stream: sixteen bytes a lane and step (what is written beyond n_t is
            // overwritten by the next tile, or never read)
            for (uint32_t o = l * 16u; o < n_t; o += 1024u) {
                const uint64_t lo = (uint64_t)sh.tok[o / 4u] | (uint64_t)sh.tok[o / 4u + 1u] << 32, hi = (uint64_t)sh.tok[o / 4u + 2u] | (uint64_t)sh.tok[o / 4u + 3u] << 32;
                wv::st128(stream + n_tok + o, lo, hi);
            }
            n_tok += n_t;
            wv::sync();
            at += total;
            pos = wv::readlane(c.end, n_valid - 1u);
        }
    }
    if (at != out_len) return SPL_Z_SHORT;
    n_tok_out = n_tok;
    return SPL_Z_OK;
}

// ---- the copying kernel's body: one lane, one block ------------------------------------------------------------------------
#ifndef SPLZ_RING
#define SPLZ_RING 64
#endif
#ifndef SPLZ_FIFO
#define SPLZ_FIFO 128
#endif
constexpr uint32_t RING = SPLZ_RING;        // bytes of the output a lane keeps at hand in LDS: a match up to RING - 16 back is copied from there
constexpr uint32_t RING_BYTES = 16u + RING + 32u; // ... with room in front and behind, so that no piece of 16 bytes ever wraps (mirrored)
constexpr uint32_t FIFO = SPLZ_FIFO;        // bytes of token stream a lane holds in LDS
constexpr uint32_t FIFO_BYTES = FIFO + 16u;
constexpr uint32_t BEAT = 8;          // every BEAT turns: the 64 bytes asked for a beat ago go into the FIFO, the next 64 are asked for
constexpr uint32_t COPY_LANE_BYTES = RING_BYTES + FIFO_BYTES + 4u; // (an odd number of words: the lanes' windows start in different banks)

// ring: this lane's RING_BYTES, fifo: its FIFO_BYTES (shared memory).  Returns the number of bytes made (the caller compares).
WV_DEV uint32_t copy_block(uint8_t *out, uint32_t out_len, const uint8_t *stream, uint32_t n_tok, uint8_t *ring_mem, uint8_t *fifo)
{
    if (n_tok == 0u) return 0u;
    wv::settle(n_tok);
    wv::settle(out_len);
    uint8_t *const ring = ring_mem + 16; // (ring[-16 .. RING + 32))
    uint32_t sp = 0, f_wr = 0, f_req = 0;   // token bytes consumed / in the FIFO / asked for
    wv::q128 qa = {}, qb = {}, qc = {}, qd = {}; // the 64 bytes on their way
    bool pending = false;
    uint32_t op = 0, flushed = 0;           // bytes made / bytes of them in memory (a multiple of 16)
    uint32_t lit_left = 0, left = 0, dist = 1;
    uint64_t ahead_lo = 0, ahead_hi = 0;    // the next piece of a far match, asked for a turn ahead
    bool ahead = false;
    // ... and the FIRST piece of the next far match, asked for several turns ahead: a second reader walks the tokens in the FIFO
    // in front of the first (one token a turn), keeping count of where in the output each begins, and when it comes to a match
    // whose source lies further back than the window -- and in memory already -- it asks for those 16 bytes and stands still
    // until the first reader has got there and taken them.
    uint32_t la_sp = 0, la_op = 0;          // the second reader: token position, and the output position where that token begins
    uint64_t pre_lo = 0, pre_hi = 0;
    uint32_t pre_sp = 0;                    // the token the bytes asked for belong to
    bool pre = false, first = false;
    // a piece of 16 bytes (the first k meant) to the window at `op`, mirrored where the window wraps
    auto to_ring = [&](uint64_t lo, uint64_t hi) {
        const uint32_t o = op & (RING - 1u);
        wv::lds_st128(ring + o, lo, hi);
        if (o < 16u) wv::lds_st128(ring + RING + o, lo, hi);
        if (o > RING - 16u) wv::lds_st128(ring + o - RING, lo, hi);
    };
    for (uint32_t turn = 0;; ++turn) {
        if ((turn & (BEAT - 1u)) == 0u) { // the beat
            if (pending && f_wr + 64u - sp <= FIFO) {
                uint8_t *f = fifo + (f_wr & (FIFO - 1u));
                wv::mem_wait4(qa, qb, qc, qd);
                wv::lds_st128q(f, qa); wv::lds_st128q(f + 16, qb); wv::lds_st128q(f + 32, qc); wv::lds_st128q(f + 48, qd);
                if ((f_wr & (FIFO - 1u)) == 0u) wv::lds_st128q(fifo + FIFO, qa); // (the FIFO's first piece once more behind its end)
                f_wr += 64u;
                pending = false;
            }
            if (!pending && f_req < n_tok) {
                const uint8_t *s = stream + f_req;
                wv::mem_ld128_async(s, qa); wv::mem_ld128_async(s + 16, qb); wv::mem_ld128_async(s + 32, qc); wv::mem_ld128_async(s + 48, qd);
                f_req += 64u;
                pending = true;
            }
        }
        const uint32_t have = (f_wr < n_tok ? f_wr : n_tok) - sp; // token bytes at hand
        if (lit_left == 0u && left == 0u) { // the next token
            if (sp >= n_tok) break;
            if (have < 3u && sp + have < n_tok) continue; // (its bytes are still on their way)
            const uint8_t *t = fifo + (sp & (FIFO - 1u));
            const uint32_t c = wv::lds_ld8(t);
            if (c < 0x80u) { lit_left = c + 1u; sp += 1u; }
            else {
                const uint32_t b1 = wv::lds_ld8(t + 1), b2 = wv::lds_ld8(t + 2);
                left = ((c & 0x7fu) | (b1 & 1u) << 7) + 3u;
                dist = ((b1 >> 1) | b2 << 7) + 1u;
                first = pre && pre_sp == sp;       // (its first sixteen bytes are on their way, or here)
                if (first) pre = false;
                sp += 3u;
                if (dist > op) return 0xffffffffu; // (the decoding kernel checked: a damaged stream)
            }
        }
        uint32_t k = 0;
        uint64_t lo = 0, hi = 0;
        if (lit_left) {
            const uint32_t avail = (f_wr < n_tok ? f_wr : n_tok) - sp;
            k = lit_left < 16u ? lit_left : 16u;
            if (avail < k) { if (avail == 0u) continue; k = avail; }
            wv::lds_ld128(fifo + (sp & (FIFO - 1u)), lo, hi);
            sp += k;
            lit_left -= k;
        } else {
            k = left < 16u ? left : 16u;
            if (dist >= 8u && dist < k) k = dist;    // (only what is there already; below 8 the piece is made from the period)
            if (dist <= RING - 16u) wv::lds_ld128(ring + ((op - dist) & (RING - 1u)), lo, hi);
#ifdef SPLZ_X_NOFAR
            else wv::lds_ld128(ring + ((op - dist) & (RING - 1u)), lo, hi);
#else
            else if (ahead) { lo = ahead_lo; hi = ahead_hi; wv::settle64(lo); wv::settle64(hi); } // (asked for by the turn before)
            else if (first) { lo = pre_lo; hi = pre_hi; wv::settle64(lo); wv::settle64(hi); }     // (asked for by the second reader)
            else { wv::mem_ld128(out + op - dist, lo, hi); wv::settle64(lo); wv::settle64(hi); }
#endif
            first = false; // (further back than the window: in memory, whole pieces behind `flushed`; waited for here, not where the two ways meet)
            if (dist < 8u) { // the bytes repeat with a period shorter than the piece: sixteen bytes of the period
                lo &= (1ull << (8u * dist)) - 1ull;
                lo |= lo << (8u * dist);
                if (dist < 4u) lo |= lo << (16u * dist);
                if (dist < 2u) lo |= lo << 32;
                const uint32_t r = (0x1230200u >> (4u * (dist - 1u))) & 7u; // 8 mod dist: where in the period the second half begins
                hi = r ? lo >> (8u * r) | lo << (8u * (dist - r)) : lo;
            }
            left -= k;
        }
        if (op + k > out_len) return 0xffffffffu;
        to_ring(lo, hi);
        op += k;
        if (op - flushed >= 16u) { // a whole piece of the output: to memory, once
            uint64_t a, b;
            wv::lds_ld128(ring + (flushed & (RING - 1u)), a, b);
#ifndef SPLZ_X_NOSTORE
            wv::mem_st128(out + flushed, a, b);
#endif
            flushed += 16u;
        }
        // a far match that goes on: its next piece is asked for now (behind this turn's store: what it reads is in memory, the
        // source ends at op - dist + 16 <= flushed as dist > RING - 16 >= 31) and used by the next turn, which then has not
        // waited for memory at all -- a long match was a chain of round trips, one per 16 bytes
        ahead = left != 0u && lit_left == 0u && dist > RING - 16u;
#ifdef SPLZ_X_NOFAR
        ahead = false;
#endif
        if (ahead) wv::mem_ld128(out + op - dist, ahead_lo, ahead_hi);
#ifndef SPLZ_X_NOFAR
        if (!pre) { // the second reader's turn: one token
            const uint32_t next_sp = sp + lit_left, next_op = op + lit_left + left; // where the first reader's next token begins
            if (la_sp < next_sp) { la_sp = next_sp; la_op = next_op; }
            const uint32_t there = f_wr < n_tok ? f_wr : n_tok;
            if (la_sp + 3u <= there) {
                const uint8_t *t = fifo + (la_sp & (FIFO - 1u));
                const uint32_t c = wv::lds_ld8(t);
                if (c < 0x80u) { la_sp += c + 2u; la_op += c + 1u; }
                else {
                    const uint32_t b1 = wv::lds_ld8(t + 1), b2 = wv::lds_ld8(t + 2);
                    const uint32_t len = ((c & 0x7fu) | (b1 & 1u) << 7) + 3u, d = ((b1 >> 1) | b2 << 7) + 1u;
                    if (d > RING - 16u && d <= la_op) {
                        if (la_op - d + 16u <= flushed) { // its source is in memory (else: next turn again, more will be)
                            wv::mem_ld128(out + la_op - d, pre_lo, pre_hi);
                            pre = true;
                            pre_sp = la_sp;
                            la_sp += 3u;
                            la_op += len;
                        }
                    } else { la_sp += 3u; la_op += len; }
                }
            }
        }
#endif
    }
    if (op > flushed) {
        uint64_t a, b;
        wv::lds_ld128(ring + (flushed & (RING - 1u)), a, b);
        store_n(out + flushed, a, b, op - flushed);
    }
    return op;
}

} // namespace splz

#endif
