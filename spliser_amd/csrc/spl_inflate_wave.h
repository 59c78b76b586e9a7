// spl_inflate_wave.h -- DEFLATE (RFC 1951) on the device in two kernels (spl_inflate.hip launches them; the bodies are here,
// written against the primitives of spl_wave.h so that the same source runs on the host under tests/hostsim/wave_emul.h):
//
//   decode_block   one BGZF block per WAVE: the Huffman decoding.  Literals go to their places in the inflated stream; a match
//                  is appended to the block's list of matches: where it goes, distance, length, eight bytes.
//   copy_block     one BGZF block per LANE: the list's matches made one after the other, each a copy of bytes that are there.
//
// Why the split.  Round 2 gave every lane a block of its own for everything: a lane's Huffman tables were 356 bytes of LDS and
// thirty-odd registers (1.75 waves per SIMD), every turn of the 64 lanes ran through all the decoder's branches (330 vector
// instructions a turn).  Decoding is what a wave can share: here 64 lanes use ONE pair of tables (two-level look-up tables in
// LDS, 2.9 KB) and read the compressed bytes from a tile of them staged in LDS by coalesced loads.  Copying is what it cannot: a
// BAM record's matches copy from the record before, whose matches copy from the one before that -- measured with
// tools/inflate_sim.cpp: about 2 000 dependent steps per block, a dozen copies ready at any time -- so 64 lanes on one block's
// copies idle (built and measured: 33 of the kernel's 58 ms, profiles/r03_inflate_wave_account.md), while 64 lanes on 64 blocks'
// copies are all busy and need no tables.
//
// Huffman codes have no markers, so where a lane should start decoding is not known: it is FOUND.  The data of a DEFLATE block
// is worked off in tiles of 64 subsequences of SUB_BITS bits, one per lane.  Every lane decodes from a guessed start (the beginning
// of its subsequence; lane 0 from the true position) to the first symbol boundary at or past its subsequence's end, counting what
// it would produce; then every lane takes its predecessor's end as its start and decodes again if that differs, until nothing
// changes.  Wrongly started decoders fall into step with the true sequence of symbols sooner or later, so a few passes do; lane k
// is right after pass k + 1 whatever the data.  A prefix sum over the lanes' byte counts places every lane in the output, and a last
// pass decodes once more for good.
//
// Replaces what SpliSER_v0_1_8.py:422 (samtools view) does to every BGZF block it touches.
#ifndef SPL_INFLATE_WAVE_H
#define SPL_INFLATE_WAVE_H

#include "spl_inflate.h"

namespace splz {

constexpr uint32_t ROOT_L = 9, ROOT_D = 6, ROOT_C = 7;
constexpr uint32_t LUT_L = 852, LUT_D = 592; // entries: root table + the most sub-tables a valid code can need (zlib's ENOUGH_LENS / ENOUGH_DISTS for these roots)
constexpr uint32_t SUB_BITS = 256;           // bits of DEFLATE data per lane and tile
constexpr uint32_t TILE_WORDS = 64u * SUB_BITS / 32u;
constexpr uint32_t TILE_PAD = 16;            // words behind the tile: a symbol that begins in the last subsequence ends there
constexpr uint32_t QCAP = 640;               // matches per tile (a tile with more is cut short)
constexpr uint32_t FL_OK = 0, FL_EOB = 1, FL_ERR = 2;
constexpr uint32_t SYM_EOB = 256, SYM_MATCH = 257, SYM_BAD = 0xffffffffu;

// One wave's shared memory: 10120 bytes (16 waves on a CU's 160 KB).
struct Shared {
    uint16_t lut_l[LUT_L];                // literal/length code.  Entry: symbol << 4 | bits; 0x8000 | offset << 4 | sub-table bits; 0 = no such code
    uint16_t lut_d[LUT_D];                // distance code
    uint32_t tile[TILE_WORDS + TILE_PAD]; // the compressed bytes being worked on (while tables are built: work space)
    uint64_t q[QCAP];                     // the tile's matches in output order (while a header is read: code lengths, the code-length code's table)
};
static_assert(SUB_BITS == 256u, "the tile is staged thirty-two bytes a lane");
static_assert(QCAP * 8u >= 352u + 256u, "code lengths and the code-length code's table lie in q while a header is read");

// the order in which a dynamic header lists the lengths of the code-length code (RFC 1951, 3.2.7), five bits a place
constexpr uint64_t pack5(const int *v, int n) { uint64_t r = 0; for (int i = 0; i < n; ++i) r |= (uint64_t)v[i] << (5 * i); return r; }
constexpr int k_clen_a[12] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4}, k_clen_b[7] = {12, 3, 13, 2, 14, 1, 15};
constexpr uint64_t k_clen_lo = pack5(k_clen_a, 12), k_clen_hi = pack5(k_clen_b, 7);
WV_DEV uint32_t clen_order(uint32_t i) { return (uint32_t)((i < 12u ? k_clen_lo >> (5u * i) : k_clen_hi >> (5u * (i - 12u))) & 31u); }

WV_DEV uint32_t bitrev(uint32_t v, uint32_t n) { return wv::brev32(v) >> (32u - n); }

// 32 bits of the block's data from bit `pos`, straight from memory (headers: every lane asks for the same bytes)
WV_DEV uint32_t gbits(const uint8_t *in, uint32_t pos) { return (uint32_t)(wv::ld64(in + (pos >> 3)) >> (pos & 7u)); }

// Canonical code -> look-up table, by the whole wave.  lens[0..n): code lengths (bytes in shared memory); work: 1 << root words.
// false: the lengths over-subscribe the code space, or need more sub-tables than any valid code does.
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
            lut[bitrev(P, root)] = (uint16_t)(0x8000u | off << 4 | b);
            for (uint32_t k = 0; k < (1u << b); ++k) lut[off + k] = 0;
            off += 1u << b;
        }
    }
    wv::sync();
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const uint32_t len = L[c], s = (uint32_t)c * 64u + l;
        if (!len) continue;
        if (len <= root) {
            for (uint32_t k = bitrev(cd[c], len); k < n_root; k += 1u << len) lut[k] = (uint16_t)(s << 4 | len);
        } else {
            const uint32_t e = work[cd[c] >> (len - root)], sub = e >> 4, b = e & 15u, rest = len - root;
            for (uint32_t k = bitrev(cd[c] & ((1u << rest) - 1u), rest); k < (1u << b); k += 1u << rest) lut[sub + k] = (uint16_t)(s << 4 | rest);
        }
    }
    wv::sync();
    return true;
}

// 32 bits of the tile from bit `pos` of the block's data (the tile begins at bit `base`)
WV_DEV uint32_t tbits(const uint32_t *tile, uint32_t base, uint32_t pos)
{
    const uint32_t rel = pos - base, w = rel >> 5;
    const uint64_t two = (uint64_t)tile[w] | (uint64_t)tile[w + 1u] << 32;
    return (uint32_t)(two >> (rel & 31u));
}

// the code at the head of `w`: symbol, and the bits it takes in `used`; SYM_BAD when there is no such code
WV_DEV uint32_t lut_symbol(const uint16_t *lut, uint32_t root, uint32_t w, uint32_t &used)
{
    uint32_t e = lut[w & ((1u << root) - 1u)];
    used = 0;
    if (e & 0x8000u) {
        used = root;
        e = lut[((e >> 4) & 0x7ffu) + ((w >> root) & ((1u << (e & 15u)) - 1u))];
    }
    if ((e & 15u) == 0u) return SYM_BAD;
    used += e & 15u;
    return e >> 4;
}

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

// One symbol at bit `pos` of the tile: a literal (its value), SYM_EOB, SYM_MATCH (len, dist set) or SYM_BAD.  `pos` moves past it.
WV_DEV uint32_t decode(const Shared &sh, uint32_t base, uint32_t &pos, uint32_t &len, uint32_t &dist)
{
    uint32_t w = tbits(sh.tile, base, pos), used;
    const uint32_t sym = lut_symbol(sh.lut_l, ROOT_L, w, used);
    if (sym == SYM_BAD) return SYM_BAD;
    if (sym <= 256u) { pos += used; return sym; }
    if (sym > 285u) return SYM_BAD;
    uint32_t b, x;
    length_code(sym - 257u, b, x);
    len = b + ((w >> used) & ((1u << x) - 1u)); // (a code and its extra bits: 20 at most)
    pos += used + x;
    w = tbits(sh.tile, base, pos);
    const uint32_t ds = lut_symbol(sh.lut_d, ROOT_D, w, used);
    if (ds == SYM_BAD || ds >= 30u) return SYM_BAD;
    distance_code(ds, b, x);
    dist = b + ((w >> used) & ((1u << x) - 1u)); // (28 at most)
    pos += used + x;
    return SYM_MATCH;
}

// What a lane would produce from `start` to the first symbol boundary at or past `sub_end`.
struct Count { uint32_t end, n_out, n_match, flag; };
WV_DEV Count count_from(const Shared &sh, uint32_t base, uint32_t start, uint32_t sub_end)
{
    Count c{start, 0, 0, FL_OK};
    while (c.end < sub_end) {
        uint32_t len = 0, dist = 0;
        const uint32_t s = decode(sh, base, c.end, len, dist);
        if (s < 256u) { c.n_out++; continue; }
        if (s == SYM_MATCH) { c.n_out += len; c.n_match++; continue; }
        c.flag = s == SYM_EOB ? FL_EOB : FL_ERR;
        break;
    }
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

// The block `zb` of the file image, by one wave.  midx: room for `stride` places of matches; n_match_out: how many were written.  Returns the block's status (every lane the same).
WV_DEV uint32_t decode_block(Shared &sh, const uint8_t *image, const spl_zblock &zb, uint8_t *out_all, uint64_t *midx, uint32_t stride, uint32_t &n_match_out)
{
    n_match_out = 0;
    uint32_t n_match = 0;
    const uint32_t l = wv::lane();
    const uint8_t *const in = image + zb.in;
    uint8_t *const out = out_all + zb.out;
    const uint32_t in_len = zb.in_len, out_len = zb.out_len, end_bits = in_len * 8u;
    if (out_len == 0u) return SPL_Z_OK; // (the EOF marker and other empty blocks: nothing to decode into)
    if (out_len > 65536u || in_len > 65536u) return SPL_Z_OVERRUN; // (not a BGZF block)
    uint8_t *const lens = (uint8_t *)sh.q; // 352 code lengths while a header is read
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
            for (uint32_t i = l * 4u; i + 4u <= len; i += 256u) wv::st32(out + at + i, wv::ld32(src + i));
            if ((len & ~3u) + l < len) out[at + (len & ~3u) + l] = src[(len & ~3u) + l];
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
            uint16_t *const lut_c = (uint16_t *)sh.q + 176;
            if (!build_lut(lens, 19u, ROOT_C, lut_c, 1u << ROOT_C, work)) return SPL_Z_BAD_LENGTHS;
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
        if (!build_lut(code_lens, n_lit, ROOT_L, sh.lut_l, LUT_L, work)) return SPL_Z_BAD_LENGTHS;
        if (!build_lut(code_lens + n_lit, n_dist, ROOT_D, sh.lut_d, LUT_D, work)) return SPL_Z_BAD_LENGTHS;
        // ---- the symbols, a tile at a time
        for (bool eob = false; !eob;) {
            if (pos >= end_bits) return SPL_Z_OVERRUN;
            const uint32_t base = pos & ~31u, byte0 = base >> 3;
            // the tile: 32 bytes per lane, and the words behind it (what lies beyond the block's data is never used: zeros will do,
            // and the image is readable for SPL_Z_IMAGE_PAD bytes past any block)
            {
                const uint32_t o = byte0 + 32u * l;
                uint64_t a = 0, b = 0, c = 0, d = 0;
                if (o + 16u <= in_len + SPL_Z_IMAGE_PAD) { a = wv::ld64(in + o); b = wv::ld64(in + o + 8u); }
                if (o + 32u <= in_len + SPL_Z_IMAGE_PAD) { c = wv::ld64(in + o + 16u); d = wv::ld64(in + o + 24u); }
                uint32_t *t = sh.tile + 8u * l;
                t[0] = (uint32_t)a; t[1] = (uint32_t)(a >> 32); t[2] = (uint32_t)b; t[3] = (uint32_t)(b >> 32);
                t[4] = (uint32_t)c; t[5] = (uint32_t)(c >> 32); t[6] = (uint32_t)d; t[7] = (uint32_t)(d >> 32);
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
            for (uint32_t pass = 0;; ++pass) {
                if (pass > 66u) return SPL_Z_OVERRUN; // (cannot happen: lane k is settled after pass k + 1)
                const uint32_t p_end = wv::shfl_up(c.end, 1u), p_flag = wv::shfl_up(c.flag, 1u), p_dead = wv::shfl_up(dead ? 1u : 0u, 1u);
                const bool want_dead = l != 0u && (p_dead != 0u || p_flag != FL_OK || p_end >= end_bits);
                const bool redo = l != 0u && (want_dead != dead || (!want_dead && p_end != start));
                if (!wv::any(redo)) break;
                if (redo) {
                    dead = want_dead;
                    start = p_end;
                    c = Count{start, 0, 0, FL_OK};
                    if (!dead) c = count_from(sh, base, start, sub_end);
                }
            }
            // the lanes that count: all that are alive (a suffix of the lanes is dead), short of the one whose matches overflow the queue
            const uint32_t cum_m = wv::scan_add(dead ? 0u : c.n_match), cum_o = wv::scan_add(dead ? 0u : c.n_out);
            const uint64_t m_ok = wv::ballot(!dead && cum_m <= QCAP);
            const uint32_t n_valid = ~m_ok ? wv::ffs64(~m_ok) : 64u; // (the low run of ones)
            if (n_valid == 0u) return SPL_Z_OVERRUN;
            const bool valid = l < n_valid;
            if (wv::any(valid && c.flag == FL_ERR)) return SPL_Z_BAD_CODE;
            eob = wv::any(valid && c.flag == FL_EOB);
            const uint32_t total = wv::readlane(cum_o, n_valid - 1u), n_q = wv::readlane(cum_m, n_valid - 1u);
            if (at + total > out_len) return SPL_Z_OVERRUN;
            if (n_match + n_q > stride) return SPL_Z_TOO_MANY;
            // ---- the writing pass: literals to their places, matches to the list
            bool bad_dist = false;
#ifndef SPL_EXP_NO_WRITE
            if (valid) {
                uint32_t p = start, wr = at + cum_o - c.n_out, qi = cum_m - c.n_match;
                while (p < sub_end) {
                    uint32_t len = 0, dist = 0;
                    const uint32_t s = decode(sh, base, p, len, dist);
                    if (s < 256u) { out[wr++] = (uint8_t)s; continue; }
                    if (s != SYM_MATCH) break; // (the end of the block; errors were seen by the counting pass)
                    if (dist > wr) { bad_dist = true; break; }
                    sh.q[qi++] = (uint64_t)(wr | (dist - 1u) << 16) | (uint64_t)(len - 3u) << 32;
                    wr += len;
                }
            }
#endif
            if (wv::any(bad_dist)) return SPL_Z_BAD_DISTANCE;
            wv::sync();
            for (uint32_t t = l; t < n_q; t += 64u) midx[n_match + t] = sh.q[t];
            n_match += n_q;
            wv::sync();
            at += total;
            pos = wv::readlane(c.end, n_valid - 1u);
        }
    }
    if (at != out_len) return SPL_Z_SHORT;
    n_match_out = n_match;
    return SPL_Z_OK;
}

// The matches of a block made in the order of its list: every one a copy of bytes that are there by then (literals, and the
// matches before it).  One lane's work; 64 blocks to a wave, 64 different cache lines to every memory instruction: two matches'
// entries in one 16-byte load, a piece of up to 16 bytes in one load and at most two stores (the second overlaps the first: a
// piece of 11 bytes is bytes 0..7 and bytes 3..10).  (Asking for the next piece's bytes before this piece is stored -- two loads
// on their way -- was built and is slower, 22.5 ms against 19.6 per window of 49 152 blocks.)
WV_DEV void copy_block(uint8_t *out, const uint64_t *midx, uint32_t n)
{
    if (n == 0u) return;
    uint64_t e0, e1; // entries i (and i + 1 when i is even): two to a load; midx + 2k is 16-byte aligned
    wv::ld128((const uint8_t *)midx, e0, e1);
    uint32_t i = 0, d = 0, left = 0, dist = 1;
    for (;;) {
        if (left == 0u) {
            if (i >= n) break;
            d = (uint32_t)e0 & 0xffffu;
            dist = ((uint32_t)e0 >> 16) + 1u;
            left = (uint32_t)(e0 >> 32) + 3u;
            ++i;
            if (i & 1u) e0 = e1;
            else if (i < n) wv::ld128((const uint8_t *)(midx + i), e0, e1);
        }
        uint32_t k = left < 16u ? left : 16u;
        if (dist < 8u) k = k < 8u ? k : 8u;      // (made from the period, below)
        else if (dist < k) k = dist;             // (only what is there already)
        uint64_t lo, hi;
        wv::ld128(out + d - dist, lo, hi);
        if (dist < 8u) { // the bytes repeat with a period shorter than the piece: the period, as often as it fits
            lo &= (1ull << (8u * dist)) - 1ull;
            lo |= lo << (8u * dist);
            if (dist < 4u) lo |= lo << (16u * dist);
            if (dist < 2u) lo |= lo << 32;
        }
        uint8_t *const p = out + d;
        if (k == 16u) wv::st128(p, lo, hi);
        else if (k >= 8u) { // bytes 0..7, then the last eight (they overlap when k < 16)
            wv::st64(p, lo);
            const uint32_t sh8 = 8u * (k - 8u);
            if (k > 8u) wv::st64(p + k - 8u, sh8 ? lo >> sh8 | hi << (64u - sh8) : lo);
        } else if (k >= 4u) {
            wv::st32(p, (uint32_t)lo);
            if (k > 4u) wv::st32(p + k - 4u, (uint32_t)(lo >> (8u * (k - 4u))));
        } else {
            if (k & 2u) wv::st16(p, (uint32_t)lo);
            if (k & 1u) p[k - 1u] = (uint8_t)(lo >> (8u * (k - 1u)));
        }
        d += k;
        left -= k;
    }
}

} // namespace splz

#endif
