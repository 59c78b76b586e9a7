// inflate_sim.cpp -- a CPU model of the wave-cooperative DEFLATE decoder of spl_inflate.hip (one BGZF block per wave), lane by
// lane and turn by turn: what the design costs on a given file (passes of the boundary search per tile, turns of the writing
// pass, stalls on bytes another lane has not written yet) before any of it runs on a GPU.  Every block is checked against zlib.
//
//   g++ -O2 -o /tmp/inflate_sim tools/inflate_sim.cpp -lz && /tmp/inflate_sim file.bam [sub_bits=256] [max_blocks]
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static const int ROOT_L = 9, ROOT_D = 6;
static int SUB = 256; // bits of compressed data per lane and tile

struct Lut {
    std::vector<uint16_t> e; // direct: sym << 4 | len; pointer: 0x8000 | off << 4 | sub_bits; 0 = no code
};

static uint32_t bitrev(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// canonical code -> two-level table (root bits, then one sub-table per root prefix that has longer codes)
static bool build_lut(const uint8_t *len, int n, int root, Lut &t)
{
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[len[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
    int first[16], code = 0;
    for (int l = 1; l < 16; ++l) { first[l] = code; code = (code + count[l]) << 1; }
    t.e.assign((size_t)1 << root, 0);
    std::vector<int> next(first, first + 16);
    std::vector<uint32_t> codes(n);
    for (int s = 0; s < n; ++s) if (len[s]) codes[s] = (uint32_t)next[len[s]]++;
    // sub-table sizes per root prefix (MSB-first prefix P)
    std::vector<int> sub_bits((size_t)1 << root, 0);
    for (int s = 0; s < n; ++s) if (len[s] > root) { const uint32_t P = codes[s] >> (len[s] - root); sub_bits[P] = std::max(sub_bits[P], len[s] - root); }
    std::vector<int> sub_off((size_t)1 << root, 0);
    int off = 1 << root;
    for (uint32_t P = 0; P < (1u << root); ++P) if (sub_bits[P]) { sub_off[P] = off; off += 1 << sub_bits[P]; t.e[bitrev(P, root)] = (uint16_t)(0x8000u | (uint32_t)sub_off[P] << 4 | (uint32_t)sub_bits[P]); }
    t.e.resize((size_t)off, 0);
    for (int s = 0; s < n; ++s) {
        const int L = len[s];
        if (!L) continue;
        if (L <= root) {
            const uint32_t r = bitrev(codes[s], L);
            for (uint32_t k = r; k < (1u << root); k += 1u << L) t.e[k] = (uint16_t)((uint32_t)s << 4 | (uint32_t)L);
        } else {
            const uint32_t P = codes[s] >> (L - root), rest = codes[s] & ((1u << (L - root)) - 1u);
            const uint32_t r = bitrev(rest, L - root);
            for (uint32_t k = r; k < (1u << sub_bits[P]); k += 1u << (L - root)) t.e[(size_t)sub_off[P] + k] = (uint16_t)((uint32_t)s << 4 | (uint32_t)(L - root));
        }
    }
    return true;
}

struct Bits {
    const uint8_t *d;
    size_t n; // bytes
    uint64_t peek(uint64_t pos) const // 57 bits at least, from bit `pos`
    {
        uint64_t v = 0;
        const size_t b = pos >> 3;
        for (int i = 0; i < 8; ++i) v |= (uint64_t)(b + i < n ? d[b + i] : 0) << (8 * i);
        return v >> (pos & 7);
    }
    uint32_t get(uint64_t &pos, int k) { const uint32_t v = (uint32_t)(peek(pos) & ((1ull << k) - 1)); pos += k; return v; }
};

static inline int lut_decode(const Lut &t, int root, uint64_t w, int &used)
{
    uint16_t e = t.e[w & ((1u << root) - 1)];
    used = 0;
    if (e & 0x8000u) {
        const int sb = e & 15, off = (e >> 4) & 0x7ff;
        used = root;
        e = t.e[(size_t)off + ((w >> root) & ((1u << sb) - 1))];
    }
    if ((e & 15) == 0) return -1;
    used += e & 15;
    return e >> 4;
}

static void length_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 8u || i == 28u ? 0u : (i >> 2) - 1u;
    base = i < 4u ? 3u + i : (i == 28u ? 258u : 3u + ((4u + (i & 3u)) << extra));
}
static void distance_code(uint32_t i, uint32_t &base, uint32_t &extra)
{
    extra = i < 4u ? 0u : (i >> 1) - 1u;
    base = i < 2u ? 1u + i : 1u + ((2u + (i & 1u)) << extra);
}

enum { FL_OK = 0, FL_EOB = 1, FL_ERR = 2 };
struct Sym { int kind; uint32_t lit, len, dist; }; // kind 0 literal, 1 match, 2 eob, 3 error

static Sym decode_one(const Bits &b, const Lut &L, const Lut &D, uint64_t &pos, uint64_t limit)
{
    Sym s{3, 0, 0, 0};
    if (pos >= limit) return s;
    uint64_t w = b.peek(pos);
    int used;
    const int sym = lut_decode(L, ROOT_L, w, used);
    if (sym < 0) return s;
    pos += used;
    if (sym < 256) { s.kind = 0; s.lit = (uint32_t)sym; return s; }
    if (sym == 256) { s.kind = 2; return s; }
    if (sym > 285) return s;
    uint32_t base, extra;
    length_code((uint32_t)sym - 257u, base, extra);
    w = b.peek(pos);
    s.len = base + (uint32_t)(w & ((1u << extra) - 1));
    pos += extra;
    w = b.peek(pos);
    const int ds = lut_decode(D, ROOT_D, w, used);
    if (ds < 0 || ds >= 30) return s;
    pos += used;
    distance_code((uint32_t)ds, base, extra);
    w = b.peek(pos);
    s.dist = base + (uint32_t)(w & ((1u << extra) - 1));
    pos += extra;
    s.kind = 1;
    return s;
}

struct Stats {
    uint64_t blocks = 0, dblocks = 0, tiles = 0, passes = 0, count_steps = 0, turns = 0, ideal_turns = 0, stalls = 0, lane_turns = 0, symbols = 0, literals = 0,
             matches = 0, match_bytes = 0, out_bytes = 0, in_bytes = 0, header_syms = 0, lanes_used = 0, searches = 0, max_passes = 0, short_dist = 0, stored = 0, rounds = 0, copy_rounds = 0, coop = 0, coop_steps = 0, active = 0, max_queue = 0, self_resolved = 0, self_pieces = 0, merged = 0, unmerged = 0, fwd_iters = 0;
    uint64_t pass_hist[66] = {0};
};

// One BGZF block the way a wave would do it.  Returns false on a mismatch with `want`.
static int coop_min = 16;
static int exact_dep = 0;
static int WIN = 64;
static int self_resolve = 0;
static int FWD = 0; // forward a match's source through the matches it copies from, before the copies are made
static int MERGE_K = 0; // 0 = plain re-decode; else two-pointer merge with the lane's previous chain for up to K steps
static int MERGE_FROM = 2; // first pass (1-based) that uses it
static bool wave_inflate(const uint8_t *data, size_t n, const std::vector<uint8_t> &want, Stats &st, int piece_max)
{
    Bits b{data, n};
    const uint64_t end_bits = (uint64_t)n * 8;
    std::vector<uint8_t> out(want.size() + 64, 0xA5);
    std::vector<uint8_t> written(want.size() + 64, 0);
    uint64_t pos = 0;
    size_t at = 0;
    st.blocks++;
    st.in_bytes += n;
    for (int last = 0; !last;) {
        last = (int)b.get(pos, 1);
        const uint32_t type = b.get(pos, 2);
        st.dblocks++;
        if (type == 0) {
            pos = (pos + 7) & ~7ull;
            const uint32_t len = b.get(pos, 16), nlen = b.get(pos, 16);
            if ((len ^ 0xffff) != nlen) return false;
            if (at + len > want.size()) return false;
            memcpy(out.data() + at, data + (pos >> 3), len);
            for (uint32_t i = 0; i < len; ++i) written[at + i] = 1;
            at += len;
            pos += 8ull * len;
            st.stored++;
            continue;
        }
        if (type == 3) return false;
        uint8_t lengths[320];
        int nlen, ndist;
        if (type == 1) {
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            nlen = 288; ndist = 30;
            for (s = 0; s < 30; ++s) lengths[288 + s] = 5;
        } else {
            nlen = (int)b.get(pos, 5) + 257; ndist = (int)b.get(pos, 5) + 1;
            const int ncode = (int)b.get(pos, 4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.get(pos, 3);
            Lut C;
            if (!build_lut(cl, 19, 7, C)) return false;
            int idx = 0;
            while (idx < nlen + ndist) {
                int used;
                const int sym = lut_decode(C, 7, b.peek(pos), used);
                if (sym < 0) return false;
                pos += used;
                st.header_syms++;
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                int prev = 0, rep;
                if (sym == 16) { if (!idx) return false; prev = lengths[idx - 1]; rep = 3 + (int)b.get(pos, 2); }
                else if (sym == 17) rep = 3 + (int)b.get(pos, 3);
                else rep = 11 + (int)b.get(pos, 7);
                if (idx + rep > nlen + ndist) return false;
                while (rep--) lengths[idx++] = (uint8_t)prev;
            }
        }
        Lut L, D;
        if (!build_lut(lengths, nlen, ROOT_L, L)) return false;
        if (!build_lut(lengths + nlen, ndist, ROOT_D, D)) return false;
        // ---- tiles
        bool eob = false;
        while (!eob) {
            st.tiles++;
            const uint64_t B = pos & ~31ull;
            uint64_t start[64], endp[64];
            uint32_t nout[64], nsym[64];
            int flag[64];
            bool dead[64];
            int n_lanes = 0;
            for (int l = 0; l < 64; ++l) {
                start[l] = l ? B + (uint64_t)SUB * l : pos;
                dead[l] = start[l] >= end_bits || (l && B + (uint64_t)SUB * l <= pos); // (lane 0's start may lie beyond the first subsequence's... it cannot: pos - B < 32)
                if (!dead[l]) n_lanes = l + 1;
            }
            auto count_lane = [&](int l) {
                uint64_t p = start[l];
                const uint64_t sub_end = B + (uint64_t)SUB * (l + 1);
                uint32_t no = 0, ns = 0;
                int f = FL_OK;
                while (p < sub_end) {
                    const Sym s = decode_one(b, L, D, p, end_bits);
                    ns++;
                    if (s.kind == 3) { f = FL_ERR; break; }
                    if (s.kind == 2) { f = FL_EOB; break; }
                    no += s.kind == 0 ? 1u : s.len;
                }
                endp[l] = p; nout[l] = no; nsym[l] = ns; flag[l] = f;
                return ns;
            };
            uint32_t step_max = 0;
            for (int l = 0; l < n_lanes; ++l) if (!dead[l]) step_max = std::max(step_max, (uint32_t)count_lane(l));
            st.count_steps += step_max;
            int passes = 1;
            for (;;) {
                // Jacobi: what every lane wants as its start, from its predecessor's last result
                uint64_t want_start[64];
                bool want_dead[64], redo[64];
                bool any = false;
                for (int l = 1; l < n_lanes; ++l) {
                    want_dead[l] = dead[l - 1] || flag[l - 1] != FL_OK || endp[l - 1] >= end_bits;
                    want_start[l] = endp[l - 1];
                    redo[l] = want_dead[l] != dead[l] || (!want_dead[l] && want_start[l] != start[l]);
                    any = any || redo[l];
                }
                if (!any) break;
                step_max = 0;
                for (int l = 1; l < n_lanes; ++l) {
                    if (!redo[l]) continue;
                    const bool was_dead = dead[l];
                    dead[l] = want_dead[l];
                    if (dead[l]) continue;
                    if (MERGE_K && passes + 1 >= MERGE_FROM && !was_dead) {
                        // the chain from the new start (a) and the lane's chain so far (b), the one behind moves
                        const uint64_t sub_end = B + (uint64_t)SUB * (l + 1);
                        uint64_t pa = want_start[l], pb = start[l];
                        uint32_t na = 0, nb = 0, sa = 0;
                        bool merged = false, a_done = false;
                        int fa = FL_OK;
                        uint32_t steps = 0;
                        while ((int)steps < MERGE_K) {
                            if (pa == pb) { merged = pa < sub_end; break; }
                            if (pa < pb) {
                                if (pa >= sub_end) break;
                                const Sym y = decode_one(b, L, D, pa, end_bits); ++steps; ++sa;
                                if (y.kind == 3) { fa = FL_ERR; a_done = true; break; }
                                if (y.kind == 2) { fa = FL_EOB; a_done = true; break; }
                                na += y.kind == 0 ? 1u : y.len;
                            } else {
                                if (pb >= sub_end) break;
                                const Sym y = decode_one(b, L, D, pb, end_bits); ++steps;
                                if (y.kind >= 2) break;
                                nb += y.kind == 0 ? 1u : y.len;
                            }
                        }
                        start[l] = want_start[l];
                        if (merged) { nout[l] = nout[l] - nb + na; st.merged++; }
                        else {
                            while (!a_done && pa < sub_end) {
                                const Sym y = decode_one(b, L, D, pa, end_bits); ++steps; ++sa;
                                if (y.kind == 3) { fa = FL_ERR; break; }
                                if (y.kind == 2) { fa = FL_EOB; break; }
                                na += y.kind == 0 ? 1u : y.len;
                            }
                            endp[l] = pa; nout[l] = na; flag[l] = fa; nsym[l] = sa; st.unmerged++;
                        }
                        step_max = std::max(step_max, steps);
                        continue;
                    }
                    start[l] = want_start[l];
                    step_max = std::max(step_max, (uint32_t)count_lane(l));
                }
                st.count_steps += step_max;
                passes++;
            }
            st.passes += passes;
            st.pass_hist[std::min(passes, 65)]++;
            st.max_passes = std::max<uint64_t>(st.max_passes, passes);
            // the valid lanes, their regions of the output
            int n_valid = 0;
            uint64_t reg[65];
            uint64_t o = at;
            for (int l = 0; l < n_lanes && !dead[l]; ++l) {
                if (flag[l] == FL_ERR) { fprintf(stderr, "error flag on a valid lane\n"); return false; }
                reg[l] = o;
                o += nout[l];
                n_valid = l + 1;
                if (flag[l] == FL_EOB) { eob = true; break; }
            }
            reg[n_valid] = o;
            if (o > want.size()) { fprintf(stderr, "overrun\n"); return false; }
            st.lanes_used += n_valid;
            // ---- the writing pass: every lane decodes its stretch once more, stores its literals where they belong and queues
            // its matches; then the matches of the tile are made in output order by a window of 64 of them (one per lane, token k
            // in lane k % 64): a match may be made when its source lies below the first byte that is still to be written (the
            // destination of the oldest match not made yet), the oldest one always; long ones by the whole wave together.
            struct Tok { uint64_t dest; uint32_t len, dist; int64_t src; bool fin; };
            std::vector<Tok> q;
            uint32_t ideal = 0;
            for (int l = 0; l < n_valid; ++l) {
                ideal = std::max(ideal, nsym[l]);
                uint64_t p = start[l], w = reg[l], taint_end = reg[l];
                const uint64_t sub_end = B + (uint64_t)SUB * (l + 1);
                while (p < sub_end) {
                    const Sym y = decode_one(b, L, D, p, end_bits);
                    st.symbols++;
                    if (y.kind == 2) break;
                    if (y.kind == 3) return false;
                    if (y.kind == 0) { out[w] = (uint8_t)y.lit; written[w] = 1; ++w; st.literals++; continue; }
                    if (y.dist > w) { fprintf(stderr, "distance beyond the start\n"); return false; }
                    st.matches++; st.match_bytes += y.len;
                    if (y.dist < 8) st.short_dist++;
                    if (self_resolve && w - y.dist >= taint_end) { // its source: this lane's own literals (and copies made the same way)
                        for (uint32_t i = 0; i < y.len; ++i) { out[w + i] = out[w + i - y.dist]; written[w + i] = 1; }
                        st.self_resolved++; st.self_pieces += (y.len + 15) / 16;
                        w += y.len;
                        continue;
                    }
                    q.push_back(Tok{w, y.len, y.dist, (int64_t)w - (int64_t)y.dist, false});
                    w += y.len;
                    taint_end = w;
                }
                if (w != reg[l + 1]) { fprintf(stderr, "a lane did not fill its region\n"); return false; }
            }
            st.ideal_turns += ideal;
            st.turns += ideal;
            st.max_queue = std::max<uint64_t>(st.max_queue, q.size());
            {
                const size_t nq = q.size();
                std::vector<uint8_t> written_before;
                if (FWD) {
                    // a match whose source lies inside ONE earlier match of the tile that does not overlap itself copies from that
                    // match's source instead; all matches at once, until nothing moves
                    int it = 0;
                    for (;; ++it) {
                        bool moved = false;
                        std::vector<int64_t> ns(nq);
                        for (size_t t = 0; t < nq; ++t) {
                            ns[t] = q[t].src;
                            if (q[t].fin || q[t].dist < q[t].len) continue; // (periodic copies keep their source)
                            const int64_t s0 = q[t].src, s1 = s0 + q[t].len;
                            if (s1 <= (int64_t)reg[0]) { q[t].fin = true; continue; }
                            // the earlier match that holds s0
                            size_t lo = 0, hi = t;
                            while (lo < hi) { const size_t mid = (lo + hi) / 2; if ((int64_t)(q[mid].dest + q[mid].len) <= s0) lo = mid + 1; else hi = mid; }
                            if (lo < t && (int64_t)q[lo].dest <= s0 && s1 <= (int64_t)(q[lo].dest + q[lo].len) && q[lo].dist >= q[lo].len) {
                                ns[t] = q[lo].src + (s0 - (int64_t)q[lo].dest);
                                moved = true;
                            } else q[t].fin = true;
                        }
                        for (size_t t = 0; t < nq; ++t) q[t].src = ns[t];
                        if (!moved) break;
                    }
                    st.fwd_iters += it + 1;
                    for (size_t t = 0; t < nq; ++t) q[t].dist = (uint32_t)((int64_t)q[t].dest - q[t].src);
                }
                std::vector<size_t> tok(WIN);
                std::vector<uint32_t> left(WIN);
                std::vector<char> pend(WIN);
                for (int i = 0; i < WIN; ++i) { tok[i] = (size_t)i; pend[i] = tok[i] < nq; left[i] = pend[i] ? q[tok[i]].len : 0; }
                for (;;) {
                    size_t base = (size_t)-1;
                    for (int i = 0; i < WIN; ++i) if (pend[i]) base = std::min(base, tok[i]);
                    if (base == (size_t)-1) break;
                    const Tok &bt = q[base];
                    const int bl = (int)(base % (size_t)WIN);
                    const uint64_t hwm = bt.dest + (bt.len - left[bl]);
                    st.rounds++;
                    if (left[bl] > (uint32_t)coop_min && (!exact_dep || left[bl] > 64u)) { // the whole wave on the oldest match
                        const uint64_t d0 = hwm;
                        for (uint32_t i = 0; i < left[bl]; ++i) {
                            if (!written[d0 + i - bt.dist]) { fprintf(stderr, "long copy of a byte that is not there\n"); return false; }
                            out[d0 + i] = out[d0 + i - bt.dist]; written[d0 + i] = 1;
                        }
                        st.coop++; st.coop_steps += (left[bl] + 63) / 64;
                        left[bl] = 0;
                    } else {
                        int active = 0;
                        if (exact_dep) written_before = written; // (what was there when the round began)
                        for (int i = 0; i < WIN; ++i) {
                            if (!pend[i]) continue;
                            const Tok &t = q[tok[i]];
                            const uint64_t d = t.dest + (t.len - left[i]);
                            const uint32_t piece = std::min<uint32_t>(left[i], (uint32_t)piece_max);
                            const uint64_t src = d - t.dist, src_end = src + std::min(piece, t.dist);
                            if (exact_dep) {
                                bool ok = true;
                                for (uint64_t x = src; x < src_end && ok; ++x) ok = written_before[x] != 0;
                                if (!ok) continue;
                            } else {
                                if (!(tok[i] == base || src_end <= hwm)) continue;
                                if (tok[i] != base && left[i] > (uint32_t)coop_min) continue; // (long ones wait for their turn as the oldest)
                            }
                            for (uint32_t k = 0; k < piece; ++k) {
                                if (!written[d + k - t.dist]) { fprintf(stderr, "copy of a byte that is not there (block %llu)\n", (unsigned long long)st.blocks); return false; }
                                out[d + k] = out[d + k - t.dist]; written[d + k] = 1;
                            }
                            left[i] -= piece;
                            ++active;
                        }
                        st.active += active;
                        st.copy_rounds++;
                    }
                    for (int i = 0; i < WIN; ++i)
                        if (pend[i] && left[i] == 0) { tok[i] += WIN; pend[i] = tok[i] < nq; if (pend[i]) left[i] = q[tok[i]].len; }
                }
            }
            at = o;
            pos = endp[n_valid - 1];
        }
    }
    st.out_bytes += at;
    if (at != want.size() || memcmp(out.data(), want.data(), at) != 0) { fprintf(stderr, "output differs from zlib's\n"); return false; }
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [sub_bits] [max_blocks] [piece_max]\n", argv[0]); return 2; }
    if (argc > 2) SUB = atoi(argv[2]);
    const size_t max_blocks = argc > 3 ? (size_t)atoll(argv[3]) : (size_t)-1;
    const int piece_max = argc > 4 ? atoi(argv[4]) : 8;
    if (argc > 5) coop_min = atoi(argv[5]);
    if (getenv("EXACT")) exact_dep = atoi(getenv("EXACT"));
    if (getenv("WIN")) WIN = atoi(getenv("WIN"));
    if (getenv("SELF")) self_resolve = atoi(getenv("SELF"));
    if (getenv("MERGE_K")) MERGE_K = atoi(getenv("MERGE_K"));
    if (getenv("FWD")) FWD = atoi(getenv("FWD"));
    if (getenv("MERGE_FROM")) MERGE_FROM = atoi(getenv("MERGE_FROM"));
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t fsize = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file(fsize + 16);
    if (fread(file.data(), 1, fsize, f) != fsize) return 1;
    fclose(f);
    Stats st;
    size_t at = 0, nb = 0;
    while (at + 18 <= fsize && nb < max_blocks) {
        const uint8_t *h = file.data() + at;
        if (h[0] != 0x1f || h[1] != 0x8b) { fprintf(stderr, "not a BGZF block at %zu\n", at); return 1; }
        const size_t bsize = (size_t)(h[16] | h[17] << 8) + 1;
        const uint8_t *d = h + 18;
        const size_t dlen = bsize - 26;
        const uint32_t isize = (uint32_t)h[bsize - 4] | (uint32_t)h[bsize - 3] << 8 | (uint32_t)h[bsize - 2] << 16 | (uint32_t)h[bsize - 1] << 24;
        std::vector<uint8_t> want(isize);
        if (isize) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = const_cast<Bytef *>(d); zs.avail_in = (uInt)dlen; zs.next_out = want.data(); zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) { fprintf(stderr, "zlib: block %zu does not inflate\n", nb); return 1; }
            std::vector<uint8_t> padded(d, d + dlen);
            padded.resize(dlen + 16, 0);
            if (!wave_inflate(padded.data(), dlen, want, st, piece_max)) { fprintf(stderr, "block %zu (at %zu) FAILED\n", nb, at); return 1; }
        }
        at += bsize;
        nb++;
    }
    printf("%llu blocks, %llu deflate blocks (%llu stored), %.1f KB in / %.1f KB out per block (ratio %.2f)\n", (unsigned long long)st.blocks, (unsigned long long)st.dblocks,
           (unsigned long long)st.stored, st.in_bytes / 1e3 / st.blocks, st.out_bytes / 1e3 / st.blocks, (double)st.out_bytes / st.in_bytes);
    printf("symbols %.0f per block: %.1f %% literals, matches avg %.1f bytes, %.1f %% of them at distance < 8; %.2f bytes per symbol; header symbols %.0f per block\n",
           (double)st.symbols / st.blocks, 100.0 * st.literals / st.symbols, st.matches ? (double)st.match_bytes / st.matches : 0.0,
           st.matches ? 100.0 * st.short_dist / st.matches : 0.0, (double)st.out_bytes / st.symbols, (double)st.header_syms / st.blocks);
    printf("sub = %d bits: %.2f tiles per block, %.1f valid lanes per tile; boundary search %.2f passes per tile (max %llu), %.1f decode steps per tile\n", SUB,
           (double)st.tiles / st.blocks, (double)st.lanes_used / st.tiles, (double)st.passes / st.tiles, (unsigned long long)st.max_passes, (double)st.count_steps / st.tiles);
    printf("writing pass: %.1f decode turns per tile; matches: %.1f per tile (max %llu), %.1f rounds per tile = %.1f short-copy rounds (%.1f lanes active) + %.1f whole-wave copies (%.2f steps each)\n",
           (double)st.turns / st.tiles, (double)st.matches / st.tiles, (unsigned long long)st.max_queue, (double)st.rounds / st.tiles, (double)st.copy_rounds / st.tiles,
           st.copy_rounds ? (double)st.active / st.copy_rounds : 0.0, (double)st.coop / st.tiles, st.coop ? (double)st.coop_steps / st.coop : 0.0);
    printf("self-resolved in the decode pass: %.1f %% of matches (%.2f pieces each)\n", st.matches ? 100.0 * st.self_resolved / st.matches : 0.0, st.self_resolved ? (double)st.self_pieces / st.self_resolved : 0.0);
    printf("wave-steps per block: search %.0f + decode-and-store %.0f + copy rounds %.0f\n", (double)st.count_steps / st.blocks, (double)st.turns / st.blocks, (double)st.rounds / st.blocks);
    printf("source forwarding: %.2f iterations per tile\n", (double)st.fwd_iters / st.tiles);
    printf("recounts that merged with the lane's earlier chain: %llu, that did not: %llu\n", (unsigned long long)st.merged, (unsigned long long)st.unmerged);
    printf("passes histogram:");
    for (int i = 1; i < 66; ++i) if (st.pass_hist[i]) printf(" %d:%llu", i, (unsigned long long)st.pass_hist[i]);
    printf("\n");
    return 0;
}
