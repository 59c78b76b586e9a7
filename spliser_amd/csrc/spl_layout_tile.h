// spl_layout_tile.h -- the layout of one chunk: a workgroup's reads out of the BAM-native arrays, classified once, to their places in
// the chunk's four runs (spl_pack.h).  The body of spl_layout_kernel (spl_devpack.hip: records to memory) and of the fused
// counting kernel's first phase (spl_kernels.hip: records into LDS, counted from there, never written) -- one source for both.
#ifndef SPL_LAYOUT_TILE_H
#define SPL_LAYOUT_TILE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spl_pack.h"
#include "spl_devpack.h"

namespace spllay {

typedef uint32_t lay_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lay_u32x4 __attribute__((ext_vector_type(4)));
typedef lay_u32x2 lay_u32x2_a4 __attribute__((aligned(4)));
typedef lay_u32x4 lay_u32x4_a8 __attribute__((aligned(8)));
typedef __attribute__((address_space(3))) uint32_t lay_lds_w32;   // (LDS, typed: a generic pointer makes every access a flat one)
typedef __attribute__((address_space(3))) int32_t lay_lds_i32;
typedef __attribute__((address_space(3))) uint8_t lay_lds_u8;
typedef __attribute__((address_space(3))) uint16_t lay_lds_u16;

// A read's ops out of the workgroup's stage in LDS: its first SPL_PACK_SCAN_OPS words lie inside it whatever the number of its ops
// (what a short CIGAR's accessor gives back beyond them is the neighbour's, and ignored).  Typed by address space: through a
// generic pointer every read would be a flat load.
typedef __attribute__((address_space(3))) const uint32_t lay_lds_u32;
struct StagedOps {
    static constexpr bool padded = true;
    lay_lds_u32 *p;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return p[k]; }
};
// Inclusive prefix sum inside every row of 16 lanes.
__device__ __forceinline__ uint32_t row_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

// Inclusive prefix sum over the 64 lanes in six DPP adds: shifts by 1, 2, 4, 8 inside the rows of 16, then row 0's total into row
// 1 and row 2's into row 3 (row_bcast:15), then the first half's into the second (row_bcast:31).
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}


// Where a chunk's records go.  In memory: the chunk's slot (spl_layout_kernel).
struct RecordsInMemory {
    static constexpr bool wants_index = false;
    static constexpr bool clip_tier = false;   // (the layout kernel lives on its 64 registers: the clips' straight-line path would spill six of them)
    uint8_t *rec_base;
    size_t slot_bytes;
    __device__ __forceinline__ uint8_t *rec() const { return rec_base + (size_t)blockIdx.x * slot_bytes; } // (not kept across the classification: two scalar registers fewer there)
    __device__ __forceinline__ void st2(uint32_t d, lay_u32x2 v) const { *(lay_u32x2 *)(rec() + d) = v; }
    __device__ __forceinline__ void st4(uint32_t d, lay_u32x4 v) const { *(lay_u32x4_a8 *)(rec() + d) = v; }
    __device__ __forceinline__ void index(uint32_t, uint32_t) const {}
};
// In LDS (the fused counting kernel): the same bytes at the same offsets, and for every read that is not a simple one the place
// it has in the arrays (read q of the runs behind the simple one -> its index in the chunk's cell), which is what the literal
// kernel is told when such a read is handed on.
struct RecordsInLds {
    static constexpr bool wants_index = true;
    static constexpr bool clip_tier = true;
    lay_lds_u8 *rec;
    lay_lds_u16 *idx;
    uint32_t base;         // of the tile in the chunk's cell
    __device__ __forceinline__ void st2(uint32_t d, lay_u32x2 v) const { *(__attribute__((address_space(3))) lay_u32x2 *)(rec + d) = v; }
    __device__ __forceinline__ void st4(uint32_t d, lay_u32x4 v) const { *(__attribute__((address_space(3))) lay_u32x4_a8 *)(rec + d) = v; }
    __device__ __forceinline__ void index(uint32_t q, uint32_t cell_index) const { idx[q] = (uint16_t)(base + cell_index); }
};

// One tile of C reads -- the cell [cell0, cell0 + C) of the arrays' indexes, or the part [lo, hi) of it a segment has -- by a
// workgroup of C / 4 threads, in two steps so that a kernel can ask for its NEXT tile's reads before it works on this one's:
//   tile_issue   the thread's four reads (one load per array) and its share of the tile's ops cigar[o_lo, o_hi) (the first STAGE
//                words of them), everything asked for at once, into registers;
//   tile_finish  the ops into s_ops (STAGE = 4 C words of LDS), classification, ranks (s_cnt: 2 words a wave), the records to
//                `sink` -- AFTER the workgroup's last look at s_ops (a barrier lies between): the sink's memory may be s_ops itself.
// n[r] = reads of run r; the runs begin at 0, align16(8 n0), + 16 n1, + 24 n2 (chunk_view's arithmetic).
struct TileSpan { int64_t cell0, lo, hi; uint32_t o_lo, o_hi, o_fetch_hi, seg_op0; }; // (o_fetch_hi >= o_hi: how far tile_issue may read ops)
template <int C, int R>   // R = reads a thread: 4 (a workgroup of C / 4 threads) or 2 (C / 2)
struct TileLoads {
    int32_t pos[R];
    uint32_t fw[R / 2];    // the flags, two a word, as loaded
    uint32_t co[R + 1];
    lay_u32x4 v[R];        // STAGE / (4 T) = R quads of ops a thread
};

template <int C, int R>
__device__ __forceinline__ void tile_issue(const spl_devreads &src, int64_t n_rec, int64_t n_ops, const TileSpan &sp, TileLoads<C, R> &L)
{
    static_assert(R == 4 || R == 2, "two or four reads a thread");
    constexpr uint32_t T = C / R, STAGE = 4 * C;
    static_assert(STAGE / (4 * T) == R, "R quads of ops a thread");
    const uint32_t t = threadIdx.x;
    const int64_t g = sp.cell0 + R * (int64_t)t; // the thread's first read
#pragma unroll
    for (int j = 0; j < R; ++j) L.pos[j] = 0;
#pragma unroll
    for (int j = 0; j <= R; ++j) L.co[j] = 0;
#pragma unroll
    for (int j = 0; j < R / 2; ++j) L.fw[j] = 0;
    // (places in the cell are 32-bit: the span's bounds relative to the cell are uniform, a thread's reads R t .. R t + R - 1)
    const uint32_t r0 = (uint32_t)R * t, lo_r = (uint32_t)(sp.lo - sp.cell0), hi_r = (uint32_t)(sp.hi - sp.cell0);
    const int64_t left = n_rec - sp.cell0;
    const uint32_t rec_r = left < (int64_t)C ? (uint32_t)left : (uint32_t)C; // reads the arrays have from the cell's first on, at most C
    const bool mine = r0 + R > lo_r && r0 < hi_r; // (a partial cell's threads outside the segment load nothing)
    if (mine) {
        if (r0 + R <= rec_r) {
            if constexpr (R == 4) {
                const lay_u32x4 pv = *(const lay_u32x4 *)(src.pos + g);
                const lay_u32x2 fv = *(const lay_u32x2 *)(src.flag + g);
                const lay_u32x4 cv = *(const lay_u32x4 *)(src.cig_off + g);
                L.co[4] = src.cig_off[g + 4];
                L.pos[0] = (int32_t)pv.x; L.pos[1] = (int32_t)pv.y; L.pos[2] = (int32_t)pv.z; L.pos[3] = (int32_t)pv.w;
                L.fw[0] = fv.x; L.fw[1] = fv.y;
                L.co[0] = cv.x; L.co[1] = cv.y; L.co[2] = cv.z; L.co[3] = cv.w;
            } else {
                const lay_u32x2 pv = *(const lay_u32x2 *)(src.pos + g);
                L.fw[0] = *(const uint32_t *)(src.flag + g);
                const lay_u32x2 cv = *(const lay_u32x2 *)(src.cig_off + g);
                L.co[2] = src.cig_off[g + 2];
                L.pos[0] = (int32_t)pv.x; L.pos[1] = (int32_t)pv.y;
                L.co[0] = cv.x; L.co[1] = cv.y;
            }
        } else { // (the arrays' last reads: one by one)
            uint32_t f[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int64_t i = g + j;
                f[j] = 0;
                if (i < n_rec) { L.pos[j] = src.pos[i]; f[j] = src.flag[i]; L.co[j] = src.cig_off[i]; L.co[j + 1] = src.cig_off[i + 1]; }
                else L.co[j + 1] = L.co[j];
            }
#pragma unroll
            for (int j = 0; j < R / 2; ++j) L.fw[j] = f[2 * j] | (f[2 * j + 1] << 16);
        }
    }
    const uint64_t ws = sp.o_lo & ~3u;
    const uint64_t o_end = (uint64_t)sp.o_fetch_hi < ws + STAGE ? (uint64_t)sp.o_fetch_hi : ws + STAGE;
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)R; ++q) {
        const uint64_t at = ws + 4ull * (t + q * T);
        L.v[q] = lay_u32x4{0u, 0u, 0u, 0u};
        if (at < o_end) {
            if (at + 4 <= (uint64_t)n_ops) L.v[q] = *(const lay_u32x4 *)(src.cigar + at);
            else {
                L.v[q].x = src.cigar[at];
                if (at + 1 < (uint64_t)n_ops) L.v[q].y = src.cigar[at + 1];
                if (at + 2 < (uint64_t)n_ops) L.v[q].z = src.cigar[at + 2];
            }
        }
    }
}

template <int C, int R, class Sink>
__device__ __forceinline__ void tile_finish(const spl_devreads &src, int64_t n_ops, const TileSpan &sp, const TileLoads<C, R> &L, lay_lds_w32 *s_ops,
                                            lay_lds_w32 *s_cnt, const Sink sink, uint32_t (&n)[4])
{
    constexpr uint32_t T = C / R, NW = T / 64, STAGE = 4 * C; // (threads, waves, words of the op stage: 4 ops a read on average)
    static_assert(NW <= 16, "the waves' totals are summed inside one row of lanes");
    constexpr uint32_t PAD = SPL_PACK_SCAN_OPS;               // a read is classified from a stage that holds its first eight ops
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const int64_t g = sp.cell0 + R * (int64_t)t;
    const uint32_t r0 = (uint32_t)R * t, lo_r = (uint32_t)(sp.lo - sp.cell0), hi_r = (uint32_t)(sp.hi - sp.cell0);
    const uint32_t o_hi = sp.o_hi, seg_op0 = sp.seg_op0;
    int32_t pos[R];
    uint32_t flag[R], co[R + 1];
#pragma unroll
    for (int j = 0; j < R; ++j) { pos[j] = L.pos[j]; flag[j] = (j & 1) ? L.fw[j / 2] >> 16 : L.fw[j / 2] & 0xffffu; }
#pragma unroll
    for (int j = 0; j <= R; ++j) co[j] = L.co[j];
    // The tile's ops into LDS, 16 bytes a lane and load, a WINDOW of STAGE words at a time: one window for all but long-read
    // CIGARs (more than four ops a read on average), whose tiles take several -- a read is classified from the window that
    // holds its first eight ops (more are never looked at: such a read is WIDE, its ops stay where they are).
    auto fill = [&](uint64_t ws) {
        const uint64_t o_end = (uint64_t)o_hi < ws + STAGE ? (uint64_t)o_hi : ws + STAGE;
        lay_u32x4 v[R];
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)R; ++q) {
            const uint64_t at = ws + 4ull * (t + q * T);
            v[q] = lay_u32x4{0u, 0u, 0u, 0u};
            if (at < o_end) {
                if (at + 4 <= (uint64_t)n_ops) v[q] = *(const lay_u32x4 *)(src.cigar + at);
                else {
                    v[q].x = src.cigar[at];
                    if (at + 1 < (uint64_t)n_ops) v[q].y = src.cigar[at + 1];
                    if (at + 2 < (uint64_t)n_ops) v[q].z = src.cigar[at + 2];
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)R; ++q) *(__attribute__((address_space(3))) lay_u32x4 *)(s_ops + 4u * (t + q * T)) = v[q];
    };
    uint64_t ws = sp.o_lo & ~3u;
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)R; ++q) *(__attribute__((address_space(3))) lay_u32x4 *)(s_ops + 4u * (t + q * T)) = L.v[q];
    __syncthreads();

    // ---- classify: four records in registers.  Straight-line, no branch, the four reads' chains side by side: in the fused kernel
    // the three plain shapes from the op codes alone (classify_plain), then -- a wave that has anything else -- every class of
    // at most five consuming ops between at most one clip either side (classify_clipped, below); in the layout kernel the reads of
    // at most five ops that all consume the reference (classify_fast5).  What is left -- insertions, clips on clips, long CIGARs,
    // ops beyond the first window -- is pending and done by the lanes that hold it, through the general classifier.
    uint32_t w[R][6];
    uint32_t runs = 0, pend = 0; // run of read j: bits 3j .. 3j + 2 (4 = no read); pending: bit j
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool valid = r0 + (uint32_t)j >= lo_r && r0 + (uint32_t)j < hi_r;
        const uint32_t rel0 = co[j] - (uint32_t)ws;
        const bool inside = rel0 + PAD <= STAGE;
        lay_lds_u32 *o = (lay_lds_u32 *)s_ops + (inside ? rel0 : 0u);
        splrec::Rec r;
        bool fast;
        if constexpr (Sink::clip_tier) fast = splrec::classify_plain(pos[j], flag[j], o[0], o[1], o[2], o[3], o[4], co[j + 1] - co[j], r); // (what is not one of the three plain shapes: the next tier)
        else fast = splrec::classify_fast5(pos[j], flag[j], o[0], o[1], o[2], o[3], o[4], co[j + 1] - co[j], co[j] - seg_op0, r);
        pend |= (valid && !(fast && inside) ? 1u : 0u) << j;
        runs |= (valid ? r.run : (uint32_t)SPL_RC_RUNS) << (3 * j);
#pragma unroll
        for (int q = 0; q < 6; ++q) w[j][q] = r.w[q];
        __builtin_amdgcn_sched_barrier(0); // (one read after the other: four chains side by side do not fit the 64 registers that keep two workgroups on a CU)
    }
    // ---- reads with ONE op in front and / or behind that does not consume the reference (a local aligner's soft and hard clips:
    // a sizeable share of a real library's reads): the same straight-line decision from the ops between the clips
    // (classify_clipped), taken by a wave only if it has such a read at all.  What is still pending afterwards -- insertions,
    // clips on clips, long CIGARs -- goes through the general classifier below.
    if (Sink::clip_tier && __any(pend != 0u)) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t rel0 = co[j] - (uint32_t)ws;
            const bool inside = rel0 + PAD <= STAGE;
            splrec::Rec r;
            const bool fast = splrec::classify_clipped(pos[j], flag[j], StagedOps{(lay_lds_u32 *)s_ops + (inside ? rel0 : 0u)}, co[j + 1] - co[j], co[j] - seg_op0, r);
            const bool take = ((pend >> j) & 1u) != 0u && fast && inside;
            runs = take ? (runs & ~(7u << (3 * j))) | (r.run << (3 * j)) : runs;
            pend = take ? pend & ~(1u << j) : pend;
#pragma unroll
            for (int q = 0; q < 6; ++q) w[j][q] = take ? r.w[q] : w[j][q];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (;;) {
        if (__any(pend != 0u)) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if ((pend >> j) & 1u) {
                    // (the read's offsets come from memory again, its POS and flag out of the record's first words: nothing but the
                    //  records stays in registers across this branch, which most waves never take)
                    const uint32_t c0 = src.cig_off[g + j], c1 = src.cig_off[g + j + 1], rel0 = c0 - (uint32_t)ws;
                    if (rel0 + PAD <= STAGE) {
                        splrec::Rec r;
                        splrec::classify_lean((int32_t)w[j][0], w[j][1] & 0xffffu, StagedOps{(lay_lds_u32 *)s_ops + rel0}, c1 - c0, c0 - seg_op0, r);
                        runs = (runs & ~(7u << (3 * j))) | (r.run << (3 * j));
                        pend &= ~(1u << j);
#pragma unroll
                        for (int q = 0; q < 6; ++q) w[j][q] = r.w[q];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ws + STAGE >= (uint64_t)o_hi + PAD) break; // (wave-uniform: the chunk's ops, and eight words behind them, were in this window)
        ws += STAGE - PAD;
        __syncthreads();
        fill(ws);
        __syncthreads();
    }
    uint32_t c01 = 0, c23 = 0; // reads per run: run 0 | run 1 << 16, run 2 | run 3 << 16
    uint32_t run[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        run[j] = (runs >> (3 * j)) & 7u;
        c01 += run[j] == SPL_RC_SIMPLE ? 1u : (run[j] == SPL_RC_MNM ? 0x10000u : 0u);
        c23 += run[j] == SPL_RC_M2 ? 1u : (run[j] == SPL_RC_OTHER ? 0x10000u : 0u);
    }

    // ---- ranks: prefix sums over the lanes (DPP: row shifts, then the rows' totals broadcast), the waves' totals through LDS
    const uint32_t i01 = wave_scan(c01), i23 = wave_scan(c23);
    if (lane == 63u) { s_cnt[2u * wave] = i01; s_cnt[2u * wave + 1u] = i23; }
    __syncthreads();
    // the waves' totals: lane q of every wave takes wave q's, a prefix sum over those NW lanes (one row of 16: four DPP adds), and
    // the sums come out by lane number -- the wave's own number is uniform
    const uint32_t x = row_scan(lane < NW ? s_cnt[2u * (lane < NW ? lane : 0u)] : 0u), y = row_scan(lane < NW ? s_cnt[2u * (lane < NW ? lane : 0u) + 1u] : 0u);
    const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)x, NW - 1), t23 = (uint32_t)__builtin_amdgcn_readlane((int)y, NW - 1);
    const uint32_t b01 = wu ? (uint32_t)__builtin_amdgcn_readlane((int)x, wu - 1u) : 0u, b23 = wu ? (uint32_t)__builtin_amdgcn_readlane((int)y, wu - 1u) : 0u;
    const uint32_t n0 = t01 & 0xffffu, n1 = t01 >> 16, n2 = t23 & 0xffffu, n3 = t23 >> 16;
    const uint32_t off1 = (n0 * SPL_REC_SIMPLE + 15u) & ~15u, off2 = off1 + n1 * SPL_REC_MNM, off3 = off2 + n2 * SPL_REC_M2;
    // (reads of each run before this thread's: the waves below, the lanes below) -> byte offsets of the thread's next record of each run
    const uint32_t e01 = b01 + i01 - c01, e23 = b23 + i23 - c23;
    uint32_t at[4] = {(e01 & 0xffffu) * SPL_REC_SIMPLE, off1 + (e01 >> 16) * SPL_REC_MNM, off2 + (e23 & 0xffffu) * SPL_REC_M2, off3 + (e23 >> 16) * SPL_REC_OTHER};
    [[maybe_unused]] uint32_t q_at[3] = {e01 >> 16, n1 + (e23 & 0xffffu), n1 + n2 + (e23 >> 16)}; // (Sink::wants_index: the thread's next read of runs 1 .. 3 among those runs' reads)

    // ---- the records, each to its place in its run (the slot's address is uniform, the place a 32-bit offset)
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t r = run[j];
        const uint32_t d = r == SPL_RC_SIMPLE ? at[0] : (r == SPL_RC_MNM ? at[1] : (r == SPL_RC_M2 ? at[2] : at[3]));
        if (r == SPL_RC_SIMPLE) sink.st2(d, lay_u32x2{w[j][0], w[j][1]});
        else if (r < (uint32_t)SPL_RC_RUNS) {
            sink.st4(d, lay_u32x4{w[j][0], w[j][1], w[j][2], w[j][3]});
            if (r != SPL_RC_MNM) sink.st2(d + 16, lay_u32x2{w[j][4], w[j][5]});
            if constexpr (Sink::wants_index) {
                sink.index(r == SPL_RC_MNM ? q_at[0] : (r == SPL_RC_M2 ? q_at[1] : q_at[2]), (uint32_t)R * t + (uint32_t)j);
                q_at[0] += r == SPL_RC_MNM ? 1u : 0u;
                q_at[1] += r == SPL_RC_M2 ? 1u : 0u;
                q_at[2] += r == SPL_RC_OTHER ? 1u : 0u;
            }
        }
        at[0] += r == SPL_RC_SIMPLE ? SPL_REC_SIMPLE : 0u;
        at[1] += r == SPL_RC_MNM ? SPL_REC_MNM : 0u;
        at[2] += r == SPL_RC_M2 ? SPL_REC_M2 : 0u;
        at[3] += r == SPL_RC_OTHER ? SPL_REC_OTHER : 0u;
    }
    n[0] = n0; n[1] = n1; n[2] = n2; n[3] = n3;
}

// Both steps in a row: one chunk ch of C reads (spl_layout_kernel).  *s_first = POS of the chunk's first read in file order (valid
// for every thread after the call).
template <int C, class Sink>
__device__ __forceinline__ void layout_tile(const spl_devreads &src, int64_t n_rec, int64_t n_ops, const spl_layout_chunk &ch, lay_lds_w32 *s_ops,
                                            lay_lds_w32 *s_cnt, lay_lds_i32 *s_first, const Sink sink, uint32_t (&n)[4])
{
    const TileSpan sp{ch.lo & ~(int64_t)(C - 1), ch.lo, ch.lo + ch.n, ch.o_lo, ch.o_hi, ch.o_hi, ch.seg_op0};
    TileLoads<C, 4> L;
    tile_issue<C, 4>(src, n_rec, n_ops, sp, L);
    const int64_t g = sp.cell0 + 4 * (int64_t)threadIdx.x;
    if (sp.lo >= g && sp.lo < g + 4) { // (a record's first word is its read's POS, whatever its class)
        const uint32_t e = (uint32_t)(sp.lo - g);
        *s_first = e == 0u ? L.pos[0] : (e == 1u ? L.pos[1] : (e == 2u ? L.pos[2] : L.pos[3]));
    }
    tile_finish<C, 4>(src, n_ops, sp, L, s_ops, s_cnt, sink, n);
}

} // namespace spllay
#endif
