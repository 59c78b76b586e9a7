// spl_kernels.hip -- CDNA4 (gfx950, wave64) kernels of the SpliSER `process` hot path.
//
//   spl_count_ranges_kernel  the checkBam loop (SpliSER_v0_1_8.py:408-559, called per site from processSites
//                            :686-688) turned inside out: every read walks its CIGAR ONCE and turns each op
//                            into a range of site rows -- an aligned block makes every row it covers (t and
//                            t+1) a beta1 read, an N op makes every row strictly inside it a
//                            mutually-exclusive (beta2Simple) read -- recorded as +1/-1 in LDS-privatised
//                            difference arrays.  Cost per read is O(ops), whatever the number of sites a
//                            500 kb intron spans.  Reads come as per-class records laid out by the host
//                            (spl_pack.h).  Only the sites whose outcome can depend on their own partner /
//                            competitor lists (rivals of the read's junction: the junction table) are
//                            corrected one by one, by the wave itself for once- and twice-spliced reads.
//   spl_count_literal_kernel the reads the range kernel queues (odd CIGAR shapes with a flagged junction end,
//                            records with flag 0x4, combine mode): table-driven walk, or the literal state machine
//                            (spl_classify.h).  Also clears the spare counter copy and takes the scan's block sums.
//   spl_scan_apply_kernel    prefix sums that turn the difference arrays into beta1 / beta2Simple counters, then
//                            findBeta2Counts + calculateSSE of every row (SpliSER_v0_1_8.py:581-639).
//   spl_count_pairs_kernel   the literal formulation: every (read, site) pair through spl_classify_pair.  An
//                            on-device cross-check of the range kernel in the tests; the product never selects it.
//   spl_junction_kernel      the junction table of a read set (what the README asks regtools for).
//   spl_sse_kernel           findBeta2Counts + calculateSSE alone, one site per lane (spl_sse on caller's counters).
//
// Integer / indexing work: no MFMA.  The roofline that bounds the count kernels is HBM.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define SPL_HD __device__ __forceinline__
#include "spl_classify.h"
#include "spl_device.h"
#include "spl_layout_tile.h"

namespace {

// First table row whose position is >= pos: a direct-address bucket index (bucket b covers positions
// [base + (b << shift), base + ((b+1) << shift))) narrows the search to the rows of one bucket, a short
// binary search finishes it.  bucket[] has n_buckets + 1 entries; bucket[n_buckets] == n_sites.
__device__ __forceinline__ int32_t first_site_at_or_after(const spl_count_params &p, int32_t pos)
{
    const int64_t rel = (int64_t)pos - (int64_t)p.bucket_base;
    uint32_t b = 0;
    if (rel > 0) {
        const int64_t q = rel >> p.bucket_shift;
        b = q >= (int64_t)p.n_buckets ? p.n_buckets : (uint32_t)q;
    }
    uint32_t lo = p.bucket[b];
    uint32_t hi = p.bucket[b < p.n_buckets ? b + 1 : b];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.site_pos[mid] < pos) lo = mid + 1; else hi = mid;
    }
    return (int32_t)lo;
}

// XCD-aware chunk order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one contiguous
// eighth of the (coordinate-sorted) reads -- its L2 then sees one moving window of the site table.
__device__ __forceinline__ uint32_t my_chunk()
{
    const uint32_t per = (gridDim.x + 7u) >> 3;
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

// ---- the packed read layout (spl_pack.h), device side ---------------------------------------------------

// A chunk's descriptor as wave-uniform values (the index is uniform: scalar loads).
// (the addresses arrive as integers: typed as GLOBAL pointers, or every load through them is a flat_load -- and a wave with a
//  flat load outstanding can only ever wait for ALL its memory operations, which is the end of any prefetching)
#define SPL_GLOBAL __attribute__((address_space(1)))
typedef SPL_GLOBAL const char spl_gchar;
typedef SPL_GLOBAL const uint32_t spl_gu32;
typedef uint32_t spl_u32x2 __attribute__((ext_vector_type(2))); // (built-in vectors: loadable from any address space)
typedef uint32_t spl_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 ld_g2(spl_gchar *p) { const spl_u32x2 v = *(SPL_GLOBAL const spl_u32x2 *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint4 ld_g4(spl_gchar *p) { const spl_u32x4 v = *(SPL_GLOBAL const spl_u32x4 *)p; return make_uint4(v.x, v.y, v.z, v.w); }
struct ChunkView {
    spl_gchar *rec;
    spl_gu32 *wide;
    int32_t shift, first_pos;
    uint32_t start[SPL_RC_RUNS + 1]; // slot (run order) of the first read of each run; start[4] = reads in the chunk
    uint32_t off[SPL_RC_RUNS];       // byte offset of each run in the record area
};

__device__ __forceinline__ ChunkView chunk_view(const spl_chunk_meta *meta)
{
    const uint4 a = *(const uint4 *)meta, b = *((const uint4 *)meta + 1);
    ChunkView v;
    v.rec = (spl_gchar *)(((uint64_t)a.y << 32) | a.x);
    v.wide = (spl_gu32 *)(((uint64_t)a.w << 32) | a.z);
    v.shift = (int32_t)b.x;
    v.first_pos = (int32_t)b.y;
    const uint32_t n0 = b.z & 0xffffu, n1 = b.z >> 16, n2 = b.w & 0xffffu, n3 = b.w >> 16;
    v.start[0] = 0; v.start[1] = n0; v.start[2] = n0 + n1; v.start[3] = n0 + n1 + n2; v.start[4] = n0 + n1 + n2 + n3;
    v.off[0] = 0;
    v.off[1] = (n0 * SPL_REC_SIMPLE + 15u) & ~15u;
    v.off[2] = v.off[1] + n1 * SPL_REC_MNM;
    v.off[3] = v.off[2] + n2 * SPL_REC_M2;
    return v;
}

// The read at `slot` (run order) of a chunk with its CIGAR in BAM form again -- for the literal formulations, which walk ops
// (spl_classify.h).  Short CIGARs come back without their non-consuming ops (they change nothing for checkBam, :457-464), into
// `row` (5 words the caller owns: LDS or private); a wide read's ops are where the packer put them, all of them.
struct ReadView { int32_t pos; uint32_t flag, n_ops; const uint32_t *ops; bool neg; };

__device__ __forceinline__ ReadView read_at(const ChunkView &cv, uint32_t slot, uint32_t *row)
{
    ReadView v;
    v.ops = row;
    int32_t pos0;
    if (slot < cv.start[1]) {
        const uint2 r = ld_g2(cv.rec + (size_t)(cv.off[0] + SPL_REC_SIMPLE * slot));
        pos0 = (int32_t)r.x; v.flag = r.y & 0xffffu;
        row[0] = (r.y >> 16) << 4;
        v.n_ops = 1u;
    } else if (slot < cv.start[2]) {
        const uint4 r = ld_g4(cv.rec + (size_t)(cv.off[1] + SPL_REC_MNM * (slot - cv.start[1])));
        pos0 = (int32_t)r.x; v.flag = r.y & 0xffffu;
        row[0] = (r.y >> 16) << 4; row[1] = (r.z << 4) | (uint32_t)SPL_OP_N; row[2] = r.w << 4;
        v.n_ops = 3u;
    } else {
        const bool m2 = slot < cv.start[3];
        spl_gchar *q = cv.rec + (size_t)(m2 ? cv.off[2] + SPL_REC_M2 * (slot - cv.start[2]) : cv.off[3] + SPL_REC_OTHER * (slot - cv.start[3]));
        const uint2 r0 = ld_g2(q), r1 = ld_g2(q + 8), r2 = ld_g2(q + 16);
        pos0 = (int32_t)r0.x; v.flag = r0.y & 0xffffu;
        if (m2) {
            row[0] = (r0.y >> 16) << 4; row[1] = (r1.x << 4) | (uint32_t)SPL_OP_N; row[2] = (r1.y & 0xffffu) << 4;
            row[3] = (r2.x << 4) | (uint32_t)SPL_OP_N; row[4] = (r1.y >> 16) << 4;
            v.n_ops = 5u;
        } else {
            v.n_ops = r2.y;
            if ((r0.y >> SPL_RC_SHIFT) == SPL_RC_WIDE) v.ops = (const uint32_t *)(cv.wide + r2.x);
            else { row[0] = r1.x; row[1] = r1.y; row[2] = r2.x; }
        }
    }
    v.neg = pos0 < 0;
    v.pos = pos0 + cv.shift;
    return v;
}

// ---- literal (read, site) pair: shared by the pair kernel and by the slow paths of the range kernel ------

// Counter updates of one classified pair, straight to HBM (SpliSER_v0_1_8.py:519-559).
__device__ __forceinline__ void apply_pair_global(const spl_count_params &p, const spl_pair &r, int32_t s, int32_t pos,
                                                  const uint32_t *ops, uint32_t n_ops, const int32_t *part,
                                                  uint32_t n_part, uint32_t part_off)
{
    switch (r.cls) {
    case SPL_CLS_BETA1: atomicAdd(&p.beta1[s], 1u); break;
    case SPL_CLS_ME: atomicAdd(&p.beta2s_reads[s], 1u); break;
    case SPL_CLS_FLANK: if (p.combine_mode) atomicAdd(&p.beta2s_reads[s], 1u); break;
    case SPL_CLS_B1TYPE: atomicAdd(&p.beta2s_reads[s], 1u); [[fallthrough]];
    case SPL_CLS_ALPHA_COMP:
        for (uint32_t e = 0; e < n_part; ++e) { // PartnerBeta2DoubleCounts (:519-527, :544-551)
            const int32_t pp = part[e];
            if (r.cls == SPL_CLS_ALPHA_COMP && r.has_partner_used && pp == r.partner_used) continue;
            if (spl_read_splices_at(pos, ops, n_ops, pp)) atomicAdd(&p.dbl[part_off + e], 1u);
        }
        break;
    default: break;
    }
}

// One counter increment: LDS-privatised when the row falls in this workgroup's window, global otherwise.
__device__ __forceinline__ void bump(uint32_t *lds_cnt, uint32_t *glob, int32_t s, int32_t wbase)
{
    const uint32_t loc = (uint32_t)(s - wbase);
    if (loc < (uint32_t)SPL_WIN) atomicAdd(&lds_cnt[loc], 1u);
    else atomicAdd(&glob[s], 1u);
}

template <bool STRANDED>
__device__ __forceinline__ void do_pair(const spl_count_params &p, uint32_t *lds, int32_t wbase, int32_t s, int32_t t,
                                        int32_t pos, const uint32_t *ops, uint32_t n_ops, bool has_n, uint8_t rstrand)
{
    bool strand_ok = true;
    if (STRANDED) strand_ok = (p.site_strand[s] == rstrand);
    const int32_t *part = nullptr, *comp = nullptr;
    uint32_t n_part = 0, n_comp = 0, part_off = 0;
    if (has_n) {
        const uint4 m = p.site_meta[s];
        part_off = m.x;
        if (m.w != 0u) { // without competitors compSplicing can never be set (:494-501): lists not needed
            n_part = m.y;
            n_comp = m.w;
            part = p.part_pos + m.x;
            comp = p.comp_pos + m.z;
        }
    }
    const spl_pair r = spl_classify_pair(pos, ops, n_ops, t, part, n_part, comp, n_comp, strand_ok);
    if (r.cls == SPL_CLS_BETA1) bump(lds, p.beta1, s, wbase);
    else if (r.cls == SPL_CLS_ME) bump(lds + SPL_WIN, p.beta2s_reads, s, wbase);
    else if (r.cls != SPL_CLS_NONE) apply_pair_global(p, r, s, pos, ops, n_ops, part, n_part, part_off);
}

} // namespace

// =========================================================================================================
// Pair kernel: every read against every site of its fetch window, literally.
// =========================================================================================================
template <bool STRANDED>
__global__ __launch_bounds__(SPL_BLOCK) void spl_count_pairs_kernel(const spl_count_params p)
{
    __shared__ uint32_t lds[2 * SPL_WIN]; // [0,WIN): beta1   [WIN,2WIN): beta2Simple (read-derived)
    __shared__ uint32_t s_ops[SPL_BLOCK][5]; // a lane's short CIGAR, rebuilt from its packed record
    __shared__ int32_t s_wbase;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t chunk = my_chunk();
    const bool live = chunk < p.n_chunks;
    const ChunkView cv = chunk_view(p.chunk_meta + (live ? chunk : 0u));
    const uint32_t n_here = live ? cv.start[SPL_RC_RUNS] : 0u;

    for (int j = tid; j < 2 * SPL_WIN; j += SPL_BLOCK) lds[j] = 0u;
    if (tid == 0) s_wbase = live ? first_site_at_or_after(p, cv.first_pos + cv.shift) : 0;
    __syncthreads();
    const int32_t wbase = s_wbase;
    const int32_t n_sites = p.n_sites;

    for (uint32_t base = 0; base < n_here; base += SPL_BLOCK) {
        const uint32_t slot = base + (uint32_t)tid;
        bool valid = slot < n_here;
        int32_t pos = 0, end = -1, s = n_sites;
        uint32_t flag = 0, n_ops = 0;
        const uint32_t *ops = s_ops[tid];
        bool has_n = false;
        uint8_t rstrand = 0;
        if (valid) {
            const ReadView rv = read_at(cv, slot, s_ops[tid]);
            pos = rv.pos; flag = rv.flag; n_ops = rv.n_ops; ops = rv.ops;
            int64_t ref_len;
            spl_read_extent(ops, n_ops, &ref_len, &has_n);
            const int64_t end64 = (int64_t)pos + spl_fetch_len(flag, ref_len) - 1;
            // the CIGAR walk runs in int32: the whole read must fit, not only its fetch window
            if ((int64_t)pos + ref_len > (int64_t)SPL_COORD_MAX || rv.neg) {
                atomicOr(p.err, SPL_DEV_ERR_RANGE);
                valid = false;
            } else {
                end = (int32_t)end64;
                s = first_site_at_or_after(p, pos);
                if (STRANDED) rstrand = spl_read_strand(flag, p.stranded);
            }
        }
        // lane-serial part: the first few sites of this lane's read
        int served = 0;
        while (valid && s < n_sites && served < SPL_SERIAL_MAX) {
            const int32_t t = p.site_pos[s];
            if (t > end) break;
            do_pair<STRANDED>(p, lds, wbase, s, t, pos, ops, n_ops, has_n, rstrand);
            ++s;
            ++served;
        }
        // reads that span many sites (long introns) are finished by the whole wave, one site per lane
        const bool heavy = valid && s < n_sites && p.site_pos[s] <= end;
        unsigned long long todo = __ballot(heavy);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const int32_t b_pos = __shfl(pos, src);
            const int32_t b_end = __shfl(end, src);
            const int32_t b_s = __shfl(s, src);
            const uint32_t b_nops = __shfl(n_ops, src);
            const bool b_has_n = __shfl((int)has_n, src) != 0;
            const uint8_t b_rs = (uint8_t)__shfl((int)rstrand, src);
            // (a generic pointer: the source lane's row in LDS or its stretch of the wide-op array)
            const uintptr_t op_addr = (uintptr_t)ops;
            const uint32_t *b_ops = (const uint32_t *)(((uintptr_t)(uint32_t)__shfl((int)(uint32_t)(op_addr >> 32), src) << 32) |
                                                       (uintptr_t)(uint32_t)__shfl((int)(uint32_t)op_addr, src));
            for (int32_t s0 = b_s;; s0 += 64) {
                const int32_t my = s0 + lane;
                int32_t t = 0;
                const bool in = my < n_sites && (t = p.site_pos[my]) <= b_end;
                if (in) do_pair<STRANDED>(p, lds, wbase, my, t, b_pos, b_ops, b_nops, b_has_n, b_rs);
                if (__ballot(in) != ~0ull) break; // rows are sorted: a lane out of range ends the read
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < SPL_WIN; j += SPL_BLOCK) {
        const uint32_t a = lds[j], b = lds[SPL_WIN + j];
        if (a) atomicAdd(&p.beta1[wbase + j], a);
        if (b) atomicAdd(&p.beta2s_reads[wbase + j], b);
    }
}

// =========================================================================================================
// Range kernel.
//
// With compSplicing false -- which is the case for every site except the rivals enumerated below -- the
// if/elif chain of checkBam (:519-559) reduces to two coordinate tests per (read, site t):
//     beta1        <=>  one aligned op [c, c2-1] has c <= t and t+1 <= c2-1      (:469-477)
//     beta2Simple  <=>  one N op has lSite < t < rSite  (mutually exclusive)      (:507-512)
// (an alpha read changes no counter; flanking reads need compSplicing).  Both are "all rows with position in
// [c, c2-2]", a contiguous row range [lo, hi) of the sorted table, so each op costs two +-1 updates of a
// difference array instead of one classification per site.  Strand: a read only counts for rows of its own
// strand, so a stranded run keeps one pair of arrays per read strand and the scan picks by row strand.
//
// compSplicing can only become true for a site t that has a junction end of the read in its partner list
// (:494-501 need lSite or rSite in `partners`); partner links are mutual (:352-355), so those sites are the
// partners of the table rows AT the read's junction ends, and only those of them that have competitors at all
// (rows leading to such a partner carry SPL_SF_RIVALS; a constitutive junction has none).  Each rival inside
// the read's fetch window is classified literally, with and without its lists; if the two outcomes differ the
// range contribution is taken back and the literal one applied.  A rival reachable from several junction ends
// of one read is handled at the first of them.
// =========================================================================================================
namespace {

__device__ __forceinline__ void agg_add(uint32_t *addr, int32_t delta);

// Look (l, r) up in the junction table (spl_capi.cpp, build_junction_table).  The table is complete: a junction that is not in it
// cannot make compSplicing true for any site.
__device__ __forceinline__ bool jt_find(const uint4 *jhash, uint32_t jhash_mask, int32_t l, int32_t r, uint4 &ent)
{
    uint32_t h = (uint32_t)l * 0x9E3779B1u ^ (uint32_t)r * 0x85EBCA77u;
    h ^= h >> 15;
    for (int probe = 0; probe < 8; ++probe) {
        ent = jhash[2u * ((h + (uint32_t)probe) & jhash_mask)];
        if ((int32_t)ent.x == l && (int32_t)ent.y == r) return true;
        if (ent.x == 0x80000000u) return false;
    }
    return false;
}

// The literal walk for one read, any CIGAR, any table, both modes: the sites whose compSplicing test (:494-501) can succeed for
// the read are the rivals the junction table lists under the read's junctions; each of them inside the read's fetch window is
// classified with the literal state machine (spl_classify.h), with and without its lists -- where the two outcomes differ,
// what the ranges counted for that row is taken back and the literal outcome applied.  A rival listed under several junctions
// of the read is handled under the first of them.
template <bool STRANDED>
__device__ __forceinline__ void rivals_literal(const spl_count_params &p, int32_t pos, uint32_t flag, const uint32_t *ops,
                                               uint32_t n_ops, int32_t end_fetch)
{
    int32_t cur = pos;
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if (!((SPL_PROG_MASK >> code) & 1u)) continue;
        const int32_t d = (int32_t)(op >> 4);
        cur += d;
        if (code != SPL_OP_N) continue;
        const int32_t l = cur - d - 1, r = cur - 1;
        uint4 ent;
        if (!jt_find(p.jhash, p.jhash_mask, l, r, ent)) continue;
        {   // the same junction earlier in the read (0N ops, repeats): its rivals are done
            bool repeat = false;
            int32_t c2 = pos;
            for (uint32_t k2 = 0; k2 < k; ++k2) {
                const uint32_t op2 = ops[k2];
                if (!((SPL_PROG_MASK >> (op2 & 15u)) & 1u)) continue;
                const int32_t d2 = (int32_t)(op2 >> 4);
                c2 += d2;
                repeat |= (op2 & 15u) == SPL_OP_N && c2 - d2 - 1 == l && c2 - 1 == r;
            }
            if (repeat) continue;
        }
        const uint32_t n_riv = ent.w & SPL_JF_COUNT_MASK;
        for (uint32_t i = 0; i < n_riv; ++i) {
            const uint4 rx = p.jrivals[2u * (ent.z + i) + 1u]; // {row of t, its partner list offset, length, -}
            const int32_t trow = (int32_t)rx.x;
            const int32_t t = p.site_pos[trow];
            if (t < pos || t > end_fetch) continue;
            // listed under an earlier junction of this read too?  (compSplicing became true there: handled there)
            bool earlier = false;
            int32_t c2 = pos;
            for (uint32_t k2 = 0; k2 < k && !earlier; ++k2) {
                const uint32_t op2 = ops[k2];
                if (!((SPL_PROG_MASK >> (op2 & 15u)) & 1u)) continue;
                const int32_t d2 = (int32_t)(op2 >> 4);
                c2 += d2;
                if ((op2 & 15u) != SPL_OP_N) continue;
                uint4 e2;
                if (!jt_find(p.jhash, p.jhash_mask, c2 - d2 - 1, c2 - 1, e2)) continue;
                for (uint32_t i2 = 0; i2 < (e2.w & SPL_JF_COUNT_MASK); ++i2) earlier |= p.jrivals[2u * (e2.z + i2) + 1u].x == rx.x;
            }
            if (earlier) continue;
            const uint4 mt = p.site_meta[trow];
            const int32_t *part = p.part_pos + mt.x;
            const int32_t *comp = p.comp_pos + mt.z;
            bool strand_ok = true;
            if (STRANDED) strand_ok = (p.site_strand[trow] == spl_read_strand(flag, p.stranded));
            const spl_pair full = spl_classify_pair(pos, ops, n_ops, t, part, mt.y, comp, mt.w, strand_ok);
            const spl_pair base = spl_classify_pair(pos, ops, n_ops, t, nullptr, 0u, nullptr, 0u, strand_ok);
            if (full.cls == base.cls) continue;
            // take back what the ranges counted for this row, apply the literal outcome
            if (base.cls == SPL_CLS_BETA1) agg_add(&p.beta1[trow], -1);
            else if (base.cls == SPL_CLS_ME) agg_add(&p.beta2s_reads[trow], -1);
            apply_pair_global(p, full, trow, pos, ops, n_ops, part, mt.y, mt.x);
        }
    }
}

// A read flagged unmapped (0x4) is fetched as a 1-base record (htslib bam_endpos) whatever its CIGAR says,
// then walked with its full CIGAR: window [pos, pos], literal pairs.
template <bool STRANDED>
__device__ __forceinline__ void unmapped_read(const spl_count_params &p, int32_t pos, uint32_t flag, const uint32_t *ops, uint32_t n_ops)
{
    const int32_t n_sites = p.n_sites;
    for (int32_t s = first_site_at_or_after(p, pos); s < n_sites && p.site_pos[s] == pos; ++s) {
        bool strand_ok = true;
        if (STRANDED) strand_ok = (p.site_strand[s] == spl_read_strand(flag, p.stranded));
        const uint4 m = p.site_meta[s];
        const spl_pair r = spl_classify_pair(pos, ops, n_ops, pos, p.part_pos + m.x, m.y, p.comp_pos + m.z, m.w, strand_ok);
        apply_pair_global(p, r, s, pos, ops, n_ops, p.part_pos + m.x, m.y, m.x);
    }
}

// Position -> distinct-position index ("dpos": rows sharing a position, e.g. the '+' and '-' site of a stranded
// table, share one index).  32 bp buckets, one 8-byte entry each: {dpos of the first site at or after the bucket
// start, occupancy mask of the bucket's 32 positions}.  One load answers both "how many site positions are < x" and
// "is x a site" with a popcount -- no dependent second access, so all boundaries of a read resolve in one memory
// trip, in four instructions each (bit-field mask, and, counting add, bit-field extract).  Which of the bucket's
// positions are sites WITH RIVALS is a second array of 32-bit masks, asked only for junction ends.
template <class P>
__device__ __forceinline__ uint32_t dbk_slot(const P &p, int32_t x)
{
    // the table starts with empty buckets before the first site (dbase >= -64) and ends with an empty bucket whose first
    // dpos is n_dpos, so clamping the index is all the range handling there is; x - dbase cannot wrap because
    // coordinates stay <= SPL_COORD_MAX = 2^31 - 67
    int32_t b = (x - p.dbase) >> 5;
    b = b < 0 ? 0 : b;
    const int32_t last = (int32_t)p.n_dbuckets - 1;
    return (uint32_t)(b > last ? last : b);
}

// For x outside the table the clamped entry is an empty bucket: count 0, not a site, not flagged, whatever the bit index says.
template <class P>
__device__ __forceinline__ void dbk_resolve(const P &p, int32_t x, const spl_dbk &e, int32_t &u, uint32_t &nv)
{
    const uint32_t bit = (uint32_t)(x - p.dbase) & 31u;
    u = (int32_t)(e.first + (uint32_t)__popc(e.occ & ((1u << bit) - 1u)));
    nv = (e.occ >> bit) & 1u;
}

// The same with the bucket's mask of flagged positions: rv = x is an end of a junction that has rivals.
template <class P>
__device__ __forceinline__ void dbk_resolve(const P &p, int32_t x, const spl_dbk &e, int32_t &u, uint32_t &nv, uint32_t &rv)
{
    dbk_resolve(p, x, e, u, nv);
    rv = (e.rival >> ((uint32_t)(x - p.dbase) & 31u)) & 1u;
}

typedef __attribute__((address_space(3))) int32_t spl_lds_i32; // the difference windows, typed as what they are: LDS

// All 64 lanes call this together.  Lanes that add `sign` to the same key = (dpos << 2 | array) and sit next to each
// other form a run; the first lane of each run adds sign * run-length once.  (Equal keys that are NOT adjacent make
// several runs: still correct, just more atomics -- that only happens for unsorted input.)
template <int NARR, bool AGG, int WIN = (NARR == 4 ? SPL_WIN_STRANDED : SPL_WIN)>
__device__ __forceinline__ void commit_key(const spl_hot_params &p, spl_lds_i32 *lds, int32_t wbase, bool valid, uint32_t key, int32_t sign)
{
    int32_t amount = sign;
    bool go = valid;
    if (AGG) {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const uint32_t k = valid ? key : 0xffffffffu;
        // previous lane's key by a DPP wave shift (no LDS round trip); lane 0 keeps its own and is a head anyway
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const bool head = valid && (lane == 0 || prev != k);
        const unsigned long long heads = __ballot(head);
        const unsigned long long act = __ballot(valid);
        const unsigned long long stop = (heads | ~act) & ~((2ull << lane) - 1ull); // later lanes that end my run
        const int len = stop ? (__ffsll((long long)stop) - 1 - lane) : (64 - lane);
        amount = sign * len;
        go = head;
    }
    if (go) {
        const int arr = (int)(key & 3u);
        const int32_t d = (int32_t)(key >> 2);
        const uint32_t loc = (uint32_t)(d - wbase);
        // (an LDS-typed pointer: ds_add on one side, a global atomic on the other, never a flat atomic on a selected address)
        if (loc <= (uint32_t)WIN) __hip_atomic_fetch_add(lds + (arr * (WIN + 1) + (int)loc), amount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(&p.diff[(int64_t)arr * p.diff_stride + d], amount);
    }
}

// A range of distinct positions [lo, ub) of difference array `arr`: +1 at lo, -1 at ub -- the two commit_keys of every range, with
// what they share done once: both ends inside the workgroup's window (nearly always: the window is the chunk's) are two LDS adds
// behind ONE test; anything else takes the two separate ways.  All 64 lanes call this together.
template <int NARR, bool AGG, int WIN>
__device__ __forceinline__ void commit_range(const spl_hot_params &p, spl_lds_i32 *lds, int32_t wbase, bool em, int32_t lo, int32_t ub, uint32_t arr)
{
    if (AGG) {
        commit_key<NARR, AGG, WIN>(p, lds, wbase, em, ((uint32_t)lo << 2) | arr, 1);
        commit_key<NARR, AGG, WIN>(p, lds, wbase, em, ((uint32_t)ub << 2) | arr, -1);
        return;
    }
    const uint32_t a = (uint32_t)(lo - wbase), b = (uint32_t)(ub - wbase);
    const bool inside = a <= (uint32_t)WIN && b <= (uint32_t)WIN;
    if (em && inside) {
        spl_lds_i32 *const row = lds + (int)arr * (WIN + 1);
        __hip_atomic_fetch_add(row + (int)a, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(row + (int)b, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (__any(em && !inside)) {
        commit_key<NARR, false, WIN>(p, lds, wbase, em && !inside, ((uint32_t)lo << 2) | arr, 1);
        commit_key<NARR, false, WIN>(p, lds, wbase, em && !inside, ((uint32_t)ub << 2) | arr, -1);
    }
}

// atomicAdd(addr, delta) with delta = +1 or -1, merged over the lanes of the wave that are executing it right now
// and target the same word: one atomic per distinct address instead of one per lane.
__device__ __forceinline__ void agg_add(uint32_t *addr, int32_t delta)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long todo = __ballot(1);
    const uint32_t lo = (uint32_t)(uintptr_t)addr, hi = (uint32_t)((uintptr_t)addr >> 32);
    // A hot counter is shared by many lanes and therefore shows up among the first few distinct addresses; when the lanes
    // all differ there is nothing to merge, and electing 64 leaders one after the other costs far more than 64 plain
    // atomics.  So: a few rounds of merging, then everybody left adds for himself.
    for (int round = 0; round < SPL_AGG_ROUNDS; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const bool same = (__shfl((int)lo, leader) == (int)lo) && (__shfl((int)hi, leader) == (int)hi);
        const unsigned long long grp = __ballot(same);
        const unsigned long long plus = __ballot(same && delta > 0);
        if (lane == leader) {
            const int32_t sum = 2 * (int32_t)__popcll(plus) - (int32_t)__popcll(grp);
            if (sum) atomicAdd(addr, (uint32_t)sum);
        }
        if (same) return;
        todo &= ~grp;
    }
    atomicAdd(addr, (uint32_t)delta);
}


// Rivals of a read whose ONLY junction (l, r) is a junction of the BED file, resolved inside the range kernel from a
// table built at upload (spl_capi.cpp, build_junction_table): the sites t for which compSplicing is true given (l, r)
// -- evaluated there with the literal membership tests of checkBam (:494-501) -- with their dpos, strand and the
// partner edges that take a double count.  With one junction the if/elif chain (:516-559) leaves two corrections:
//   t strictly inside the intron  -> flanking read: the +1 the ME range gave t's beta2Simple is taken back;
//   an aligned block covers t,t+1 -> beta1-type read: -1 beta1, +1 beta2Simple, +1 double count per listed edge.
// Point updates go to the same LDS difference arrays as the ranges.  Returns false when the read must take the
// literal kernel instead (junction not in the table = not a BED junction, table entry marked complex, ...).
__device__ __forceinline__ uint32_t junction_hash(int32_t l, int32_t r)
{
    uint32_t h = (uint32_t)l * 0x9E3779B1u ^ (uint32_t)r * 0x85EBCA77u;
    return h ^ (h >> 15);
}

// (ent, first) = the two quads of slot h & mask, already loaded by the caller -- who may have asked for several reads' slots in
// one trip -- further probes and further rival records are fetched here.
template <bool STRANDED, int NARR, int WIN = (NARR == 4 ? SPL_WIN_STRANDED : SPL_WIN)>
__device__ __forceinline__ bool rivals_inline_from(const spl_hot_params &p, spl_lds_i32 *lds, int32_t wbase, int32_t l, int32_t r, uint32_t h,
                                                   uint4 ent, uint4 first, const int32_t *blk_a, const int32_t *blk_b, uint32_t sidx)
{
    bool found = false;
    for (int probe = 0; probe < 8; ++probe) {
        if (probe) {
            const uint4 *slot = p.jhash + 2u * ((h + (uint32_t)probe) & p.jhash_mask);
            ent = slot[0];
            first = slot[1]; // the first rival's record rides along: one trip for the usual one-rival junction
        }
        if ((int32_t)ent.x == l && (int32_t)ent.y == r) { found = true; break; }
        if (ent.x == 0x80000000u) break; // empty slot: not a BED junction with flagged ends
    }
    if (!found) return true; // the table is complete: no site anywhere has this junction as a rival's
    const uint32_t n_riv = ent.w & SPL_JF_COUNT_MASK;
    if ((ent.w & SPL_JF_COMPLEX) || n_riv > 4u) return false;
    if (!STRANDED && (ent.w & SPL_JF_MULTIROW)) return false; // several rows share a rival's position: per-row only
    for (uint32_t i = 0; i < n_riv; ++i) {
        const uint4 rv = i ? p.jrivals[2u * (ent.z + i)] : first; // {t_pos, t_dpos | strand << 30, edge0, edge1} (+ a second quad: literal kernel)
        const int32_t t = (int32_t)rv.x;
        const uint32_t td = rv.y & 0x3fffffffu;
        if (STRANDED && ((rv.y >> 30) != (sidx ? 2u : 1u))) continue; // strand_ok false: the ranges added nothing
        const uint32_t a_b1 = sidx, a_me = (STRANDED ? 2u : 1u) + sidx;
        if (t > l && t < r) { // flanking: not counted by `process` (:529-536)
            commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_me, -1);
            commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_me, 1);
        } else {
            bool cov = false;
#pragma unroll
            for (int j = 0; j < 2; ++j) cov |= (blk_a[j] <= t) && (t + 1 <= blk_b[j]);
            if (cov) { // beta1-type (:544-556)
                commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_b1, -1);
                commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_b1, 1);
                commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_me, 1);
                commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_me, -1);
                if (rv.z != 0xffffffffu) agg_add(&p.dbl[rv.z & 0x7fffffffu], 1);
                if (rv.w != 0xffffffffu) agg_add(&p.dbl[rv.w], 1);
            }
        }
    }
    return true;
}

// (Measured and not kept, round 4: the wave taking its marked reads a GROUP at a time -- the lanes that hold the same junction
//  and read strand as the first lane still open, the group's table entry and rivals in scalar registers, a ballot for who covers a
//  rival with an aligned block, ONE lane adding the group's count -- instead of sixty-four atomics on one LDS word.  Right, and
//  2 % slower on the human-scale sample, 6 % on A. thaliana: the serialised atomics were never what the pass waited for.
//  profiles/r04r_range_groups_ab.txt)
// The same for a twice-spliced read (junctions jl/jr[0..1], aligned blocks blk[0..2]): the two-junction case of
// rivals_table_path (see there for the rules: a rival is handled under the first junction that lists it, flanking needs
// "inside that or a later intron", alpha reads and beta1-type reads take double counts on the rival's edges to any junction end
// of the read except the partner used).  flagged[j]: an end of junction j carries a rival bit -- such a junction must be in
// the table, or the read is the literal kernel's.  Point updates go to the LDS difference windows.
template <bool STRANDED, int NARR, int WIN = (NARR == 4 ? SPL_WIN_STRANDED : SPL_WIN)>
__device__ __forceinline__ bool rivals_inline2(const spl_hot_params &p, spl_lds_i32 *lds, int32_t wbase, const int32_t (&jl)[2],
                                               const int32_t (&jr)[2], const bool (&flagged)[2], const int32_t (&blk_a)[3],
                                               const int32_t (&blk_b)[3], uint32_t sidx)
{
    // the table slots of both junctions (each with its first rival record riding along) in one trip; at load 1/4 the
    // first probe almost always decides
    uint32_t r_off[2] = {0, 0}, r_n[2] = {0, 0};
    uint4 ent[2], first[2];
    uint32_t h[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        h[j] = (uint32_t)jl[j] * 0x9E3779B1u ^ (uint32_t)jr[j] * 0x85EBCA77u;
        h[j] ^= h[j] >> 15;
        const uint4 *slot = p.jhash + 2u * (h[j] & p.jhash_mask);
        ent[j] = slot[0];
        first[j] = slot[1];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int32_t l = jl[j], r = jr[j];
        bool found = false;
        for (int probe = 0; probe < 8; ++probe) {
            if (probe) {
                const uint4 *slot = p.jhash + 2u * ((h[j] + (uint32_t)probe) & p.jhash_mask);
                ent[j] = slot[0];
                first[j] = slot[1];
            }
            if ((int32_t)ent[j].x == l && (int32_t)ent[j].y == r) { found = true; break; }
            if (ent[j].x == 0x80000000u) break;
        }
        if (found) {
            if ((ent[j].w & SPL_JF_COMPLEX) || (ent[j].w & SPL_JF_COUNT_MASK) > 16u) return false;
            if (!STRANDED && (ent[j].w & SPL_JF_MULTIROW)) return false;
            r_off[j] = ent[j].z;
            r_n[j] = ent[j].w & SPL_JF_COUNT_MASK;
        } // (not in the table: the table is complete, the junction has no rivals)
    }
    const uint32_t want = sidx ? 2u : 1u;
    const uint32_t a_b1 = sidx, a_me = (STRANDED ? 2u : 1u) + sidx;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        for (uint32_t i = 0; i < r_n[j]; ++i) {
            const uint4 rv = i ? p.jrivals[2u * (r_off[j] + i)] : first[j];
            bool earlier = false; // listed under the first junction too: handled there
            if (j == 1)
                for (uint32_t i2 = 0; i2 < r_n[0]; ++i2) earlier |= ((i2 ? p.jrivals[2u * (r_off[0] + i2)].y : first[0].y) == rv.y);
            if (earlier) continue;
            const int32_t t = (int32_t)rv.x;
            const uint32_t td = rv.y & 0x3fffffffu;
            const bool strand_ok = !STRANDED || (rv.y >> 30) == want;
            int inside = -1;
            bool cov = false, alpha = false;
            int32_t pu = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (t > jl[a] && t < jr[a]) inside = a;
                if (jl[a] == t) { pu = jr[a]; alpha = true; } // :487-492, in op order
                if (jr[a] == t) { pu = jl[a]; alpha = true; }
            }
#pragma unroll
            for (int b = 0; b < 3; ++b) cov |= (blk_a[b] <= t) && (t + 1 <= blk_b[b]);
            const bool beta1type = !alpha && inside < 0 && cov && strand_ok;
            if (alpha || beta1type) {
                const uint4 rx = p.jrivals[2u * (r_off[j] + i) + 1u]; // {row of t, its partner list offset, length, -}
                for (uint32_t e2 = 0; e2 < rx.z; ++e2) {
                    const int32_t pp = p.part_pos[rx.y + e2];
                    const bool is_end = pp == jl[0] || pp == jr[0] || pp == jl[1] || pp == jr[1];
                    if (is_end && !(alpha && pp == pu)) agg_add(&p.dbl[rx.y + e2], 1);
                }
                if (beta1type) {
                    commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_b1, -1);
                    commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_b1, 1);
                    commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_me, 1);
                    commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_me, -1);
                }
            } else if (inside >= j && strand_ok) { // flanking: the ME range counted it, `process` does not
                commit_key<NARR, false, WIN>(p, lds, wbase, true, (td << 2) | a_me, -1);
                commit_key<NARR, false, WIN>(p, lds, wbase, true, ((td + 1u) << 2) | a_me, 1);
            }
        }
    }
    return true;
}

} // namespace

// The range kernel proper: one workgroup per chunk, one read per lane per wave-iteration, straight-line, two batched memory
// trips per read --
//   trip 1  the read's record (8, 16 or 24 bytes by run; asked for one iteration ahead)
//   trip 2  one bucket entry per boundary (read start + end of every reference-consuming op), all independent
// then popcounts turn boundaries into dpos ranges.  A chunk's reads come as four runs (spl_pack.h): a wave-iteration takes 64
// reads of ONE run, so every wave runs one code path at a time on a branch that is decided by scalar arithmetic on the
// chunk's descriptor, not by loaded data; the wave-iterations of a chunk (32 + at most 3 partial ones) are dealt round-robin
// to the four waves.  No LDS staging, no barriers inside the loop, few registers and a small argument block (everything the
// literal paths need lives in spl_count_literal_kernel): reads that need a literal decision -- unmapped-but-placed records and
// reads with a junction end that has rival sites the junction table cannot settle -- are appended to a queue.
// (The instrumented builds this kernel was tuned with -- phase stamps, knock-out variants -- are a patch, not part of this
//  source: profiles/experiments/r05_range_kernel_instrumentation.patch.)

// FUSED (round 5): the same pass over a read set that is still BAM-native arrays in device memory (spl_devpack.h: the chunks are
// cells of the grid over the arrays' indexes, p.cells) -- the workgroup takes its chunk a TILE of 1024 reads at a time, makes the
// tile's records itself with the layout kernel's code (spl_layout_tile.h: a thread's four reads, the tile's ops through LDS,
// classification, ranks by prefix sums) but into LDS, where the ops were, and counts them from there with the code below: the
// records -- 12.7 bytes a read written and read again, beside 18.7 of arrays -- never go to memory, and there is no layout
// launch.  The difference windows, the lists' front parts and the window's base belong to the chunk, not the tile.  39 KB of
// LDS, 96 VGPRs: four workgroups a CU -- a stranded pass with windows of 508 distinct positions instead of 956 (with 956, three
// workgroups a CU: slower than layout + range; what lies outside a window goes to the global arrays either way).  A queue entry
// names its read by its place in the arrays (s_idx), which is where the literal kernel then reads it.
// Measured (human-scale, 100 M reads a launch): 0.94-0.98 ms against 0.655-0.67 + 0.375-0.39 for layout + range.  Asking for the next
// tile's reads before this tile is counted gained nothing (a wave's memory operations return in order: the first bucket entry
// waits for them), 512 threads -- two reads each, eight waves counting a tile -- took 1.25 ms: profiles/r05X_fused_pass.txt.
// Default (AGG false): plain LDS atomics, 64 VGPRs = 8 waves per SIMD (the kernel lives on how many waves are there to
// cover each other's memory trips and barriers; the register cap costs nothing -- no scratch).  Merging the atomics of
// neighbouring lanes first (AGG, SPL_OPT_WAVE_AGGREGATION) needs a few more registers than that cap allows and was
// never faster in measurements, not even at 8000 reads per site; it stays as a variant for parity tests.
template <bool STRANDED, bool AGG, bool BIG, bool FUSED>
__global__ __launch_bounds__(FUSED ? SPL_BLOCK_FUSED : SPL_BLOCK) __attribute__((amdgpu_waves_per_eu(FUSED || AGG ? 4 : 8, 8))) void spl_count_ranges_kernel(const spl_hot_params p)
{
    constexpr int NARR = STRANDED ? 4 : 2; // {beta1, ME} x {read strand +, -}
    constexpr int WIN = STRANDED ? (FUSED ? SPL_WIN_STRANDED_FUSED : SPL_WIN_STRANDED) : SPL_WIN;
    __shared__ int32_t lds_words[NARR * (WIN + 1)];
    spl_lds_i32 *const lds = (spl_lds_i32 *)lds_words;
    // Each wave owns one segment of s_q (as many entries as it can have reads) with two lists of chunk-relative slots:
    // from the front the reads for the literal kernel, from the back the once- and twice-spliced reads whose junction has rivals
    // (finished from the junction table by the wave itself right after its loop, see below).  Only the owning wave touches a
    // segment, so the fill counts are wave-uniform registers and slots are handed out by ballot, not by atomics.
    constexpr int BLOCK = FUSED ? SPL_BLOCK_FUSED : SPL_BLOCK, NWAVE = BLOCK / 64;
    constexpr int RPT = (int)SPL_TILE_FUSED / SPL_BLOCK_FUSED; // FUSED: reads a thread takes of a tile
    constexpr uint32_t SEG = FUSED ? SPL_WAVE_READS_FUSED : SPL_WAVE_READS;
    constexpr uint32_t CSHIFT = BIG ? SPL_CHUNK_BIG_SHIFT : SPL_CHUNK_SHIFT;
    // FUSED: the chunk's records are made here, in LDS, from the BAM-native arrays (spl_layout_tile.h: the layout kernel's body),
    // a TILE of SPL_TILE_FUSED reads at a time, and counted from there -- they never exist in memory (a tile's reads are asked for
    // when the tile before is through: asking earlier was measured and bought nothing, profiles/r05X_fused_pass.txt).  s_rec: first the stage of the tile's ops, then its records.
    constexpr uint32_t TILE = SPL_TILE_FUSED, TILES = FUSED ? (1u << CSHIFT) / TILE : 1u;
    constexpr uint32_t REC_BYTES = FUSED ? (uint32_t)SPL_LAYOUT_SLOT(TILE) : 16u;
    __shared__ uint4 s_rec[REC_BYTES / 16u];
    __shared__ uint16_t s_idx[FUSED ? TILE : 1];   // read q of the tile's runs 1 .. 3 -> its index in the chunk's cell of the arrays
    __shared__ uint32_t s_lay[2 * NWAVE];
    typedef __attribute__((address_space(3))) const char spl_lchar;
    spl_lchar *const lrec = (spl_lchar *)s_rec;
    __shared__ uint16_t s_q[NWAVE * SEG];
    __shared__ uint32_t s_qcnt[NWAVE], s_qbase;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform, in an SGPR
    const uint32_t seg0 = wave * SEG;
    const uint32_t lane = (uint32_t)threadIdx.x & 63u;
    uint32_t n_front = 0, n_back = 0, back_done = 0; // (back_done: entries of the back list the list pass is through with)
    // The once-spliced reads with rivals are not listed: a lane remembers WHICH of its reads they were, two bits per iteration
    // of the run (a wave has eight iterations of it at most), and the run is streamed a second time for them after the loops
    // -- coalesced, asked for an iteration ahead, out of the L2 it has just come through -- instead of being gathered read by
    // read from slots kept in LDS: one memory trip per batch of 128 (the table slots) where the list pass had three in a row.
    uint32_t fm_mnm = 0, it_mnm = 0;
    auto rank_in = [](unsigned long long m) { // how many lanes below mine are in m (mbcnt: no per-lane mask to keep around)
        return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    // workgroup -> slot of its XCD's share (my_chunk) -> chunk: every share is walked longest chunk first (chunk_order, made
    // from the layout's cost estimate by spl_chunk_order_kernel), so the workgroups that finish a launch are short ones.
    // (Measured and not kept, round 4: workgroups that STAY and take chunk after chunk of their XCD's share from a counter, as
    // many as the chip holds -- to fill the eighth of the workgroup slots that stand empty between a workgroup's end and its
    // successor's first records: 0.443 ms a launch against 0.39, with ten registers in scratch for the loop around the body.)
    const uint32_t chunk_slot = my_chunk();
    const uint32_t ordered = chunk_slot < p.n_chunks ? p.chunk_order[chunk_slot] : 0xffffffffu; // (p.n_chunks: the SLOTS; a share's last ones may be empty)
    const bool live = ordered != 0xffffffffu;
    const uint32_t chunk = live ? ordered : 0u;
    // A segment holds what a wave lists in all but pathological chunks (SEG entries for the ~1000 reads a wave walks; simple reads
    // are never listed).  A list that is full does not drop a read: its entries go straight into the literal queue -- one
    // returning atomic per push instead of one per workgroup, on a path that a chunk of mostly flagged spliced reads takes --,
    // and a twice-spliced read that finds the back list full is the literal kernel's (which takes any read).
    uint32_t n_simple = 0; // (FUSED: s_idx counts from the first read that is not a simple one)
    auto entry_of = [&](uint32_t slot) { return FUSED ? (uint32_t)((__attribute__((address_space(3))) const uint16_t *)s_idx)[(slot & 0x3fffu) - n_simple] : slot; };
    auto push_direct = [&](bool want, uint32_t slot, unsigned long long m, uint32_t n) {
        const uint32_t shard = blockIdx.x & 7u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&p.queue_n[shard * SPL_COUNTER_STRIDE], n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (want) p.queue[(size_t)shard * p.queue_cap + base + rank_in(m)] = (chunk << CSHIFT) | entry_of(slot);
    };
    auto push_front = [&](bool want, uint32_t slot) {
        const unsigned long long m = __ballot(want);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n_front + n_back - back_done + n > SEG) { push_direct(want, slot, m, n); return; }
        if (want) s_q[seg0 + n_front + rank_in(m)] = (uint16_t)entry_of(slot); // (FUSED: the tile's records are gone when the list is handed over)
        n_front += n;
    };
    auto push_back = [&](bool want, uint32_t slot) {
        const unsigned long long m = __ballot(want);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n_front + n_back + n > SEG) { push_front(want, slot & 0x3fffu); return; }
        if (want) s_q[seg0 + SEG - 1u - n_back - rank_in(m)] = (uint16_t)slot;
        n_back += n;
    };

    const int tid = threadIdx.x;
    int32_t wbase = 0;
    ChunkView cv;
    // (the first two words of a bucket entry: all a boundary needs that is no junction end)
    // (an entry's place as a 32-bit byte offset from the table's base -- 12 s by a shift and a shift-and-add; the table is below 4 GB --
    //  so that the load is "scalar base + 32-bit lane offset": the 64-bit base + 12 s the compiler makes of an indexed pointer is a
    //  v_mad_u64_u32 per boundary, a quarter-rate instruction, three boundaries a read)
    typedef __attribute__((address_space(1))) const uint32_t spl_gdw;
    auto dbk_at = [&](uint32_t s) { return (spl_gdw *)((__attribute__((address_space(1))) const char *)p.dbucket + (size_t)((s << 3) + (s << 2))); };
    auto dbk2 = [&](uint32_t s) { spl_gdw *q = dbk_at(s); spl_dbk e; e.first = q[0]; e.occ = q[1]; e.rival = 0u; return e; };
    auto dbk3 = [&](uint32_t s) { spl_gdw *q = dbk_at(s); spl_dbk e; e.first = q[0]; e.occ = q[1]; e.rival = q[2]; return e; };
    // ---- FUSED: the chunk's tiles.  Tile k is the part [lo, hi) of the chunk in cell [g0 + k TILE, + TILE) of the arrays' indexes.
    spl_layout_chunk ch;
    uint32_t tile = 0, tile_last = 0;
    uint32_t ob[TILES + 1]; // cig_off at the tiles' boundaries
    auto span_of = [&](uint32_t k) {
        const int64_t g0 = ch.lo & ~(int64_t)((1u << CSHIFT) - 1u), hi = ch.lo + ch.n;
        spllay::TileSpan sp;
        sp.cell0 = g0 + (int64_t)k * TILE;
        sp.lo = sp.cell0 > ch.lo ? sp.cell0 : ch.lo;
        sp.hi = sp.cell0 + TILE < hi ? sp.cell0 + TILE : hi;
        uint32_t a = ob[0], b = ob[1];
#pragma unroll
        for (uint32_t q = 1; q < TILES; ++q) { a = k == q ? ob[q] : a; b = k == q ? ob[q + 1] : b; }
        sp.o_lo = a; sp.o_hi = b; sp.o_fetch_hi = ch.o_hi; sp.seg_op0 = ch.seg_op0;
        return sp;
    };
    if constexpr (FUSED) {
        if (!live) return; // (the whole workgroup: before any barrier)
        ch = p.cells[chunk];
        const int64_t g0 = ch.lo & ~(int64_t)((1u << CSHIFT) - 1u), hi = ch.lo + ch.n;
        tile = (uint32_t)((ch.lo - g0) >> SPL_TILE_FUSED_SHIFT);
        tile_last = (uint32_t)((hi - 1 - g0) >> SPL_TILE_FUSED_SHIFT);
        // (the window's base: the bucket entry of the chunk's first POS is asked for now, beside the arrays)
        const int32_t fp0 = p.src.pos[ch.lo] + ch.shift;
        const spl_dbk e_first = p.dbucket[dbk_slot(p, fp0 - 1)];
#pragma unroll
        for (uint32_t q = 0; q <= TILES; ++q) {
            int64_t i = g0 + (int64_t)q * TILE;
            i = i < ch.lo ? ch.lo : (i > hi ? hi : i);
            ob[q] = q == 0u ? ch.o_lo : (q == TILES ? ch.o_hi : p.src.cig_off[i]);
        }
        for (int j = tid; j < NARR * (WIN + 1); j += BLOCK) lds[j] = 0; // (a barrier lies between this and the first count: tile_finish has three)
        cv.rec = nullptr;
        cv.wide = (spl_gu32 *)(p.src.cigar + ch.seg_op0);
        cv.shift = ch.shift;
        cv.first_pos = fp0 - ch.shift;
        { uint32_t nv; dbk_resolve(p, fp0 - 1, e_first, wbase, nv); }
    } else {
        cv = chunk_view(p.chunk_meta + chunk);
    }
    // a record's 16 / 8 bytes at a byte offset of the chunk's record area
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    auto ld_r4 = [&](uint32_t at) {
        if constexpr (FUSED) { const u32x4_a8 v = *(__attribute__((address_space(3))) const u32x4_a8 *)(lrec + at); return make_uint4(v.x, v.y, v.z, v.w); }
        else return ld_g4(cv.rec + (size_t)at);
    };
    auto ld_r2 = [&](uint32_t at) {
        if constexpr (FUSED) { const spl_u32x2 v = *(__attribute__((address_space(3))) const spl_u32x2 *)(lrec + at); return make_uint2(v.x, v.y); }
        else return ld_g2(cv.rec + (size_t)at);
    };
    for (;;) { // the chunk's tiles (FUSED), or the chunk in one piece
    if constexpr (FUSED) {
        const spllay::TileSpan sp = span_of(tile);
        uint32_t n[4];
        {
            spllay::TileLoads<(int)TILE, RPT> L;
            spllay::tile_issue<(int)TILE, RPT>(p.src, p.src_n_rec, p.src_n_ops, sp, L);
            spllay::tile_finish<(int)TILE, RPT>(p.src, p.src_n_ops, sp, L, (spllay::lay_lds_w32 *)s_rec, (spllay::lay_lds_w32 *)s_lay,
                                                spllay::RecordsInLds{(spllay::lay_lds_u8 *)s_rec, (spllay::lay_lds_u16 *)s_idx, tile * TILE}, n);
        }
        cv.start[0] = 0; cv.start[1] = n[0]; cv.start[2] = n[0] + n[1]; cv.start[3] = n[0] + n[1] + n[2]; cv.start[4] = n[0] + n[1] + n[2] + n[3];
        cv.off[0] = 0;
        cv.off[1] = (n[0] * SPL_REC_SIMPLE + 15u) & ~15u;
        cv.off[2] = cv.off[1] + n[1] * SPL_REC_MNM;
        cv.off[3] = cv.off[2] + n[2] * SPL_REC_M2;
        n_simple = n[0];
        __syncthreads(); // the tile's records are there
    }
    // Wave-iterations of the chunk.  A wave-iteration takes 64 * K consecutive reads of ONE run, K per lane: K = 4 for simple reads
    // (32 bytes of records per lane), 2 for once-spliced ones (32 bytes), 1 for the rest (24 bytes).  Run r has iters[r]
    // wave-iterations, the first being number g_start[r] of the chunk.
    // Why several reads per lane: vector-memory results come back in order, so waiting for this iteration's bucket entries also
    // waits for the next iteration's records -- a wave never has more than ONE load of the stream in flight across that wait, and
    // every iteration costs a round trip to HBM.  What can grow is the size of that one load: 2 KB per wave instead of 512 bytes.
    constexpr uint32_t KS = SPL_K_SIMPLE, KM = SPL_K_MNM;
    uint32_t g_start[SPL_RC_RUNS + 1];
    g_start[0] = 0;
    g_start[1] = (cv.start[1] + 64u * KS - 1u) / (64u * KS);
    g_start[2] = g_start[1] + (cv.start[2] - cv.start[1] + 64u * KM - 1u) / (64u * KM);
    g_start[3] = g_start[2] + ((cv.start[3] - cv.start[2] + 63u) >> 6);
    g_start[4] = g_start[3] + ((cv.start[4] - cv.start[3] + 63u) >> 6);
    const uint32_t g_total = live ? g_start[SPL_RC_RUNS] : 0u;
    // Pipeline over the wave's iterations: while one is worked on the records of the next are in flight
    // (its bucket entries too was tried: 25 more registers, occupancy 4, slower).
    // (plain loads: marking the stream non-temporal looked 3...11 % faster in a benchmark that runs pass after pass
    //  over one 400 MB sample -- the words then survive in the 256 MB last-level cache from one pass to
    //  the next -- and is 2...5 % SLOWER when every pass reads a different copy of the sample: DESIGN.md section 6)
    typedef uint4 W4;
    W4 cu0 = make_uint4(0, 0, 0, 0), cu1 = make_uint4(0, 0, 0, 0); // ---- trip 1: 32 or 24 bytes of records per lane
    uint32_t cu_i0 = 0;                               // the index, in its run, of the lane's first read
    // (select chains, no indexing by `run`: the descriptor stays in scalar registers)
    auto fetch = [&](uint32_t g) {
        const uint32_t run = (g >= g_start[1] ? 1u : 0u) + (g >= g_start[2] ? 1u : 0u) + (g >= g_start[3] ? 1u : 0u);
        uint32_t gs = 0, n_run = cv.start[1], off = 0, size = SPL_REC_SIMPLE, per = KS;
        if (run == 1u) { gs = g_start[1]; n_run = cv.start[2] - cv.start[1]; off = cv.off[1]; size = SPL_REC_MNM; per = KM; }
        if (run == 2u) { gs = g_start[2]; n_run = cv.start[3] - cv.start[2]; off = cv.off[2]; size = SPL_REC_M2; per = 1u; }
        if (run == 3u) { gs = g_start[3]; n_run = cv.start[4] - cv.start[3]; off = cv.off[3]; size = SPL_REC_OTHER; per = 1u; }
        const uint32_t i0 = (((g - gs) << 6) + lane) * per;

        cu_i0 = i0;
        // (a lane wholly past the end of the run reads the run's first records and is masked; a lane whose LAST reads are past
        //  the end reads at most 24 bytes beyond the run: other runs, or the padding every segment's allocation ends with)
        // Always the same two 16-byte loads, no branch anywhere in here: the compiler counts outstanding loads statically, and
        // only if every way through the loop body issues the same number after the bucket loads can the wait for the bucket
        // entries leave these two in flight (a branch here turns that wait into "wait for everything").  A 24-byte record's
        // second load takes 8 bytes of its neighbour along.
        const uint32_t r = off + (i0 < n_run ? i0 : 0u) * size;
        cu0 = ld_r4(r);
        cu1 = ld_r4(r + 16u);
    };
    if constexpr (!FUSED) { if (wave < g_total) fetch(wave); }
    if constexpr (!FUSED) {
        { const int32_t fp = cv.first_pos + cv.shift; uint32_t nv; const spl_dbk e = p.dbucket[dbk_slot(p, fp - 1)]; dbk_resolve(p, fp - 1, e, wbase, nv); }
        for (int j = tid; j < NARR * (WIN + 1); j += BLOCK) lds[j] = 0;
        __syncthreads();
    }

    {
        // One loop per run, the wave's iterations g = wave, wave + 4, ... running through all of them; the records of the NEXT
        // iteration -- of this run or the next -- are asked for right AFTER this iteration's bucket entries (memory operations
        // retire in order, and the entries must not wait for the stream), in straight-line code: one place per loop body, so
        // that no pass of the compiler finds common code to move to the end of the iteration.
        // Straight-line up to the commits: lanes past the end of a run are masked, all loads of a trip issue back to
        // back.  Control flow is wave-uniform around every commit_key (all 64 lanes reach it).
        uint32_t g = live ? wave : g_start[SPL_RC_RUNS]; // (a workgroup without a chunk has chunk 0's descriptor and no iteration at all)
        auto fetch_next = [&]() { if constexpr (!FUSED) fetch(g + NWAVE < g_total ? g + NWAVE : g); }; // (the last iteration asks for itself again)
        // FUSED: the records are in LDS -- nothing to ask for ahead: an iteration reads its own at its top, its run known
        auto take = [&](uint32_t gs, uint32_t n_run, uint32_t off, uint32_t size, uint32_t per) {
            if constexpr (FUSED) {
                const uint32_t i0 = (((g - gs) << 6) + lane) * per;
                cu_i0 = i0;
                const uint32_t r = off + (i0 < n_run ? i0 : 0u) * size;
                cu0 = ld_r4(r);
                cu1 = ld_r4(r + 16u);
            }
        };
        // ---- simple reads (one aligned op, mapped, in range: the packer checked all that): two boundaries, one range,
        //      nothing else can happen.  Four of them per lane.
        for (; g < g_start[1]; g += NWAVE) {
            take(0u, cv.start[1], 0u, SPL_REC_SIMPLE, KS);
            const uint32_t i0 = cu_i0;
            const W4 r0 = cu0, r1 = cu1;
            const uint32_t n_run = cv.start[1];
            const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            int32_t pos[KS], c1[KS];
            spl_dbk e0[KS], e1[KS];
            // (Measured and not kept, round 4: a bit per bucket -- "it holds a site" -- asked first, for the wave's whole stretch of
            //  reads with three scalar words and then per lane, so that only the one read in five whose stretch of buckets is not
            //  empty fetches its two bucket entries (two reads in a hundred lie over a site at all).  Two thirds of the iterations
            //  then issue no gather -- and the launch is not a microsecond shorter, 2-7 % longer on the other samples: the second
            //  question is a second trip.  profiles/r04q_range_simple_filter.txt)
#pragma unroll
            for (uint32_t j = 0; j < KS; ++j) {
                pos[j] = (int32_t)w[2 * j] + cv.shift;
                c1[j] = pos[j] + (int32_t)(w[2 * j + 1] >> 16);
                e0[j] = dbk2(dbk_slot(p, pos[j] - 1));                                               // ---- trip 2
                e1[j] = dbk2(dbk_slot(p, c1[j] - 1));
            }
            fetch_next();
#pragma unroll
            for (uint32_t j = 0; j < KS; ++j) {
                int32_t ua, ub; uint32_t nva, nvb;
                dbk_resolve(p, pos[j] - 1, e0[j], ua, nva);
                dbk_resolve(p, c1[j] - 1, e1[j], ub, nvb);
                const int32_t lo = ua + (int32_t)nva;
                const bool emit = i0 + j < n_run && ub > lo;
                uint32_t arr = 0;
                if (STRANDED) arr = (spl_read_strand(w[2 * j + 1] & 0xffffu, p.stranded) == (uint8_t)'-') ? 1u : 0u;
                if (__any(emit)) commit_range<NARR, AGG, WIN>(p, lds, wbase, emit, lo, ub, arr);
            }
        }
        // ---- once-spliced reads (aligned, N, aligned): the kinds are known, so are the arrays; three ranges
        //      and the junction-table look-up when an end of the junction has rivals.  Two of them per lane.
        for (; g < g_start[2]; g += NWAVE) {
            take(g_start[1], cv.start[2] - cv.start[1], cv.off[1], SPL_REC_MNM, KM);
            const uint32_t i0 = cu_i0;
            const W4 r0 = cu0, r1 = cu1;
            const uint32_t n_run = cv.start[2] - cv.start[1];
            const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            int32_t pos[KM], c0[KM], c1[KM], c2[KM];
            spl_dbk ea[KM], eb[KM], ec[KM], ed[KM];
#pragma unroll
            for (uint32_t j = 0; j < KM; ++j) {
                pos[j] = (int32_t)w[4 * j] + cv.shift;
                c0[j] = pos[j] + (int32_t)(w[4 * j + 1] >> 16); c1[j] = c0[j] + (int32_t)w[4 * j + 2]; c2[j] = c1[j] + (int32_t)w[4 * j + 3];
                ea[j] = dbk2(dbk_slot(p, pos[j] - 1));                                               // ---- trip 2
                eb[j] = dbk3(dbk_slot(p, c0[j] - 1));   // (the junction ends: with the mask of flagged positions)
                ec[j] = dbk3(dbk_slot(p, c1[j] - 1));
                ed[j] = dbk2(dbk_slot(p, c2[j] - 1));
            }
            fetch_next();
#pragma unroll
            for (uint32_t j = 0; j < KM; ++j) {
                const bool alive = i0 + j < n_run;
                uint32_t sidx = 0;
                if (STRANDED) sidx = (spl_read_strand(w[4 * j + 1] & 0xffffu, p.stranded) == (uint8_t)'-') ? 1u : 0u;
                const uint32_t a_me = (STRANDED ? 2u : 1u) + sidx;
                int32_t ua, ub; uint32_t nva, nvb, rv1, rv2;
                auto range = [&](uint32_t arr) {
                    const int32_t lo = ua + (int32_t)nva;
                    const bool em = alive && ub > lo;
                    if (__any(em)) commit_range<NARR, AGG, WIN>(p, lds, wbase, em, lo, ub, arr);
                    ua = ub; nva = nvb;
                };
                dbk_resolve(p, pos[j] - 1, ea[j], ua, nva);
                dbk_resolve(p, c0[j] - 1, eb[j], ub, nvb, rv1);
                range(sidx);                        // block 1
                dbk_resolve(p, c1[j] - 1, ec[j], ub, nvb, rv2);
                range(a_me);                        // the intron
                dbk_resolve(p, c2[j] - 1, ed[j], ub, nvb);
                range(sidx);                        // block 2
                const bool flagged = alive && ((rv1 | rv2) != 0u); // an end of the junction (c0 - 1, c1 - 1) is an end of a junction with rivals
                if (__any(flagged)) {
                    const uint32_t slot = cv.start[1] + i0 + j;
                    const bool room = KM * it_mnm + j < 32u; // (always, with chunks of 4096 reads at most; a read that could not be marked is the literal kernel's)
                    push_front(flagged && (p.combine_mode || !room), slot);
                    if (room) fm_mnm |= (flagged && !p.combine_mode ? 1u : 0u) << (KM * it_mnm + j);
                }
            }
            ++it_mnm;
        }
        // ---- twice-spliced reads (aligned, N, aligned, N, aligned; the record holds the five lengths): six boundaries,
        //      five ranges.  One read per lane.
        for (; g < g_start[3]; g += NWAVE) {
            take(g_start[2], cv.start[3] - cv.start[2], cv.off[2], SPL_REC_M2, 1u);
            const uint32_t i0 = cu_i0;
            const W4 r0 = cu0, r1 = cu1;
            const uint32_t slot = cv.start[2] + i0;     // of the read in its chunk, run order
            const bool alive = i0 < cv.start[3] - cv.start[2];
            const uint2 ra = make_uint2(r0.x, r0.y), rb = make_uint2(r0.z, r0.w), rc = make_uint2(r1.x, r1.y);
            const int32_t pos = (int32_t)ra.x + cv.shift;
            const uint32_t flag = ra.y & 0xffffu;
            const uint32_t la = ra.y >> 16, d1 = rb.x, lb = rb.y & 0xffffu, lc = rb.y >> 16, d2 = rc.x;
            const int32_t c0 = pos + (int32_t)la, c1 = c0 + (int32_t)d1, c2 = c1 + (int32_t)lb;
            const int32_t c3 = c2 + (int32_t)d2, c4 = c3 + (int32_t)lc;
            uint32_t sidx2 = 0;
            if (STRANDED) sidx2 = (spl_read_strand(flag, p.stranded) == (uint8_t)'-') ? 1u : 0u;
            const uint32_t a_me = (STRANDED ? 2u : 1u) + sidx2;
            const spl_dbk f0 = dbk2(dbk_slot(p, pos - 1)), f1 = dbk3(dbk_slot(p, c0 - 1)), f2 = dbk3(dbk_slot(p, c1 - 1)); // ---- trip 2
            const spl_dbk f3 = dbk3(dbk_slot(p, c2 - 1)), f4 = dbk3(dbk_slot(p, c3 - 1)), f5 = dbk2(dbk_slot(p, c4 - 1));
            fetch_next();
            int32_t ua, ub; uint32_t nva, nvb, rvb;
            uint32_t fl1 = 0, fl2 = 0; // junction 1 / 2 has an end with rivals
            auto range = [&](uint32_t arr) {
                const int32_t lo = ua + (int32_t)nva;
                const bool em = alive && ub > lo;
                if (__any(em)) commit_range<NARR, AGG, WIN>(p, lds, wbase, em, lo, ub, arr);
                ua = ub; nva = nvb;
            };
            dbk_resolve(p, pos - 1, f0, ua, nva);
            dbk_resolve(p, c0 - 1, f1, ub, nvb, rvb);
            range(sidx2);                       // block 1
            fl1 |= rvb;
            dbk_resolve(p, c1 - 1, f2, ub, nvb, rvb);
            range(a_me);                        // intron 1
            fl1 |= rvb;
            dbk_resolve(p, c2 - 1, f3, ub, nvb, rvb);
            range(sidx2);                       // block 2
            fl2 |= rvb;
            dbk_resolve(p, c3 - 1, f4, ub, nvb, rvb);
            range(a_me);                        // intron 2
            fl2 |= rvb;
            dbk_resolve(p, c4 - 1, f5, ub, nvb);
            range(sidx2);                       // block 3
            const bool flagged = alive && ((fl1 | fl2) != 0u);
            if (__any(flagged)) {
                push_front(flagged && p.combine_mode, slot);
                push_back(flagged && !p.combine_mode, slot | (fl1 << 14) | (fl2 << 15)); // which junction must be in the table
            }
        }
        for (; g < g_total; g += NWAVE) {
            take(g_start[3], cv.start[4] - cv.start[3], cv.off[3], SPL_REC_OTHER, 1u);
            const uint32_t i0 = cu_i0;
            const W4 r0 = cu0, r1 = cu1;
            const uint32_t slot = cv.start[3] + i0;     // of the read in its chunk, run order
            bool alive = i0 < cv.start[4] - cv.start[3];
            const uint2 ra = make_uint2(r0.x, r0.y), rb = make_uint2(r0.z, r0.w), rc = make_uint2(r1.x, r1.y);
            const int32_t pos = (int32_t)ra.x + cv.shift;
            const uint32_t flag = ra.y & 0xffffu;
            // ---- everything else: {pos, fn, op0, op1, op2 | index of op 0 among the segment's wide ops, number of ops}
            const uint32_t fn = ra.y;
            uint32_t op[SPL_INLINE_OPS] = {rb.x, rb.y, rc.x};
            uint32_t n_ops = rc.y;
            const bool wide = (fn >> SPL_RC_SHIFT) == SPL_RC_WIDE;
            uint32_t o0 = 0; // index of op 0 in the wide ops: known (and needed) only for wide reads
            if (wide) {
                o0 = op[2];
                op[2] = 0xfu; // the third word was the index: ops 2.. are walked in batches below
            }
            const uint32_t n_inline = wide ? 2u : n_ops;
            bool bad = alive && (int32_t)ra.x < 0;
            const bool literal = alive && !bad && (flag & 4u); // fetched as a 1-base record: literal kernel
            alive = alive && !bad && !literal;
            uint32_t sidx = 0;
            if (STRANDED) sidx = (spl_read_strand(flag, p.stranded) == (uint8_t)'-') ? 1u : 0u;
            const uint32_t room = (uint32_t)(SPL_COORD_MAX - (pos < 0 ? 0 : pos));
            uint32_t len = 0;
            int32_t cend[SPL_INLINE_OPS];
            uint32_t kind[SPL_INLINE_OPS]; // 0 none, 1 aligned, 2 N, 3 D
            int32_t pu = 0; uint32_t pnv = 0, prv = 0; // the dpos AT the previous boundary's last base (if pnv) and its rival bit
            bool rival = false;
            // The ops go through in batches of SPL_INLINE_OPS: the first batch is the record's words (every narrow read
            // ends there), further batches of a wide read cost two memory trips each (ops, then buckets).
            uint32_t k_next = n_inline;
            bool mine = true; // this lane has ops in the batch (a narrow read in a wave of wide ones sits the later batches out)
            for (bool first = true;; first = false) {
#pragma unroll
                for (int k = 0; k < SPL_INLINE_OPS; ++k) {
                    // 2 bits per op code: M,=,X -> 1 (aligned), N -> 2, D -> 3, everything else 0 (does not consume the reference)
                    const uint32_t kd = alive ? ((SPL_KIND_TABLE >> (2u * (op[k] & 15u))) & 3u) : 0u;
                    len += kd ? (op[k] >> 4) : 0u;                 // len <= 2^31 before, three lengths < 2^28: no wrap
                    if (kd && len > room) { bad = true; alive = false; }
                    if (first || mine) {
                        cend[k] = pos + (int32_t)len;
                        kind[k] = alive ? kd : 0u;
                    }
                }
                spl_dbk e0 = {0u, 0u, 0u}, ek[SPL_INLINE_OPS]; // ---- trip 2: the boundaries' bucket entries
                if (first) e0 = p.dbucket[dbk_slot(p, pos - 1)];
#pragma unroll
                for (int k = 0; k < SPL_INLINE_OPS; ++k) ek[k] = p.dbucket[dbk_slot(p, cend[k] - 1)];
                if (first) dbk_resolve(p, pos - 1, e0, pu, pnv, prv);
#pragma unroll
                for (int k = 0; k < SPL_INLINE_OPS; ++k) {
                    int32_t u; uint32_t nv, rv;
                    dbk_resolve(p, cend[k] - 1, ek[k], u, nv, rv);
                    const uint32_t kk = (first || mine) ? kind[k] : 0u;
                    const int32_t lo = pu + (int32_t)pnv; // first dpos at or after the op's first base
                    const bool emit = kk != 0u && kk != 3u && u > lo;
                    const uint32_t arr = (kk == 2u ? (STRANDED ? 2u : 1u) : 0u) + sidx;
                    // junction ends: lSite is the previous boundary's position, rSite this one's
                    rival |= (kk == 2u) & ((prv | rv) != 0u);
                    if (__any(emit)) commit_range<NARR, AGG, WIN>(p, lds, wbase, emit, lo, u, arr);
                    if (kk) { pu = u; pnv = nv; prv = rv; }
                }
                const bool more = alive && wide && k_next < n_ops;
                if (!__any(more)) break;
                mine = more;
#pragma unroll
                for (int k = 0; k < SPL_INLINE_OPS; ++k) op[k] = (more && k_next + (uint32_t)k < n_ops) ? cv.wide[o0 + k_next + (uint32_t)k] : 0xfu;
                k_next += (uint32_t)SPL_INLINE_OPS;
            }
            fetch_next(); // (this path asks late: its batches of bucket entries need the registers, and it is the rare one)
            if (bad) atomicOr(p.err, SPL_DEV_ERR_RANGE);
            // (a junction with rivals outside the spliced classes -- indels next to it, three and more junctions -- is the
            //  literal kernel's business)
            const bool flagged = !literal && alive && rival;
            if (__any(literal || flagged)) push_front(literal || flagged, slot);
        }
        // Once- and twice-spliced reads with rivals, the wave's own, lanes dense: the junction table says which sites of the read's
        // window are affected and how (rivals_inline); what it cannot decide joins the literal list.  The list is read
        // from its growing end, so the front list can only ever grow into entries that are done with.
        // (the list, read from its growing end, holds the twice-spliced reads)
        const uint32_t n_m2 = n_back;
        for (uint32_t r0 = 0; r0 < n_m2; r0 += 64u) {
            const uint32_t j = r0 + lane;
            bool undecided = false;
            uint32_t slot = 0;
            if (j < n_m2) {
                const uint32_t entry = s_q[seg0 + SEG - n_back + j];
                slot = entry & 0x3fffu;
                const uint32_t q = cv.off[2] + SPL_REC_M2 * (slot - cv.start[2]);
                const uint2 ra = ld_r2(q), rb = ld_r2(q + 8u), rc = ld_r2(q + 16u);
                const int32_t pos = (int32_t)ra.x + cv.shift;
                uint32_t sidx = 0;
                if (STRANDED) sidx = (spl_read_strand(ra.y & 0xffffu, p.stranded) == (uint8_t)'-') ? 1u : 0u;
                const int32_t c0 = pos + (int32_t)(ra.y >> 16), c1 = c0 + (int32_t)rb.x, c2 = c1 + (int32_t)(rb.y & 0xffffu);
                const int32_t c3 = c2 + (int32_t)rc.x, c4 = c3 + (int32_t)(rb.y >> 16);
                const int32_t jl[2] = {c0 - 1, c2 - 1}, jr[2] = {c1 - 1, c3 - 1};
                const bool jf[2] = {((entry >> 14) & 1u) != 0u, (entry >> 15) != 0u};
                const int32_t blk_a[3] = {pos, c1, c3}, blk_b[3] = {c0 - 1, c2 - 1, c4 - 1};
                undecided = !rivals_inline2<STRANDED, NARR, WIN>(p, lds, wbase, jl, jr, jf, blk_a, blk_b, sidx);
            }
            back_done = r0 + 64u < n_m2 ? r0 + 64u : n_m2;
            if (__any(undecided)) push_front(undecided, slot);
        }
        // Once-spliced reads with rivals: the run once more, for the lanes that marked a read of theirs (fm_mnm).  An iteration's
        // records are asked for while the one before is worked on; what an iteration then waits for is its reads' table slots.
        if (__any(fm_mnm != 0u)) {
            const uint32_t n_run = cv.start[2] - cv.start[1];
            auto fetch_mnm = [&](uint32_t g2) {
                const uint32_t i0 = (((g2 - g_start[1]) << 6) + lane) * KM;
                const uint32_t r = cv.off[1] + (i0 < n_run ? i0 : 0u) * SPL_REC_MNM;
                cu_i0 = i0;
                cu0 = ld_r4(r);
                cu1 = ld_r4(r + 16u);
            };
            uint32_t g2 = g_start[1] + ((wave + NWAVE - (g_start[1] % NWAVE)) % NWAVE), it2 = 0; // (the wave's first iteration of this run: its iterations are g = wave, wave + 4, ... through all runs)
            if constexpr (!FUSED) { if (g2 < g_start[2]) fetch_mnm(g2); }
            for (; g2 < g_start[2] && KM * it2 < 32u; g2 += NWAVE, ++it2) {
                const uint32_t bits = (fm_mnm >> (KM * it2)) & ((1u << KM) - 1u);
                if constexpr (FUSED) { // (records out of LDS: only an iteration with a marked read takes its own)
                    if (!__any(bits != 0u)) continue;
                    fetch_mnm(g2);
                }
                const uint32_t i0 = cu_i0;
                const W4 r0 = cu0, r1 = cu1;
                if constexpr (!FUSED) {
                    fetch_mnm(g2 + NWAVE < g_start[2] ? g2 + NWAVE : g2);
                    if (!__any(bits != 0u)) continue;
                }
                const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                bool live[KM], undecided[KM];
                int32_t pos[KM], c0[KM], c1[KM], c2[KM];
                uint32_t h[KM];
                uint4 ent[KM], first[KM];
#pragma unroll
                for (uint32_t k = 0; k < KM; ++k) {
                    live[k] = ((bits >> k) & 1u) != 0u;
                    undecided[k] = false;
                    pos[k] = (int32_t)w[4 * k] + cv.shift;
                    c0[k] = pos[k] + (int32_t)(w[4 * k + 1] >> 16); c1[k] = c0[k] + (int32_t)w[4 * k + 2]; c2[k] = c1[k] + (int32_t)w[4 * k + 3];
                    h[k] = junction_hash(c0[k] - 1, c1[k] - 1);
                    ent[k] = first[k] = make_uint4(0x80000000u, 0u, 0u, 0u);
                    if (live[k]) { // (the marked reads' slots only: one read in ten is marked where alternative sites are few)
                        const uint4 *hslot = p.jhash + 2u * (h[k] & p.jhash_mask);
                        ent[k] = hslot[0];
                        first[k] = hslot[1];
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < KM; ++k) {
                    uint32_t sidx = 0;
                    if (STRANDED) sidx = (spl_read_strand(w[4 * k + 1] & 0xffffu, p.stranded) == (uint8_t)'-') ? 1u : 0u;
                    const int32_t blk_a[2] = {pos[k], c1[k]}, blk_b[2] = {c0[k] - 1, c2[k] - 1};
                    if (live[k]) undecided[k] = !rivals_inline_from<STRANDED, NARR, WIN>(p, lds, wbase, c0[k] - 1, c1[k] - 1, h[k], ent[k], first[k], blk_a, blk_b, sidx);
                }
#pragma unroll
                for (uint32_t k = 0; k < KM; ++k)
                    if (__any(undecided[k])) push_front(undecided[k], cv.start[1] + i0 + k);
            }
        }
    }
    if constexpr (!FUSED) break;
    else {
        if (tile == tile_last) break;
        ++tile;
        fm_mnm = 0; it_mnm = 0; n_back = 0; back_done = 0; // (the wave's lists of the tile's own reads are through; the front list goes on)
        __syncthreads(); // everybody is through with the tile's records: the next tile's ops take their place
    }
    } // tiles
    if ((tid & 63) == 0) s_qcnt[tid >> 6] = n_front;
    __syncthreads();
    // Hand the chunk's queue over: one returning atomic per workgroup on the counter of its XCD shard (8 counters, so
    // no single word sees more than a few reservations per microsecond), then a dense copy of packed indexes.
    // (Per-chunk regions without any atomic were tried: the range kernel gains 1 %, the literal kernel then has to walk
    // regions of very uneven fill and loses far more.)
    uint32_t q_start[NWAVE + 1];
    q_start[0] = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) q_start[w + 1] = q_start[w] + s_qcnt[w];
    const uint32_t qn = q_start[NWAVE]; // uniform: read after the barrier above
    if (qn) {
        const uint32_t shard = blockIdx.x & 7u;
        if (tid == 0) s_qbase = atomicAdd(&p.queue_n[shard * SPL_COUNTER_STRIDE], qn); // a cache line per counter
        __syncthreads();
        uint32_t *dst = p.queue + (size_t)shard * p.queue_cap + s_qbase;
        const uint32_t first = chunk << CSHIFT;
        for (uint32_t j = tid; j < qn; j += BLOCK) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 1; k < NWAVE; ++k) w += (j >= q_start[k]) ? 1u : 0u;
            uint32_t base = 0;
#pragma unroll
            for (int k = 0; k < NWAVE; ++k) base = (w == (uint32_t)k) ? q_start[k] : base;
            dst[j] = first | (uint32_t)s_q[w * SEG + (j - base)];
        }
    }
    for (int j = tid; j < NARR * (WIN + 1); j += BLOCK) {
        const int32_t v = lds[j];
        if (v) {
            const int arr = j / (WIN + 1), loc = j - arr * (WIN + 1);
            atomicAdd(&p.diff[(int64_t)arr * p.diff_stride + wbase + loc], v);
        }
    }
}

namespace {

#define SPL_CF_JUNC 4
#define SPL_CF_BLK 5
// The same junction table the range kernel uses for single-junction reads, applied to reads with up to SPL_CF_JUNC
// junctions: compSplicing of junction j for site t is "t is listed under (l_j, r_j)", so no partner / competitor list is
// scanned.  A rival is handled under the FIRST junction that lists it (that is where compSplicing becomes true and
// stays true, :494-501): strictly inside that or a later intron -> flanking read (the ME range's +1 is taken back);
// covered with t+1 by an aligned block -> beta1-type.  Anything the table cannot decide exactly (a junction that is not
// in the BED file but touches flagged sites, a rival that is itself a junction end of the read, entries marked complex,
// combine mode) returns false and the read takes rivals_closed_form.  Updates go to the global difference arrays.
template <bool STRANDED>
__device__ __forceinline__ bool rivals_table_path(const spl_count_params &p, int32_t pos, uint32_t flag, const uint32_t *ops, uint32_t n_ops)
{
    if (p.combine_mode) return false;
    int32_t blk_a[SPL_CF_BLK], blk_b[SPL_CF_BLK], jl[SPL_CF_JUNC], jr[SPL_CF_JUNC];
    int n_blk = 0, n_j = 0;
    int32_t cur = pos;
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if (!((SPL_PROG_MASK >> code) & 1u)) continue;
        const int32_t d = (int32_t)(op >> 4);
        const int32_t start = cur;
        cur += d;
        if (code == SPL_OP_N) {
            if (n_j == SPL_CF_JUNC) return false;
#pragma unroll
            for (int j = 0; j < SPL_CF_JUNC; ++j) if (j == n_j) { jl[j] = start - 1; jr[j] = cur - 1; }
            ++n_j;
        } else if (code != SPL_OP_D && d >= 2) {
            if (n_blk == SPL_CF_BLK) return false;
#pragma unroll
            for (int j = 0; j < SPL_CF_BLK; ++j) if (j == n_blk) { blk_a[j] = start; blk_b[j] = cur - 1; }
            ++n_blk;
        }
    }
    if (n_j == 0) return true;
    // pass 1: table entries of all junctions; everything that needs the literal walk bails out before any update.
    // The first probe of every junction is issued before any of them is looked at (one trip for all junctions; at load 1/4
    // the first probe almost always decides).
    uint32_t r_off[SPL_CF_JUNC], r_n[SPL_CF_JUNC];
    uint4 first_probe[SPL_CF_JUNC];
#pragma unroll
    for (int j = 0; j < SPL_CF_JUNC; ++j) {
        uint32_t h = (uint32_t)jl[j < n_j ? j : 0] * 0x9E3779B1u ^ (uint32_t)jr[j < n_j ? j : 0] * 0x85EBCA77u;
        h ^= h >> 15;
        first_probe[j] = p.jhash[2u * (h & p.jhash_mask)];
    }
#pragma unroll
    for (int j = 0; j < SPL_CF_JUNC; ++j) {
        r_off[j] = 0; r_n[j] = 0;
        if (j >= n_j) continue;
        const int32_t l = jl[j], r = jr[j];
        uint32_t h = (uint32_t)l * 0x9E3779B1u ^ (uint32_t)r * 0x85EBCA77u;
        h ^= h >> 15;
        bool found = false;
        uint4 ent = first_probe[j];
        for (int probe = 0; probe < 8; ++probe) {
            if (probe) ent = p.jhash[2u * ((h + (uint32_t)probe) & p.jhash_mask)];
            if ((int32_t)ent.x == l && (int32_t)ent.y == r) { found = true; break; }
            if (ent.x == 0x80000000u) break;
        }
        if (found) {
            if (ent.w & SPL_JF_COMPLEX) return false;
            if ((ent.w & SPL_JF_COUNT_MASK) > 16u) return false;
            if (!STRANDED && (ent.w & SPL_JF_MULTIROW)) return false;
            r_off[j] = ent.z; r_n[j] = ent.w & SPL_JF_COUNT_MASK;
        } // (not in the table: no site has this junction as a rival's -- the table is complete)
    }
    // pass 2: apply
    const uint32_t want = (spl_read_strand(flag, STRANDED ? p.stranded : 1) == (uint8_t)'-') ? 2u : 1u;
    // (corrections go straight to the counters of the rival's row, not to the difference arrays: after the range kernel
    //  nothing writes the difference arrays any more, so their block sums can be taken inside this very launch)
#pragma unroll
    for (int j = 0; j < SPL_CF_JUNC; ++j) {
        for (uint32_t i = 0; i < r_n[j]; ++i) {
            const uint4 rv = p.jrivals[2u * (r_off[j] + i)];
            const uint4 rx = p.jrivals[2u * (r_off[j] + i) + 1u]; // {row of t, its partner list offset, length, -}: same line, same trip
            bool earlier = false; // listed under an earlier junction of this read: handled there
#pragma unroll
            for (int j2 = 0; j2 < SPL_CF_JUNC; ++j2)
                if (j2 < j) for (uint32_t i2 = 0; i2 < r_n[j2]; ++i2) earlier |= (p.jrivals[2u * (r_off[j2] + i2)].y == rv.y);
            if (earlier) continue;
            const int32_t t = (int32_t)rv.x;
            const bool strand_ok = !STRANDED || (rv.y >> 30) == want;
            int inside = -1;
            bool cov = false, alpha = false;
            int32_t pu = 0;
#pragma unroll
            for (int a = 0; a < SPL_CF_JUNC; ++a) {
                if (a >= n_j) break;
                if (t > jl[a] && t < jr[a]) inside = a;
                if (jl[a] == t) { pu = jr[a]; alpha = true; } // :487-492, in op order
                if (jr[a] == t) { pu = jl[a]; alpha = true; }
            }
#pragma unroll
            for (int b = 0; b < SPL_CF_BLK; ++b) cov |= (b < n_blk) && (blk_a[b] <= t) && (t + 1 <= blk_b[b]);
            const bool beta1type = !alpha && inside < 0 && cov && strand_ok;
            if (alpha || beta1type) {
                // double counts: set(partners) & set(spliceSites), minus partnerUsed for the alpha case (:519-527, :544-551)
                for (uint32_t e2 = 0; e2 < rx.z; ++e2) {
                    const int32_t pp = p.part_pos[rx.y + e2];
                    bool is_end = false;
#pragma unroll
                    for (int a = 0; a < SPL_CF_JUNC; ++a) is_end |= (a < n_j) && (pp == jl[a] || pp == jr[a]);
                    if (is_end && !(alpha && pp == pu)) agg_add(&p.dbl[rx.y + e2], 1);
                }
                if (beta1type) { // beta1-type (:544-556): the ranges counted it as beta1
                    agg_add(&p.beta1[rx.x], -1);
                    agg_add(&p.beta2s_reads[rx.x], 1);
                }
            } else if (inside >= j && strand_ok) { // flanking (:503-505, :529): the ME range counted it, `process` does not
                agg_add(&p.beta2s_reads[rx.x], -1);
            }
        }
    }
    return true;
}

} // namespace

// The literal kernel: SPL_LITERAL_WAVES one-wave workgroups stride over the dense queue, one queued read per lane.
// Neighbouring lanes hold neighbouring reads, which in coordinate-sorted input cross the same junction and therefore
// update the same counters: those updates are merged across the wave before they reach HBM (agg_add), because one
// counter word takes only so many atomics per microsecond no matter how many CUs send them.
template <bool STRANDED>
__global__ __launch_bounds__(64) void spl_count_literal_kernel(const spl_count_params p, const spl_queue_params q)
{
    __shared__ uint32_t s_ops[64][SPL_PACK_SCAN_OPS + 1]; // the lane's CIGAR: rebuilt from its record or fetched in one trip (+1: banks)
    // The other copy of the counter region, for the next counting pass (one 16-byte store per lane or so).
    for (size_t j = (size_t)blockIdx.x * 64 + threadIdx.x; j < q.clear_n16; j += (size_t)gridDim.x * 64) q.clear_region[j] = make_uint4(0, 0, 0, 0);
    // Then the block sums of the difference arrays for the scan that follows: the arrays are final once the range kernel
    // is done (nothing in this kernel writes them), most of this kernel's waves have no queue entry to work on, and the few
    // that have are a chain of dependent loads that this short streaming job overlaps with.
    // (taken from the far end of the grid: the queue is walked from the near end, and a wave with queue entries should
    // start its chain at once: 26 -> 23 us)
    for (uint32_t b = gridDim.x - 1u - blockIdx.x; b < (uint32_t)(q.scan_blocks * q.scan_arrays); b += gridDim.x) {
        const uint32_t arr = b / (uint32_t)q.scan_blocks, blk = b - arr * (uint32_t)q.scan_blocks;
        const int32_t *d = q.diff + (int64_t)arr * q.diff_stride;
        const int32_t base = (int32_t)blk * SPL_SCAN_BLOCK;
        int32_t acc = 0;
        for (int j = (int)threadIdx.x; j < SPL_SCAN_BLOCK; j += 64) acc += (base + j < q.n_dpos) ? d[base + j] : 0;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (threadIdx.x == 0) q.block_sums[b] = acc;
    }
    // The 8 shard regions are walked as ONE index space (a wave must not pay the latency chain once per shard).
    // Entries are packed indexes: the read is taken from the range kernel's own arrays (one trip), its ops are inline
    // or start at the stored offset.
    uint32_t start[9];
    start[0] = 0;
#pragma unroll
    for (int sh = 0; sh < 8; ++sh) start[sh + 1] = start[sh] + q.queue_n[sh * SPL_COUNTER_STRIDE];
    const uint32_t total = start[8];
    if (blockIdx.x == 0 && threadIdx.x == 0) *q.queue_total = total;
    {
        // A short queue is spread thin, 16 entries per wave: a wave is as slow as its slowest lane (trips differ with the
        // number of junctions and rivals), and there are far more waves than work (26 -> 22 us for 9 k reads).  A long one
        // (combine mode: every flagged read) fills the waves.
        const uint32_t lanes = total <= gridDim.x * 16u ? 16u : 64u;
        for (uint32_t g = blockIdx.x * lanes + threadIdx.x; threadIdx.x < lanes && g < total; g += gridDim.x * lanes) {
            uint32_t shard = 0;
#pragma unroll
            for (int sh = 1; sh < 8; ++sh) shard += (g >= start[sh]) ? 1u : 0u;
            uint32_t base = 0;
#pragma unroll
            for (int sh = 0; sh < 8; ++sh) base = (shard == (uint32_t)sh) ? start[sh] : base;
            const uint32_t entry = q.queue[(size_t)shard * q.queue_cap + (g - base)]; // chunk << SPL_CHUNK_SHIFT | slot
            uint32_t *row = s_ops[threadIdx.x];
            ReadView rv;
            if (q.cells) { // a fused pass: the read as it lies in the arrays, every op of it
                const spl_layout_chunk ch = q.cells[entry >> q.chunk_shift];
                const int64_t i = (ch.lo & ~(int64_t)((1u << q.chunk_shift) - 1u)) + (int64_t)(entry & ((1u << q.chunk_shift) - 1u));
                const int32_t pos0 = q.src.pos[i];
                const uint32_t c0 = q.src.cig_off[i], c1 = q.src.cig_off[i + 1];
                rv.flag = q.src.flag[i];
                rv.n_ops = c1 - c0;
                rv.ops = q.src.cigar + c0;
                rv.neg = pos0 < 0;
                rv.pos = pos0 + ch.shift;
            } else {
                const ChunkView cv = chunk_view(p.chunk_meta + (entry >> q.chunk_shift));
                rv = read_at(cv, entry & ((1u << q.chunk_shift) - 1u), row);
            }
            const int32_t pos = rv.pos;
            const uint32_t flag = rv.flag;
            uint32_t n_ops = rv.n_ops;
            const uint32_t *ops = rv.ops;
            if (ops != row && n_ops <= (uint32_t)SPL_PACK_SCAN_OPS) { // a wide read's ops are walked several times below: all of them in one trip, then LDS
                uint32_t w[SPL_PACK_SCAN_OPS];
#pragma unroll
                for (int k = 0; k < SPL_PACK_SCAN_OPS; ++k) w[k] = ((uint32_t)k < n_ops) ? ops[k] : 0xfu;
#pragma unroll
                for (int k = 0; k < SPL_PACK_SCAN_OPS; ++k) row[k] = w[k];
                ops = row;
            }
            if (rv.neg) { atomicOr(p.err, SPL_DEV_ERR_RANGE); continue; }
            int64_t ref_len; bool hn;
            spl_read_extent(ops, n_ops, &ref_len, &hn);
            if ((int64_t)pos + ref_len > (int64_t)SPL_COORD_MAX) { atomicOr(p.err, SPL_DEV_ERR_RANGE); continue; }
            if (flag & 4u) { unmapped_read<STRANDED>(p, pos, flag, ops, n_ops); continue; }
            if (rivals_table_path<STRANDED>(p, pos, flag, ops, n_ops)) continue;
            rivals_literal<STRANDED>(p, pos, flag, ops, n_ops, (int32_t)((int64_t)pos + (ref_len > 0 ? ref_len : 1) - 1));
        }
    }
}

// =========================================================================================================
// Difference arrays -> counters: two tiny launches (block sums, then offset + local inclusive scan).
// =========================================================================================================
// =========================================================================================================
// findBeta2Counts + calculateSSE, one site per lane.  IEEE binary64, compiled with -ffp-contract=off: every
// operation below is one correctly rounded operation in the reference's order, so the doubles are the ones
// CPython produces (int/int true division included for |values| < 2^53).
// =========================================================================================================
// findBeta2Counts for site s given its read-derived beta2Simple count (:581-623)
__device__ __forceinline__ void sse_site_core(const spl_sse_params &p, int64_t s, uint32_t b2s_reads, int64_t &b2simple, int64_t &cryptic,
                                              double &weighted)
{
    const int64_t t = p.site_pos[s];
    b2simple = b2s_reads;
    cryptic = 0;
    weighted = 0.0;
    const int64_t total_alpha = p.alpha[s];
    const uint32_t e0 = p.part_off[s], e1 = p.part_off[s + 1];
    for (uint32_t e = e0; e < e1; ++e) { // for pSite in Partners (:590)
        const int32_t ps = p.part_site[e];
        if (ps < 0) continue;
        const int64_t ppos = p.site_pos[ps];
        int64_t doubles = p.dbl[e];
        bool have_key = doubles != 0;
        // PartnerBeta2DoubleCounts is keyed by the partner's POSITION (:598): what an earlier partner at the same position added to
        // it is there when this one is looked at (:608).  Two partner sites at one position are two edges (a stranded analysis of a
        // BED file with strands that are none, a junction whose ends coincide); for every other row e2 == e is the only turn.
        for (uint32_t e2 = e0; e2 <= e; ++e2) {
            const int32_t qs = e2 == e ? ps : p.part_site[e2];
            if (qs < 0 || (e2 != e && p.site_pos[qs] != ppos)) continue;
            const uint32_t f0 = p.part_off[qs], f1 = p.part_off[qs + 1];
            for (uint32_t f = f0; f < f1; ++f) { // pSite.getPartnerCounts().items() (:592)
                const int64_t cpos = p.part_pos[f];
                if ((ppos > t && cpos < t) || (ppos < t && cpos > t)) { // junction (pSite, c) flanks t (:594-599)
                    bool listed = false; // (a dict: a position the partner's own edges name twice is one item)
                    for (uint32_t f2 = f0; f2 < f; ++f2) listed |= p.part_pos[f2] == p.part_pos[f];
                    if (listed) continue;
                    const int64_t cnt = p.edge_cnt[f];
                    if (e2 == e) b2simple += cnt;
                    doubles += cnt;
                    have_key = true;
                }
            }
        }
        const int64_t shared = p.edge_cnt[e];          // PartnerCounts[pSite.pos] (:604)
        int64_t b2 = p.alpha[ps] - shared;             // :606
        if (have_key) { b2 -= doubles; if (b2 < 0) b2 = 0; } // :608-611 subIntNoNeg
        cryptic += b2;                                 // :613
        double w = 0.0;                                // trueDivCatchZero (:562-572)
        if ((double)total_alpha > 0.0) w = (double)shared / (double)total_alpha;
        const double wb2 = (double)b2 * w;             // :618
        weighted = weighted + wb2;                     // :619
    }
}

// calculateSSE (:626-639)
__device__ __forceinline__ double sse_site_value(int64_t total_alpha, uint32_t beta1, int64_t b2simple, double weighted, bool with_cryptic)
{
    const int64_t betas_int = (int64_t)beta1 + b2simple; // :631
    double value = 0.0;
    if (with_cryptic) {
        const double betas = (double)betas_int + weighted;     // :635
        const double denom = (double)total_alpha + betas;      // :637
        if (denom > 0.0) value = (double)total_alpha / denom;
    } else {
        const int64_t denom = total_alpha + betas_int;
        if ((double)denom > 0.0) value = (double)total_alpha / (double)denom;
    }
    return value;
}

__global__ __launch_bounds__(256) void spl_sse_kernel(const spl_sse_params p)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.n_sites) return;
    int64_t b2simple, cryptic;
    double weighted;
    sse_site_core(p, s, p.beta2s_reads[s], b2simple, cryptic, weighted);
    p.beta2_simple[s] = b2simple;
    p.beta2_cryptic[s] = cryptic;
    p.beta2_weighted[s] = weighted;
    p.sse[s] = sse_site_value(p.alpha[s], p.beta1[s], b2simple, weighted, p.cryptic != 0);
}

// Difference arrays -> counters (and, fused, findBeta2Counts + calculateSSE of every row)
__global__ __launch_bounds__(256) void spl_scan_apply_kernel(const spl_scan_params p)
{
    __shared__ int32_t red[4][4];   // [array][wave]
    __shared__ int32_t wave_tot[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t base = blockIdx.x * SPL_SCAN_BLOCK;
    constexpr int Q = SPL_SCAN_BLOCK / 256; // consecutive distinct positions per thread
    int32_t run[4] = {0, 0, 0, 0};
    // offset of this block = sum of the sums of all blocks before it
    for (int a = 0; a < p.n_arrays; ++a) {
        int32_t acc = 0;
        for (int j = tid; j < (int)blockIdx.x; j += 256) acc += p.block_sums[a * p.n_blocks + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) red[a][wave] = acc;
    }
    // each thread owns Q consecutive distinct positions
    int32_t v[4][Q];
    const int32_t d0 = base + tid * Q;
    int32_t first_row[Q + 1]; // (asked for before the barrier: one trip less on the way to the rows)
#pragma unroll
    for (int q = 0; q <= Q; ++q) first_row[q] = p.dpos_first_row[d0 + q < p.n_dpos ? d0 + q : p.n_dpos];
    for (int a = 0; a < p.n_arrays; ++a) {
        const int32_t *d = p.diff + (int64_t)a * p.diff_stride;
        int32_t s = 0;
        for (int q = 0; q < Q; ++q) {
            const int32_t r = d0 + q;
            s += (r < p.n_dpos) ? d[r] : 0;
            v[a][q] = s; // inclusive within the thread
        }
        int32_t incl = s; // wave-inclusive scan of thread totals
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        run[a] = incl - s; // exclusive prefix of this thread inside its wave
        if (lane == 63) wave_tot[a][wave] = incl;
    }
    __syncthreads();
    for (int a = 0; a < p.n_arrays; ++a) {
        int32_t off = red[a][0] + red[a][1] + red[a][2] + red[a][3];
        for (int w = 0; w < wave; ++w) off += wave_tot[a][w];
        run[a] += off;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int32_t d = d0 + q;
        if (d >= p.n_dpos) break;
        for (int32_t r = first_row[q]; r < first_row[q + 1]; ++r) { // the rows at this position
            int32_t b1, me;
            if (p.n_arrays == 2) { b1 = run[0] + v[0][q]; me = run[1] + v[1][q]; }
            else {
                const uint8_t f = p.site_flags[r];
                if (f & SPL_SF_PLUS) { b1 = run[0] + v[0][q]; me = run[2] + v[2][q]; }
                else if (f & SPL_SF_MINUS) { b1 = run[1] + v[1][q]; me = run[3] + v[3][q]; }
                else { b1 = 0; me = 0; } // a row without strand matches no read in a stranded run (:406)
            }
            // (the literal kernel may have left corrections in the counters: add, do not store)
            const uint32_t v1 = p.beta1[r] + (uint32_t)b1, v2 = p.beta2s_reads[r] + (uint32_t)me;
            if (b1) p.beta1[r] = v1;
            if (me) p.beta2s_reads[r] = v2;
            if (p.with_sse) { // findBeta2Counts + calculateSSE of the row while its counters are at hand, for both settings of
                              // --beta2Cryptic (they differ in the last division only): no launch of its own for Step 3's tail
                int64_t b2simple, cryptic;
                double weighted;
                sse_site_core(p.sse, r, v2, b2simple, cryptic, weighted);
                p.sse.beta2_simple[r] = b2simple;
                p.sse.beta2_cryptic[r] = cryptic;
                p.sse.beta2_weighted[r] = weighted;
                const int64_t total_alpha = p.sse.alpha[r];
                p.sse.sse[r] = sse_site_value(total_alpha, v1, b2simple, weighted, false);
                p.sse_with_cryptic[r] = sse_site_value(total_alpha, v1, b2simple, weighted, true);
            }
        }
    }
}

// =========================================================================================================
// Junction table of a read set (what the pipeline otherwise gets from `regtools junctions extract`).
// Every N op of every mapped read is a junction (l, r) in SpliSER's site convention (l = last base before the intron,
// r = last intronic base; SpliSER_v0_1_8.py:482-483); the table holds, per distinct (l, r[, read strand]), the number
// of reads carrying it and the longest anchors seen on either side (reference bases of the read between the junction and
// the previous / next N op or read end -- the block sizes of a BED12 junction line).  Open addressing on a 64-bit key,
// one insert per N op; lanes of a wave that insert the same key (neighbours in a sorted file) merge first.
__global__ __launch_bounds__(256) void spl_junction_kernel(const spl_chunk_meta *chunk_meta, int stranded, uint32_t min_anchor, uint32_t min_intron,
                                                           uint32_t max_intron, unsigned long long *keys, uint32_t *vals, uint32_t mask,
                                                           int32_t *err)
{
    __shared__ uint32_t s_ops[256][5]; // a lane's short CIGAR, rebuilt from its packed record
    const ChunkView cv = chunk_view(chunk_meta + blockIdx.x); // one workgroup per chunk
  for (uint32_t base = 0; base < cv.start[SPL_RC_RUNS]; base += 256u) {
    const uint32_t slot = base + threadIdx.x;
    const bool valid = slot < cv.start[SPL_RC_RUNS];
    uint32_t n_ops = 0, fl = 0;
    int32_t cur = 0;
    const uint32_t *ops = s_ops[threadIdx.x];
    bool neg = false;
    if (valid) { const ReadView rv = read_at(cv, slot, s_ops[threadIdx.x]); n_ops = rv.n_ops; cur = rv.pos; fl = rv.flag; ops = rv.ops; neg = rv.neg; }
    const bool skip = !valid || (fl & 4u) || neg; // unmapped records carry no junctions
    const unsigned long long sbit = (stranded && spl_read_strand(fl, stranded) == (uint8_t)'-') ? 1ull : 0ull;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t k = 0;
    uint32_t before = 0; // reference bases since the previous N op (or the read start)
    // all lanes stay in the loop while any lane has ops left: the merge below is a wave-wide conversation
    while (__any(!skip && k < n_ops)) {
        bool have = false;
        unsigned long long key = ~0ull;
        uint32_t a_left = 0, a_right = 0;
        while (!skip && k < n_ops && !have) {
            const uint32_t op = ops[k];
            const uint32_t code = op & 15u, d = op >> 4;
            ++k;
            if (!((SPL_PROG_MASK >> code) & 1u)) continue;
            if ((int64_t)cur + d > (int64_t)SPL_COORD_MAX) { atomicOr(err, SPL_DEV_ERR_RANGE); k = n_ops; break; }
            cur += (int32_t)d;
            if (code != SPL_OP_N) { before += d; continue; }
            // anchor on the right: reference bases up to the next N op or the end of the read
            uint32_t after = 0;
            for (uint32_t k2 = k; k2 < n_ops; ++k2) {
                const uint32_t op2 = ops[k2];
                const uint32_t c2 = op2 & 15u;
                if (c2 == SPL_OP_N) break;
                if ((SPL_PROG_MASK >> c2) & 1u) after += op2 >> 4;
            }
            const int32_t l = cur - (int32_t)d - 1, r = cur - 1;
            key = ((unsigned long long)(uint32_t)l << 32) | ((unsigned long long)(uint32_t)r << 1) | sbit;
            a_left = before;
            a_right = after;
            before = 0;
            // the caller's policy (regtools' -a / -m / -M): a read supports a junction only with both anchors long enough
            have = a_left >= min_anchor && a_right >= min_anchor && d >= min_intron && (max_intron == 0u || d <= max_intron);
        }
        // merge equal keys across the wave: one insert per distinct key
        unsigned long long todo = __ballot(have);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)key, leader), khi = (uint32_t)__shfl((int)(uint32_t)(key >> 32), leader);
            const bool same = have && (uint32_t)key == klo && (uint32_t)(key >> 32) == khi;
            const unsigned long long grp = __ballot(same);
            uint32_t ml = same ? a_left : 0u, mr = same ? a_right : 0u;
            for (int off = 32; off > 0; off >>= 1) { // wave max over the group (others contribute 0)
                const uint32_t xl = (uint32_t)__shfl_xor((int)ml, off), xr = (uint32_t)__shfl_xor((int)mr, off);
                ml = xl > ml ? xl : ml;
                mr = xr > mr ? xr : mr;
            }
            if (lane == leader) {
                const unsigned long long kk = ((unsigned long long)khi << 32) | klo;
                uint32_t h = (uint32_t)(kk * 0x9E3779B97F4A7C15ull >> 32);
                for (uint32_t probe = 0;; ++probe) {
                    const uint32_t slot = (h + probe) & mask;
                    const unsigned long long old = atomicCAS(&keys[slot], ~0ull, kk);
                    if (old == ~0ull || old == kk) {
                        atomicAdd(&vals[3u * slot], (uint32_t)__popcll(grp));
                        atomicMax(&vals[3u * slot + 1u], ml);
                        atomicMax(&vals[3u * slot + 2u], mr);
                        break;
                    }
                    if (probe > mask) { atomicOr(err, SPL_DEV_ERR_TABLE); break; } // cannot happen: the table has a free slot per op
                }
            }
            todo &= ~grp;
        }
    }
  }
}

// Non-empty slots -> dense arrays (order arbitrary; the host sorts).
__global__ __launch_bounds__(256) void spl_junction_compact_kernel(const unsigned long long *keys, const uint32_t *vals, uint32_t n_slots,
                                                                   unsigned long long *out_keys, uint32_t *out_vals, uint32_t *n_out)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n_slots) return;
    const unsigned long long k = keys[j];
    if (k == ~0ull) return;
    const uint32_t at = atomicAdd(n_out, 1u);
    out_keys[at] = k;
    out_vals[3u * at] = vals[3u * j];
    out_vals[3u * at + 1u] = vals[3u * j + 1u];
    out_vals[3u * at + 2u] = vals[3u * j + 2u];
}

// The position index of a site table (see dbk_slot / dbk_resolve), built where it lives: one thread per 32 bp bucket finds the
// first distinct position at or after the bucket start by bisection and collects the occupancy mask of the positions inside;
// the same for the flagged positions (ends of junctions that have rivals: sorted, not necessarily sites) and their mask.
// (On the host this was a serial sweep over all buckets: 0.3 s for a mammalian genome.)
__global__ __launch_bounds__(256) void spl_build_dbuckets_kernel(const int32_t *site_pos, const int32_t *dpos_first_row, int32_t n_dpos,
                                                                 const int32_t *flag_pos, int32_t n_flag, int32_t dbase, uint32_t n_dbuckets,
                                                                 spl_dbk *out)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= n_dbuckets) return;
    const int64_t start = (int64_t)dbase + ((int64_t)b << 5);
    int32_t lo = 0, hi = n_dpos;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)site_pos[dpos_first_row[mid]] < start) lo = mid + 1; else hi = mid;
    }
    uint32_t mask = 0, rm = 0;
    for (int32_t j = lo; j < n_dpos; ++j) {
        const int64_t pj = site_pos[dpos_first_row[j]];
        if (pj >= start + 32) break;
        mask |= 1u << (pj - start);
    }
    int32_t fl = 0, fh = n_flag;
    while (fl < fh) {
        const int32_t mid = fl + ((fh - fl) >> 1);
        if ((int64_t)flag_pos[mid] < start) fl = mid + 1; else fh = mid;
    }
    for (int32_t j = fl; j < n_flag; ++j) {
        const int64_t pj = flag_pos[j];
        if (pj >= start + 32) break;
        rm |= 1u << (pj - start);
    }
    spl_dbk e;
    e.first = (uint32_t)lo; e.occ = mask; e.rival = rm;
    out[b] = e;
}

// Everything a counting pass starts from zero (the counter region of the site table: counters, difference arrays, queue
// counters, error word; 16-byte aligned, a multiple of 16 bytes).  Normally the literal kernel of the previous pass has
// cleared the copy a pass starts on; this launch is for the passes that have no such predecessor.
__global__ __launch_bounds__(256) void spl_clear_kernel(uint4 *region, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n16; j += stride) region[j] = make_uint4(0, 0, 0, 0);
}

// ---- launchers (called from spl_capi.cpp through spl_device.h) ------------------------------------------

extern "C" int spl_dev_launch_junctions(const spl_chunk_meta *chunk_meta, uint32_t n_chunks,
                                        int stranded, uint32_t min_anchor, uint32_t min_intron, uint32_t max_intron,
                                        unsigned long long *keys, uint32_t *vals, uint32_t n_slots,
                                        unsigned long long *out_keys, uint32_t *out_vals, uint32_t *n_out, int32_t *err, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(keys, 0xff, (size_t)n_slots * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(vals, 0, (size_t)n_slots * 12, st);
    if (e == hipSuccess) e = hipMemsetAsync(n_out, 0, 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(err, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    if (n_chunks > 0)
        hipLaunchKernelGGL(spl_junction_kernel, dim3(n_chunks), dim3(256), 0, st, chunk_meta, stranded, min_anchor, min_intron, max_intron, keys,
                           vals, n_slots - 1u, err);
    hipLaunchKernelGGL(spl_junction_compact_kernel, dim3((n_slots + 255u) / 256u), dim3(256), 0, st, keys, vals, n_slots, out_keys, out_vals, n_out);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_build_dbuckets(const int32_t *site_pos, const int32_t *dpos_first_row, int32_t n_dpos, const int32_t *flag_pos,
                                             int32_t n_flag, int32_t dbase, uint32_t n_dbuckets, spl_dbk *out, void *stream)
{
    if (n_dbuckets == 0) return 0;
    hipLaunchKernelGGL(spl_build_dbuckets_kernel, dim3((n_dbuckets + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, site_pos, dpos_first_row,
                       n_dpos, flag_pos, n_flag, dbase, n_dbuckets, out);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_clear(void *region, size_t bytes, void *stream)
{
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + 256 * 4 - 1) / (256 * 4); // four stores per thread
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(spl_clear_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint4 *)region, n16);
    return (int)hipGetLastError();
}

// ev_start / ev_stop (may be null): HIP events that take the kernel's own start and end time -- through hipExtLaunchKernelGGL,
// i.e. without marker packets of their own in the queue (two hipEventRecord calls around a launch cost the step 3 us).
extern "C" int spl_dev_launch_count(const spl_count_params *p, const spl_hot_params *h, int variant, void *stream, int *grid_out, int *lds_out,
                                    void *ev_start, void *ev_stop)
{
    *grid_out = 0;
    *lds_out = 0;
    if (p->n_reads <= 0 || p->n_sites <= 0) return 0;
    // grid = 8 * ceil(n_chunks / 8) so that every XCD's share has the same number of slots; the range kernel's slots are the
    // chunk order's (spl_chunk_order_kernel: whole blocks of 8 chunks per share, so a few more)
    const uint32_t slots = variant == 1 ? ((p->n_chunks + 7u) / 8u) * 8u : h->n_chunks;
    const uint32_t grid = slots;
    *grid_out = (int)grid;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
    if (variant == 1) {
        *lds_out = 2 * SPL_WIN * 4 + SPL_BLOCK * 5 * 4 + 4;
        if (p->stranded) hipExtLaunchKernelGGL(spl_count_pairs_kernel<true>, dim3(grid), dim3(SPL_BLOCK), 0, st, e0, e1, 0, *p);
        else hipExtLaunchKernelGGL(spl_count_pairs_kernel<false>, dim3(grid), dim3(SPL_BLOCK), 0, st, e0, e1, 0, *p);
    } else {
        const bool big = h->chunk_shift == SPL_CHUNK_BIG_SHIFT;
        *lds_out = (p->stranded ? 4 * ((h->cells ? SPL_WIN_STRANDED_FUSED : SPL_WIN_STRANDED) + 1) : 2 * (SPL_WIN + 1)) * 4 + SPL_WAVES * SPL_WAVE_READS * 2 + 4 * SPL_WAVES + 4; // difference windows + the waves' lists
        const bool agg = (variant & 2) != 0;
#define SPL_LAUNCH_RANGES(S, A, B) hipExtLaunchKernelGGL((spl_count_ranges_kernel<S, A, B, false>), dim3(grid), dim3(SPL_BLOCK), 0, st, e0, e1, 0, *h)
        if (h->cells) { // the fused pass: straight from the BAM-native arrays
            if (agg) return (int)hipErrorInvalidValue;
            *lds_out += (int)SPL_LAYOUT_SLOT(SPL_TILE_FUSED) + 2 * SPL_TILE_FUSED + (SPL_BLOCK_FUSED / 64) * SPL_WAVE_READS_FUSED * 2 - SPL_WAVES * SPL_WAVE_READS * 2;
#define SPL_LAUNCH_FUSED(S, B) hipExtLaunchKernelGGL((spl_count_ranges_kernel<S, false, B, true>), dim3(grid), dim3(SPL_BLOCK_FUSED), 0, st, e0, e1, 0, *h)
            if (p->stranded) { if (big) SPL_LAUNCH_FUSED(true, true); else SPL_LAUNCH_FUSED(true, false); }
            else { if (big) SPL_LAUNCH_FUSED(false, true); else SPL_LAUNCH_FUSED(false, false); }
#undef SPL_LAUNCH_FUSED
        } else if (p->stranded) {
            if (agg) { if (big) SPL_LAUNCH_RANGES(true, true, true); else SPL_LAUNCH_RANGES(true, true, false); }
            else { if (big) SPL_LAUNCH_RANGES(true, false, true); else SPL_LAUNCH_RANGES(true, false, false); }
        } else {
            if (agg) { if (big) SPL_LAUNCH_RANGES(false, true, true); else SPL_LAUNCH_RANGES(false, true, false); }
            else { if (big) SPL_LAUNCH_RANGES(false, false, true); else SPL_LAUNCH_RANGES(false, false, false); }
        }
#undef SPL_LAUNCH_RANGES
    }
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_literal(const spl_count_params *p, const spl_queue_params *q, void *stream)
{
    if (p->n_reads <= 0 || p->n_sites <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t grid = SPL_LITERAL_WAVES;
    if (p->stranded) hipLaunchKernelGGL(spl_count_literal_kernel<true>, dim3(grid), dim3(64), 0, st, *p, *q);
    else hipLaunchKernelGGL(spl_count_literal_kernel<false>, dim3(grid), dim3(64), 0, st, *p, *q);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_scan(const spl_scan_params *p, void *stream)
{
    if (p->n_dpos <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // (the block sums were taken by the literal kernel's launch)
    hipLaunchKernelGGL(spl_scan_apply_kernel, dim3(p->n_blocks), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_sse(const spl_sse_params *p, void *stream)
{
    if (p->n_sites <= 0) return 0;
    const uint32_t grid = (uint32_t)((p->n_sites + 255) / 256);
    hipLaunchKernelGGL(spl_sse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
