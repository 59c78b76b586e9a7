// spl_kernels.hip -- CDNA4 (gfx950, wave64) kernels of the SpliSER `process` hot path.
//
//   spl_count_ranges_kernel  the checkBam loop (SpliSER_v0_1_8.py:408-559, called per site from processSites
//                            :686-688) turned inside out: every read walks its CIGAR ONCE and turns each op
//                            into a range of site rows -- an aligned block makes every row it covers (t and
//                            t+1) a beta1 read, an N op makes every row strictly inside it a
//                            mutually-exclusive (beta2Simple) read -- recorded as +1/-1 in LDS-privatised
//                            difference arrays.  Cost per read is O(ops), whatever the number of sites a
//                            500 kb intron spans.  Only the sites whose outcome can depend on their own
//                            partner / competitor lists (rivals of the read's junction ends) are classified
//                            one by one with the literal state machine (spl_classify.h) and corrected.
//   spl_scan_*_kernel        prefix sums that turn the difference arrays into beta1 / beta2Simple counters.
//   spl_count_pairs_kernel   the literal formulation: every (read, site) pair through spl_classify_pair.
//                            Used for tables whose partner links are not mutual (combine gap-fill queries)
//                            and as an on-device cross-check of the range kernel in the tests.
//   spl_sse_kernel           findBeta2Counts + calculateSSE (SpliSER_v0_1_8.py:581-639), one site per lane.
//
// Integer / indexing work: no MFMA.  The roofline that bounds the count kernels is HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPL_HD __device__ __forceinline__
#include "spl_classify.h"
#include "spl_device.h"

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
    __shared__ int32_t s_wbase;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t chunk = my_chunk();
    const bool live = chunk < p.n_chunks;
    const int64_t chunk_base = (int64_t)chunk * SPL_CHUNK;

    for (int j = tid; j < 2 * SPL_WIN; j += SPL_BLOCK) lds[j] = 0u;
    if (tid == 0) s_wbase = live ? first_site_at_or_after(p, p.r_pos[chunk_base]) : 0;
    __syncthreads();
    const int32_t wbase = s_wbase;
    const int32_t n_sites = p.n_sites;

    if (live) {
        for (int it = 0; it < SPL_RPT; ++it) {
            const int64_t i = chunk_base + (int64_t)it * SPL_BLOCK + tid;
            bool valid = i < p.n_reads;
            int32_t pos = 0, end = -1, s = n_sites;
            uint32_t flag = 0, o0 = 0, n_ops = 0;
            bool has_n = false;
            uint8_t rstrand = 0;
            if (valid) {
                pos = p.r_pos[i];
                flag = p.r_flag[i];
                o0 = p.cig_off[i];
                n_ops = p.cig_off[i + 1] - o0;
                int64_t ref_len;
                spl_read_extent(p.cigar + o0, n_ops, &ref_len, &has_n);
                const int64_t end64 = (int64_t)pos + spl_fetch_len(flag, ref_len) - 1;
                // the CIGAR walk runs in int32: the whole read must fit, not only its fetch window
                if ((int64_t)pos + ref_len > (int64_t)SPL_COORD_MAX || pos < 0) {
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
                do_pair<STRANDED>(p, lds, wbase, s, t, pos, p.cigar + o0, n_ops, has_n, rstrand);
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
                const uint32_t b_o0 = __shfl(o0, src);
                const uint32_t b_nops = __shfl(n_ops, src);
                const bool b_has_n = __shfl((int)has_n, src) != 0;
                const uint8_t b_rs = (uint8_t)__shfl((int)rstrand, src);
                for (int32_t s0 = b_s;; s0 += 64) {
                    const int32_t my = s0 + lane;
                    int32_t t = 0;
                    const bool in = my < n_sites && (t = p.site_pos[my]) <= b_end;
                    if (in) do_pair<STRANDED>(p, lds, wbase, my, t, b_pos, p.cigar + b_o0, b_nops, b_has_n, b_rs);
                    if (__ballot(in) != ~0ull) break; // rows are sorted: a lane out of range ends the read
                }
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

template <int NARR>
__device__ __forceinline__ void diff_add(const spl_count_params &p, int32_t *lds, int32_t wbase, int arr, int32_t row, int32_t v)
{
    const uint32_t loc = (uint32_t)(row - wbase);
    if (loc <= (uint32_t)SPL_WIN) atomicAdd(&lds[arr * (SPL_WIN + 1) + (int)loc], v);
    else atomicAdd(&p.diff[(int64_t)arr * p.diff_stride + row], v);
}

// All 64 lanes call this together.  Lanes that add `sign` to the same (array, row) and sit next to each other
// form a run; the first lane of each run adds sign * run-length once.  (Equal keys that are NOT adjacent simply
// make several runs: still correct, just more atomics -- that only happens for unsorted input.)
template <int NARR>
__device__ __forceinline__ void commit_run(const spl_count_params &p, int32_t *lds, int32_t wbase, bool valid, int arr,
                                           int32_t row, int32_t sign)
{
    const int lane = threadIdx.x & 63;
    const int32_t key = valid ? row : -1;
    const int32_t prev_key = __shfl_up(key, 1);
    const int prev_arr = __shfl_up(arr, 1);
    const bool head = valid && (lane == 0 || prev_key != key || prev_arr != arr);
    const unsigned long long heads = __ballot(head);
    const unsigned long long act = __ballot(valid);
    if (head) {
        const unsigned long long stop = (heads | ~act) & ~((2ull << lane) - 1ull); // later lanes that end my run
        const int len = stop ? (__ffsll((long long)stop) - 1 - lane) : (64 - lane);
        diff_add<NARR>(p, lds, wbase, arr, row, sign * len);
    }
}

// Is any junction end of the read that comes before (stop_op, stop_side) -- in lSite, rSite order per N op --
// a member of `part`?  Used to visit a rival exactly once.
__device__ __forceinline__ bool earlier_end_in(int32_t pos, const uint32_t *ops, uint32_t stop_op, int stop_side,
                                               const int32_t *part, uint32_t n_part)
{
    int32_t cur = pos;
    for (uint32_t k = 0; k <= stop_op; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if (!((SPL_PROG_MASK >> code) & 1u)) continue;
        const int32_t d = (int32_t)(op >> 4);
        cur += d;
        if (code != SPL_OP_N) continue;
        const int32_t l = cur - d - 1, r = cur - 1;
        if (k < stop_op) {
            if (spl_contains(part, n_part, l) || spl_contains(part, n_part, r)) return true;
        } else if (stop_side == 1) {
            if (spl_contains(part, n_part, l)) return true;
        }
    }
    return false;
}

template <bool STRANDED>
__device__ __forceinline__ void rivals_of_end(const spl_count_params &p, int32_t pos, uint32_t flag, const uint32_t *ops,
                                           uint32_t n_ops, int32_t end_fetch, uint32_t k_op, int side, int32_t x)
{
    const int32_t n_sites = p.n_sites;
    const int32_t r0 = first_site_at_or_after(p, x);
    for (int32_t row = r0; row < n_sites && p.site_pos[row] == x; ++row) {
        const uint4 m = p.site_meta[row];
        for (uint32_t e = 0; e < m.y; ++e) {
            const int32_t trow = p.part_site[m.x + e];
            if (trow < 0) continue;
            const int32_t t = p.site_pos[trow];
            if (t < pos || t > end_fetch) continue;
            const uint4 mt = p.site_meta[trow];
            if (mt.w == 0u) continue; // no competitors: compSplicing impossible
            const int32_t *part = p.part_pos + mt.x;
            const int32_t *comp = p.comp_pos + mt.z;
            // visit once: skip when an earlier junction end of this read also leads here, or an earlier row at x does
            if (earlier_end_in(pos, ops, k_op, side, part, mt.y)) continue;
            bool dup = false;
            for (int32_t row2 = r0; row2 < row && !dup; ++row2) {
                const uint4 m2 = p.site_meta[row2];
                for (uint32_t e2 = 0; e2 < m2.y; ++e2) dup |= (p.part_site[m2.x + e2] == trow);
            }
            for (uint32_t e2 = 0; e2 < e; ++e2) dup |= (p.part_site[m.x + e2] == trow);
            if (dup) continue;
            bool strand_ok = true;
            if (STRANDED) strand_ok = (p.site_strand[trow] == spl_read_strand(flag, p.stranded));
            const spl_pair full = spl_classify_pair(pos, ops, n_ops, t, part, mt.y, comp, mt.w, strand_ok);
            const spl_pair base = spl_classify_pair(pos, ops, n_ops, t, nullptr, 0u, nullptr, 0u, strand_ok);
            if (full.cls == base.cls) continue;
            // take back what the ranges counted for this row, apply the literal outcome
            if (base.cls == SPL_CLS_BETA1) atomicAdd(&p.beta1[trow], 0xffffffffu);
            else if (base.cls == SPL_CLS_ME) atomicAdd(&p.beta2s_reads[trow], 0xffffffffu);
            apply_pair_global(p, full, trow, pos, ops, n_ops, part, mt.y, mt.x);
        }
    }
}

template <bool STRANDED>
__device__ __forceinline__ void rivals_pass(const spl_count_params &p, int32_t pos, uint32_t flag, const uint32_t *ops,
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
        rivals_of_end<STRANDED>(p, pos, flag, ops, n_ops, end_fetch, k, 0, cur - d - 1);
        rivals_of_end<STRANDED>(p, pos, flag, ops, n_ops, end_fetch, k, 1, cur - 1);
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

} // namespace

// One read per lane, ops walked in a loop: the fallback for chunks whose CIGARs do not fit the LDS staging area
// (long-read data); same ranges, same commits, just latency-bound.
template <bool STRANDED>
__device__ __forceinline__ void ranges_chunk_serial(const spl_count_params &p, int32_t *lds, int32_t wbase, int64_t chunk_base)
{
    constexpr int NARR = STRANDED ? 4 : 2;
    const int tid = threadIdx.x;
    const int32_t n_sites = p.n_sites;
    {
        for (int it = 0; it < SPL_RPT; ++it) {
            // Control flow below is wave-uniform (every lane reaches every commit_run) so that lanes whose ranges
            // start or end on the same row -- the normal case for coordinate-sorted reads -- share ONE LDS atomic.
            const int64_t i = chunk_base + (int64_t)it * SPL_BLOCK + tid;
            bool alive = i < p.n_reads;
            int32_t pos = 0;
            uint32_t flag = 0, n_ops = 0;
            const uint32_t *ops = p.cigar;
            if (alive) {
                pos = p.r_pos[i];
                flag = p.r_flag[i];
                const uint32_t o0 = p.cig_off[i];
                n_ops = p.cig_off[i + 1] - o0;
                ops = p.cigar + o0;
                if (pos < 0) { atomicOr(p.err, SPL_DEV_ERR_RANGE); alive = false; }
                else if (flag & 4u) {
                    int64_t rl; bool hn;
                    spl_read_extent(ops, n_ops, &rl, &hn);
                    if ((int64_t)pos + rl > (int64_t)SPL_COORD_MAX) atomicOr(p.err, SPL_DEV_ERR_RANGE);
                    else unmapped_read<STRANDED>(p, pos, flag, ops, n_ops);
                    alive = false;
                }
            }
            int sidx = 0; // which strand pair of arrays this read writes
            if (STRANDED) sidx = (spl_read_strand(flag, p.stranded) == (uint8_t)'-') ? 1 : 0;

            // rows at position pos-1 are [prev_hi, idx); idx = first row >= pos
            int32_t prev_hi = 0, idx = 0;
            if (alive) {
                prev_hi = first_site_at_or_after(p, pos - 1);
                idx = prev_hi;
                while (idx < n_sites && p.site_pos[idx] < pos) ++idx;
            }
            int32_t c = pos;
            int64_t ref_len = 0;
            bool need_rivals = false;
            for (uint32_t k = 0; __any(alive && k < n_ops); ++k) {
                bool emit = false;
                int arr = 0;
                int32_t lo = 0, hi = 0;
                if (alive && k < n_ops) {
                    const uint32_t op = ops[k];
                    const uint32_t code = op & 15u;
                    const int32_t d = (int32_t)(op >> 4);
                    const bool is_n = (code == SPL_OP_N);
                    if (!((SPL_PROG_MASK >> code) & 1u)) {
                        // I, S, H, P: no progression (:463-464)
                    } else if (d == 0) { // covers nothing; a 0N still names a junction (lSite == rSite == c-1)
                        if (is_n) for (int32_t r = prev_hi; r < idx; ++r) need_rivals |= (p.site_flags[r] & SPL_SF_RIVALS) != 0;
                    } else if ((ref_len += d) + pos > (int64_t)SPL_COORD_MAX) {
                        alive = false; // flagged below; coordinates would wrap
                    } else {
                        const int32_t c2 = c + d;
                        // hi = first row with position >= c2-1: rows [idx, hi) have positions in [c, c2-2]
                        hi = idx;
                        if (d <= 512) {
                            int steps = 0;
                            while (hi < n_sites && p.site_pos[hi] < c2 - 1) {
                                ++hi;
                                if (++steps == 6) { hi = first_site_at_or_after(p, c2 - 1); break; }
                            }
                        } else {
                            hi = first_site_at_or_after(p, c2 - 1);
                        }
                        lo = idx;
                        emit = hi > lo && code != SPL_OP_D;
                        arr = (is_n ? (STRANDED ? 2 : 1) : 0) + sidx;
                        // rows at position c2-1 (the op's last base; rSite of an N op) are [hi, v2)
                        int32_t v2 = hi;
                        while (v2 < n_sites && p.site_pos[v2] == c2 - 1) ++v2;
                        if (is_n) { // junction rows: lSite = c-1 -> [prev_hi, idx), rSite = c2-1 -> [hi, v2)
                            for (int32_t r = prev_hi; r < idx; ++r) need_rivals |= (p.site_flags[r] & SPL_SF_RIVALS) != 0;
                            for (int32_t r = hi; r < v2; ++r) need_rivals |= (p.site_flags[r] & SPL_SF_RIVALS) != 0;
                        }
                        prev_hi = hi;
                        idx = v2;
                        c = c2;
                    }
                }
                if (__any(emit)) {
                    commit_run<NARR>(p, lds, wbase, emit, arr, lo, 1);
                    commit_run<NARR>(p, lds, wbase, emit, arr, hi, -1);
                }
            }
            if ((int64_t)pos + ref_len > (int64_t)SPL_COORD_MAX) { atomicOr(p.err, SPL_DEV_ERR_RANGE); alive = false; }
            if (alive && need_rivals)
                rivals_pass<STRANDED>(p, pos, flag, ops, n_ops, (int32_t)((int64_t)pos + (ref_len > 0 ? ref_len : 1) - 1));
        }
    }
}

// Record byte of one boundary slot (see spl_count_ranges_kernel).
#define SPL_K_LIVE 1u
#define SPL_K_KIND(k) (((k) >> 1) & 3u) // 0 read start, 1 aligned op, 2 N op, 3 D op
#define SPL_K_SIDX(k) (((k) >> 3) & 1u)
#define SPL_K_RIVAL 16u
#define SPL_K_NV(k) (((k) >> 5) & 3u)   // rows AT the boundary's last position (3 = three or more: recount)

// The range kernel proper.  A workgroup owns SPL_CHUNK consecutive reads and runs four passes over LDS:
//   P0  stage the chunk's raw CIGAR ops (one coalesced sweep of the op array);
//   P1  one read per lane: walk its ops (LDS latency only) and lay down one *boundary record* per reference-
//       consuming op, preceded by one for the read start: slot = coordinate just past the op, kind, strand;
//   P2  FLAT over slots, perfectly balanced and free of cross-lane dependence: row lookup of each boundary
//       (first row at or after coordinate-1, and how many rows sit exactly there);
//   P3  FLAT over slots: an op's row range is [start boundary's rows end, end boundary's rows begin); commit
//       +1/-1 (adjacent lanes hitting one row share an atomic); N ops look at the rival flag of their junction rows;
//   P4  one read per lane: reads with a flagged junction end run the literal rival pass.
template <bool STRANDED>
__global__ __launch_bounds__(SPL_BLOCK) void spl_count_ranges_kernel(const spl_count_params p)
{
    constexpr int NARR = STRANDED ? 4 : 2; // {beta1, ME} x {read strand +, -}
    constexpr int SLOTS = SPL_CHUNK + SPL_OPS_CAP;
    __shared__ int32_t lds[NARR * (SPL_WIN + 1)];
    __shared__ uint32_t s_op[SPL_OPS_CAP];
    __shared__ int32_t s_b[SLOTS];
    __shared__ uint8_t s_kind[SLOTS];
    __shared__ int32_t s_wbase;

    const int tid = threadIdx.x;
    const uint32_t chunk = my_chunk();
    const bool live = chunk < p.n_chunks;
    const int64_t chunk_base = (int64_t)chunk * SPL_CHUNK;
    const int32_t n_sites = p.n_sites;

    uint32_t ob = 0, n_co = 0;
    int n_rd = 0;
    if (live) {
        const int64_t chunk_end = (chunk_base + SPL_CHUNK < p.n_reads) ? chunk_base + SPL_CHUNK : p.n_reads;
        n_rd = (int)(chunk_end - chunk_base);
        ob = p.cig_off[chunk_base];
        n_co = p.cig_off[chunk_end] - ob;
    }
    const bool staged = live && n_co <= (uint32_t)SPL_OPS_CAP;
    const int n_slots = staged ? n_rd + (int)n_co : 0;

    // ---- P0 ------------------------------------------------------------------------------------------------
    for (int j = tid; j < NARR * (SPL_WIN + 1); j += SPL_BLOCK) lds[j] = 0;
    if (staged) {
        for (uint32_t j = tid; j < n_co; j += SPL_BLOCK) s_op[j] = p.cigar[ob + j];
        for (int j = tid; j < n_slots; j += SPL_BLOCK) s_kind[j] = 0;
    }
    if (tid == 0) s_wbase = live ? first_site_at_or_after(p, p.r_pos[chunk_base] - 1) : 0;
    __syncthreads();
    const int32_t wbase = s_wbase;

    if (live && !staged) ranges_chunk_serial<STRANDED>(p, lds, wbase, chunk_base);

    // ---- P1 ------------------------------------------------------------------------------------------------
    int32_t r_pos[SPL_RPT];
    uint32_t r_flag[SPL_RPT], r_o0[SPL_RPT], r_nops[SPL_RPT];
    int r_base[SPL_RPT], r_nrec[SPL_RPT];
    int64_t r_len[SPL_RPT];
#pragma unroll
    for (int q = 0; q < SPL_RPT; ++q) {
        const int r = q * SPL_BLOCK + tid;
        r_nrec[q] = 0; r_pos[q] = 0; r_flag[q] = 0; r_o0[q] = 0; r_nops[q] = 0; r_base[q] = 0; r_len[q] = 0;
        if (staged && r < n_rd) {
            const int64_t i = chunk_base + r;
            r_pos[q] = p.r_pos[i];
            r_flag[q] = p.r_flag[i];
            r_o0[q] = p.cig_off[i];
            r_nops[q] = p.cig_off[i + 1] - r_o0[q];
        }
    }
#pragma unroll
    for (int q = 0; q < SPL_RPT; ++q) {
        const int r = q * SPL_BLOCK + tid;
        if (!(staged && r < n_rd)) continue;
        const int32_t pos = r_pos[q];
        const uint32_t flag = r_flag[q], n_ops = r_nops[q];
        const uint32_t lo0 = r_o0[q] - ob; // first raw op of this read in s_op
        if (pos < 0) { atomicOr(p.err, SPL_DEV_ERR_RANGE); continue; }
        if (flag & 4u) {
            int64_t rl; bool hn;
            spl_read_extent(p.cigar + r_o0[q], n_ops, &rl, &hn);
            if ((int64_t)pos + rl > (int64_t)SPL_COORD_MAX) atomicOr(p.err, SPL_DEV_ERR_RANGE);
            else unmapped_read<STRANDED>(p, pos, flag, p.cigar + r_o0[q], n_ops);
            continue;
        }
        uint32_t sbit = 0;
        if (STRANDED) sbit = (spl_read_strand(flag, p.stranded) == (uint8_t)'-') ? 8u : 0u;
        const int base = (int)lo0 + r;
        int32_t cur = pos;
        int64_t ref_len = 0;
        int nrec = 0;
        bool ok = true;
        for (uint32_t k = 0; k < n_ops; ++k) {
            const uint32_t op = s_op[lo0 + k];
            const uint32_t code = op & 15u;
            if (!((SPL_PROG_MASK >> code) & 1u)) continue;
            const int32_t d = (int32_t)(op >> 4);
            ref_len += d;
            if ((int64_t)pos + ref_len > (int64_t)SPL_COORD_MAX) { ok = false; break; }
            cur += d;
            ++nrec;
            const uint32_t kind = (code == SPL_OP_N) ? 2u : (code == SPL_OP_D ? 3u : 1u);
            s_b[base + nrec] = cur;
            s_kind[base + nrec] = (uint8_t)(SPL_K_LIVE | (kind << 1) | sbit);
        }
        if (!ok) {
            atomicOr(p.err, SPL_DEV_ERR_RANGE);
            for (int j = 1; j <= nrec; ++j) s_kind[base + j] = 0;
            continue;
        }
        s_b[base] = pos;
        s_kind[base] = (uint8_t)SPL_K_LIVE; // kind 0: read start
        r_base[q] = base;
        r_nrec[q] = nrec;
        r_len[q] = ref_len;
    }
    __syncthreads();

    // ---- P2: boundary -> rows ----------------------------------------------------------------------------
    for (int g = tid; g < n_slots; g += SPL_BLOCK) {
        const uint32_t kd = s_kind[g];
        if (!(kd & SPL_K_LIVE)) continue;
        const int32_t x = s_b[g] - 1;
        const int32_t u = first_site_at_or_after(p, x);
        uint32_t nv = 0;
        while (nv < 3u && u + (int32_t)nv < n_sites && p.site_pos[u + nv] == x) ++nv;
        s_b[g] = u;
        s_kind[g] = (uint8_t)(kd | (nv << 5));
    }
    __syncthreads();

    // ---- P3: ranges -> difference arrays ------------------------------------------------------------------
    for (int g0 = 0; g0 < n_slots; g0 += SPL_BLOCK) { // wave-uniform trip count: every lane reaches commit_run
        const int g = g0 + tid;
        bool emit = false;
        int arr = 0;
        int32_t lo = 0, hi = 0;
        if (g < n_slots) {
            const uint32_t kd = s_kind[g];
            const uint32_t kind = SPL_K_KIND(kd);
            if ((kd & SPL_K_LIVE) && kind != 0u) {
                const uint32_t ks = s_kind[g - 1];
                const int32_t us = s_b[g - 1];
                uint32_t nvs = SPL_K_NV(ks), nve = SPL_K_NV(kd);
                hi = s_b[g];
                if (nvs == 3u) { const int32_t x = p.site_pos[us]; while (us + (int32_t)nvs < n_sites && p.site_pos[us + nvs] == x) ++nvs; }
                lo = us + (int32_t)nvs;
                emit = hi > lo && kind != 3u;
                arr = (kind == 2u ? (STRANDED ? 2 : 1) : 0) + (int)SPL_K_SIDX(kd);
                if (kind == 2u) { // junction rows: lSite -> [us, lo), rSite -> [hi, hi + nve)
                    if (nve == 3u) { const int32_t x = p.site_pos[hi]; while (hi + (int32_t)nve < n_sites && p.site_pos[hi + nve] == x) ++nve; }
                    bool rival = false;
                    for (int32_t r = us; r < lo; ++r) rival |= (p.site_flags[r] & SPL_SF_RIVALS) != 0;
                    for (int32_t r = hi; r < hi + (int32_t)nve; ++r) rival |= (p.site_flags[r] & SPL_SF_RIVALS) != 0;
                    if (rival) s_kind[g] = (uint8_t)(kd | SPL_K_RIVAL);
                }
            }
        }
        if (__any(emit)) {
            commit_run<NARR>(p, lds, wbase, emit, arr, lo, 1);
            commit_run<NARR>(p, lds, wbase, emit, arr, hi, -1);
        }
    }
    __syncthreads();

    // ---- P4: rivals ----------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < SPL_RPT; ++q) {
        bool need = false;
        for (int j = 1; j <= r_nrec[q]; ++j) need |= (s_kind[r_base[q] + j] & SPL_K_RIVAL) != 0;
        if (need)
            rivals_pass<STRANDED>(p, r_pos[q], r_flag[q], p.cigar + r_o0[q], r_nops[q],
                                  (int32_t)((int64_t)r_pos[q] + (r_len[q] > 0 ? r_len[q] : 1) - 1));
    }

    for (int j = tid; j < NARR * (SPL_WIN + 1); j += SPL_BLOCK) {
        const int32_t v = lds[j];
        if (v) {
            const int arr = j / (SPL_WIN + 1), loc = j - arr * (SPL_WIN + 1);
            atomicAdd(&p.diff[(int64_t)arr * p.diff_stride + wbase + loc], v);
        }
    }
}

// =========================================================================================================
// Difference arrays -> counters: two tiny launches (block sums, then offset + local inclusive scan).
// =========================================================================================================
__global__ __launch_bounds__(256) void spl_scan_sums_kernel(const spl_scan_params p)
{
    __shared__ int32_t red[4];
    const int arr = blockIdx.y;
    const int32_t base = blockIdx.x * SPL_SCAN_BLOCK;
    const int32_t *d = p.diff + (int64_t)arr * p.diff_stride;
    int32_t acc = 0;
    for (int j = threadIdx.x; j < SPL_SCAN_BLOCK; j += 256) {
        const int32_t r = base + j;
        if (r < p.n_sites) acc += d[r];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.block_sums[arr * p.n_blocks + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void spl_scan_apply_kernel(const spl_scan_params p)
{
    __shared__ int32_t red[4][4];   // [array][wave]
    __shared__ int32_t wave_tot[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t base = blockIdx.x * SPL_SCAN_BLOCK;
    int32_t run[4] = {0, 0, 0, 0};
    // offset of this block = sum of the sums of all blocks before it
    for (int a = 0; a < p.n_arrays; ++a) {
        int32_t acc = 0;
        for (int j = tid; j < (int)blockIdx.x; j += 256) acc += p.block_sums[a * p.n_blocks + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) red[a][wave] = acc;
    }
    // each thread owns 4 consecutive rows
    int32_t v[4][4];
    const int32_t r0 = base + tid * 4;
    for (int a = 0; a < p.n_arrays; ++a) {
        const int32_t *d = p.diff + (int64_t)a * p.diff_stride;
        int32_t s = 0;
        for (int q = 0; q < 4; ++q) {
            const int32_t r = r0 + q;
            s += (r < p.n_sites) ? d[r] : 0;
            v[a][q] = s; // inclusive within the thread
        }
        // wave-inclusive scan of thread totals
        int32_t incl = s;
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
    for (int q = 0; q < 4; ++q) {
        const int32_t r = r0 + q;
        if (r >= p.n_sites) break;
        int32_t b1, me;
        if (p.n_arrays == 2) { b1 = run[0] + v[0][q]; me = run[1] + v[1][q]; }
        else {
            const uint8_t f = p.site_flags[r];
            if (f & SPL_SF_PLUS) { b1 = run[0] + v[0][q]; me = run[2] + v[2][q]; }
            else if (f & SPL_SF_MINUS) { b1 = run[1] + v[1][q]; me = run[3] + v[3][q]; }
            else { b1 = 0; me = 0; } // a row without strand matches no read in a stranded run (:406)
        }
        if (b1) p.beta1[r] += (uint32_t)b1;
        if (me) p.beta2s_reads[r] += (uint32_t)me;
    }
}

// =========================================================================================================
// findBeta2Counts + calculateSSE, one site per lane.  IEEE binary64, compiled with -ffp-contract=off: every
// operation below is one correctly rounded operation in the reference's order, so the doubles are the ones
// CPython produces (int/int true division included for |values| < 2^53).
// =========================================================================================================
__global__ __launch_bounds__(256) void spl_sse_kernel(const spl_sse_params p)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.n_sites) return;
    const int64_t t = p.site_pos[s];
    int64_t b2simple = p.beta2s_reads[s];
    int64_t cryptic = 0;
    double weighted = 0.0;
    const int64_t total_alpha = p.alpha[s];
    const uint32_t e0 = p.part_off[s], e1 = p.part_off[s + 1];
    for (uint32_t e = e0; e < e1; ++e) { // for pSite in Partners (:590)
        const int32_t ps = p.part_site[e];
        if (ps < 0) continue;
        const int64_t ppos = p.site_pos[ps];
        int64_t doubles = p.dbl[e];
        bool have_key = doubles != 0;
        const uint32_t f1 = p.part_off[ps + 1];
        for (uint32_t f = p.part_off[ps]; f < f1; ++f) { // pSite.getPartnerCounts().items() (:592)
            const int64_t cpos = p.part_pos[f];
            if ((ppos > t && cpos < t) || (ppos < t && cpos > t)) { // junction (pSite, c) flanks t (:594-599)
                const int64_t cnt = p.edge_cnt[f];
                b2simple += cnt;
                doubles += cnt;
                have_key = true;
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
    p.beta2_simple[s] = b2simple;
    p.beta2_cryptic[s] = cryptic;
    p.beta2_weighted[s] = weighted;
    const int64_t betas_int = (int64_t)p.beta1[s] + b2simple; // :631
    double value = 0.0;
    if (p.cryptic) {
        const double betas = (double)betas_int + weighted;     // :635
        const double denom = (double)total_alpha + betas;      // :637
        if (denom > 0.0) value = (double)total_alpha / denom;
    } else {
        const int64_t denom = total_alpha + betas_int;
        if ((double)denom > 0.0) value = (double)total_alpha / (double)denom;
    }
    p.sse[s] = value;
}

// ---- launchers (called from spl_capi.cpp through spl_device.h) ------------------------------------------

extern "C" int spl_dev_launch_count(const spl_count_params *p, int variant, void *stream, int *grid_out, int *lds_out)
{
    *grid_out = 0;
    *lds_out = 0;
    if (p->n_reads <= 0 || p->n_sites <= 0) return 0;
    // grid = 8 * ceil(n_chunks / 8) so that every XCD slice has the same number of slots
    const uint32_t grid = ((p->n_chunks + 7u) / 8u) * 8u;
    *grid_out = (int)grid;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 1) {
        *lds_out = 2 * SPL_WIN * 4 + 4;
        if (p->stranded) hipLaunchKernelGGL(spl_count_pairs_kernel<true>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
        else hipLaunchKernelGGL(spl_count_pairs_kernel<false>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
    } else {
        *lds_out = (p->stranded ? 4 : 2) * (SPL_WIN + 1) * 4 + 4 * SPL_OPS_CAP + 5 * (SPL_CHUNK + SPL_OPS_CAP) + 4;
        if (p->stranded) hipLaunchKernelGGL(spl_count_ranges_kernel<true>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
        else hipLaunchKernelGGL(spl_count_ranges_kernel<false>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
    }
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_scan(const spl_scan_params *p, void *stream)
{
    if (p->n_sites <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(spl_scan_sums_kernel, dim3(p->n_blocks, p->n_arrays), dim3(256), 0, st, *p);
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
