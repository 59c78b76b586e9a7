// spl_kernels.hip -- CDNA4 (gfx950, wave64) kernels of the SpliSER `process` hot path.
//
//   spl_count_kernel  read-centric restatement of the checkBam loop (SpliSER_v0_1_8.py:408-559, called per
//                     site from processSites :686-688): every read finds the splice sites it would have been
//                     fetched for and classifies itself against each of them.
//   spl_sse_kernel    findBeta2Counts + calculateSSE (SpliSER_v0_1_8.py:581-639), one site per lane.
//
// Integer / indexing work: no MFMA.  The roofline that bounds spl_count_kernel is HBM.
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

// One counter increment: LDS-privatised when the site falls in this workgroup's window, global otherwise.
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
        const uint4 m = p.site_meta[s]; // {part_off, n_part, comp_off, n_comp}
        part_off = m.x;
        if (m.w != 0u) { // without competitors compSplicing can never be set (:494-501): lists not needed
            n_part = m.y;
            n_comp = m.w;
            part = p.part_pos + m.x;
            comp = p.comp_pos + m.z;
        }
    }
    const spl_pair r = spl_classify_pair(pos, ops, n_ops, t, part, n_part, comp, n_comp, strand_ok);
    switch (r.cls) {
    case SPL_CLS_BETA1:
        bump(lds, p.beta1, s, wbase);
        break;
    case SPL_CLS_ME:
        bump(lds + SPL_WIN, p.beta2s_reads, s, wbase);
        break;
    case SPL_CLS_FLANK:
        if (p.combine_mode) bump(lds + SPL_WIN, p.beta2s_reads, s, wbase);
        break;
    case SPL_CLS_B1TYPE:
        bump(lds + SPL_WIN, p.beta2s_reads, s, wbase);
        [[fallthrough]];
    case SPL_CLS_ALPHA_COMP:
        // PartnerBeta2DoubleCounts (:519-527, :544-551): rare, straight to HBM.
        for (uint32_t e = 0; e < n_part; ++e) {
            const int32_t pp = part[e];
            if (r.cls == SPL_CLS_ALPHA_COMP && r.has_partner_used && pp == r.partner_used) continue;
            if (spl_read_splices_at(pos, ops, n_ops, pp)) atomicAdd(&p.dbl[part_off + e], 1u);
        }
        break;
    default:
        break;
    }
}

} // namespace

template <bool STRANDED>
__global__ __launch_bounds__(SPL_BLOCK) void spl_count_kernel(const spl_count_params p)
{
    __shared__ uint32_t lds[2 * SPL_WIN]; // [0,WIN): beta1   [WIN,2WIN): beta2Simple (read-derived)
    __shared__ int32_t s_wbase;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // XCD-aware chunk order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD one
    // contiguous eighth of the (coordinate-sorted) reads -- its L2 then sees one moving site window.
    const uint32_t nblk = gridDim.x;
    const uint32_t per = (nblk + 7u) >> 3;
    uint32_t chunk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
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
                    if (__ballot(in) != ~0ull) break; // sites are sorted: a lane out of range ends the read
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

template __global__ void spl_count_kernel<false>(const spl_count_params);
template __global__ void spl_count_kernel<true>(const spl_count_params);

// findBeta2Counts + calculateSSE, one site per lane.  IEEE binary64, compiled with -ffp-contract=off:
// every operation below is one correctly rounded operation in the reference's order, so the doubles
// are the ones CPython produces (int/int true division included for |values| < 2^53).
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

extern "C" int spl_dev_launch_count(const spl_count_params *p, void *stream, int *grid_out)
{
    if (p->n_reads <= 0 || p->n_sites <= 0) { *grid_out = 0; return 0; }
    const uint32_t n_chunks = p->n_chunks;
    // grid = 8 * ceil(n_chunks / 8) so that every XCD slice has the same number of slots
    const uint32_t grid = ((n_chunks + 7u) / 8u) * 8u;
    *grid_out = (int)grid;
    hipStream_t st = (hipStream_t)stream;
    if (p->stranded) hipLaunchKernelGGL(spl_count_kernel<true>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
    else hipLaunchKernelGGL(spl_count_kernel<false>, dim3(grid), dim3(SPL_BLOCK), 0, st, *p);
    return (int)hipGetLastError();
}

extern "C" int spl_dev_launch_sse(const spl_sse_params *p, void *stream)
{
    if (p->n_sites <= 0) return 0;
    const uint32_t grid = (uint32_t)((p->n_sites + 255) / 256);
    hipLaunchKernelGGL(spl_sse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
