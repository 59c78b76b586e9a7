// Host-side helpers of the C ABI that need no GPU (include/spliser.h).
#include <cstdint>

#include "../../include/spliser.h"
#include "spl_error.h"

namespace {
inline int64_t floor_div2(int64_t v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); } // Python's v // 2
} // namespace

// binary_gene_search (SpliSER_v0_1_8.py:118-173) for a batch of positions, probe for probe: overlapping genes make the
// list only partially ordered, so the outcome depends on the exact probe sequence -- the lo/hi trackers, the idx == 1
// special case, the stop when an index repeats and the last-ditch sweep over offsets -3..2 that re-bases itself on every
// hit and never looks at the final list element are all kept.  Strand bytes: '+', '-' or 0 (anything else).
extern "C" int spl_gene_search(const int64_t *left, const int64_t *right, const uint8_t *gene_strand, int64_t n_genes,
                               const int64_t *q_pos, const uint8_t *q_strand, int64_t n_queries, int is_stranded, int32_t *out)
{
    if (n_queries < 0 || n_genes < 0) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: negative count");
    if (n_queries && (!q_pos || !q_strand || !out)) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: null query arrays");
    if (n_genes && (!left || !right || !gene_strand)) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: null gene arrays");
    if (n_genes > 0x7fffffff) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: too many genes");
    const int64_t n = n_genes;
    for (int64_t q = 0; q < n_queries; ++q) {
        if (n == 0) { out[q] = -1; continue; }
        const int64_t pos = q_pos[q];
        const uint8_t qs = q_strand[q];
        const bool free = !is_stranded || (qs != '+' && qs != '-');
        auto hit = [&](int64_t g) { return left[g] <= pos && pos <= right[g] && (free || qs == gene_strand[g]); };
        int64_t idx = n / 2, hi = n, lo = 0, last = -1, nxt = n / 2;
        int64_t found_at = -1;
        for (;;) {
            if (hit(idx)) { found_at = idx; break; }
            if (pos >= right[idx]) { nxt = idx + floor_div2(hi - idx); lo = idx; }
            else if (pos <= left[idx]) { nxt = idx - floor_div2(idx - lo); hi = idx; if (idx == 1) nxt = 0; }
            if (idx == last) break;
            last = idx;
            idx = nxt;
        }
        if (found_at >= 0) { out[q] = (int32_t)found_at; continue; }
        bool found = false;
        for (int off = -3; off < 3; ++off) {
            const int64_t j = idx + off;
            if (j >= 0 && j < n - 1 && left[j] <= pos && pos <= right[j] && (free || qs == gene_strand[j])) { found = true; idx = j; }
        }
        out[q] = found ? (int32_t)idx : -1;
    }
    return SPL_OK;
}
