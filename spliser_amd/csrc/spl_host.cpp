// Host-side helpers of the C ABI that need no GPU (include/spliser.h).
#include <cstdint>
#include <cstring>

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

// ---- .SpliSER.tsv rows (outputBedFile, SpliSER_v0_1_8.py:641-664) -----------------------------------------------
// One chromosome's rows appended to an open file: the same bytes spliser_amd/tsv.py formats -- "{0:.3f}" / "{0:.5f}" are
// correctly rounded decimal conversions in CPython and in glibc's printf alike, str(dict) / str(list) of ints are
// "{a: b, c: d}" / "[a, b]".  Strings come as one blob + offsets.
#include <algorithm>
#include <atomic>
#include <cinttypes>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

extern "C" int spl_tsv_append(const char *path, const char *chrom, int64_t n_sites, const int64_t *pos, const char *strand_blob,
                              const uint32_t *strand_off, const char *gene_blob, const uint32_t *gene_off, const double *sse,
                              const int64_t *alpha, const uint32_t *beta1, const int64_t *beta2_simple, int cryptic,
                              const int64_t *beta2_cryptic, const double *beta2_weighted, const uint32_t *part_off,
                              const int64_t *part_pos, const int64_t *edge_cnt, const uint32_t *comp_off, const int64_t *comp_pos)
{
    if (!path || !chrom || n_sites < 0) return spl_set_error(SPL_ERR_ARG, "spl_tsv_append: bad argument");
    if (n_sites && (!pos || !strand_blob || !strand_off || !gene_blob || !gene_off || !sse || !alpha || !beta1 || !beta2_simple ||
                    !part_off || !comp_off || (cryptic && (!beta2_cryptic || !beta2_weighted))))
        return spl_set_error(SPL_ERR_ARG, "spl_tsv_append: null array");
    FILE *f = fopen(path, "ab");
    if (!f) return spl_set_error(SPL_ERR_IO, "cannot open %s for appending", path);
    const size_t chrom_len = strlen(chrom);
    // rows [a, b) as text
    auto format = [&](int64_t a, int64_t b, std::string &out) {
        char num[64];
        out.reserve((size_t)(b - a) * 96);
        for (int64_t i = a; i < b; ++i) {
            out.append(chrom, chrom_len);
            out.push_back('\t');
            out.append(num, (size_t)snprintf(num, sizeof num, "%" PRId64, pos[i]));
            out.push_back('\t');
            out.append(strand_blob + strand_off[i], strand_off[i + 1] - strand_off[i]);
            out.push_back('\t');
            out.append(gene_blob + gene_off[i], gene_off[i + 1] - gene_off[i]);
            out.push_back('\t');
            out.append(num, (size_t)snprintf(num, sizeof num, "%.3f", sse[i]));
            out.append(num, (size_t)snprintf(num, sizeof num, "\t%" PRId64 "\t%u\t%" PRId64 "\t", alpha[i], beta1[i], beta2_simple[i]));
            if (cryptic) out.append(num, (size_t)snprintf(num, sizeof num, "%" PRId64 "\t%.5f", beta2_cryptic[i], beta2_weighted[i]));
            else out.append("NA\tNA");
            out.append("\t{");
            for (uint32_t e = part_off[i]; e < part_off[i + 1]; ++e) {
                if (e != part_off[i]) out.append(", ");
                out.append(num, (size_t)snprintf(num, sizeof num, "%" PRId64 ": %" PRId64, part_pos[e], edge_cnt[e]));
            }
            out.append("}\t[");
            for (uint32_t e = comp_off[i]; e < comp_off[i + 1]; ++e) {
                if (e != comp_off[i]) out.append(", ");
                out.append(num, (size_t)snprintf(num, sizeof num, "%" PRId64, comp_pos[e]));
            }
            out.append("]\n");
        }
    };
    // slices of rows formatted on a few threads (snprintf of a quarter of a million rows is 80 ms on one), written in order
    const int64_t SLICE = 8192;
    const size_t n_slices = (size_t)((n_sites + SLICE - 1) / SLICE);
    std::vector<std::string> text(n_slices);
    {
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_slices) break;
                format((int64_t)k * SLICE, std::min<int64_t>(n_sites, (int64_t)(k + 1) * SLICE), text[k]);
            }
        };
        int nt = (int)std::thread::hardware_concurrency();
        nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
        nt = (int)std::min<size_t>((size_t)nt, n_slices);
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    bool ok = true;
    for (size_t k = 0; k < n_slices && ok; ++k) ok = fwrite(text[k].data(), 1, text[k].size(), f) == text[k].size();
    if (fclose(f) != 0) ok = false;
    return ok ? SPL_OK : spl_set_error(SPL_ERR_IO, "write error on %s", path);
}
