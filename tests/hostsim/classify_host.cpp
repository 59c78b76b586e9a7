// Test-only host build of the kernel's arithmetic core (spliser_amd/csrc/spl_classify.h).
//
// The build container has no GPU, so before spending GPU minutes the per-(read, site) decision used by
// spl_count_kernel is compiled here with g++ and driven read by read -- the same read-centric order the
// kernel uses (candidate sites = rows with POS <= pos <= POS+L-1) -- and compared with the site-centric
// oracle.  Not part of the product: nothing under spliser_amd/ builds, loads or calls this file.
#include <cstdint>
#include <cstring>

#define SPL_HD inline
#include "../../spliser_amd/csrc/spl_classify.h"

extern "C" int sim_count(int64_t n_sites, const int32_t *site_pos, const uint8_t *site_strand, const uint32_t *part_off,
                         const int32_t *part_pos, const uint32_t *comp_off, const int32_t *comp_pos, int64_t n_reads,
                         const int32_t *r_pos, const uint16_t *r_flag, const uint32_t *cig_off, const uint32_t *cigar,
                         int stranded, int combine_mode, uint32_t *beta1, uint32_t *b2s, uint32_t *dbl)
{
    memset(beta1, 0, sizeof(uint32_t) * n_sites);
    memset(b2s, 0, sizeof(uint32_t) * n_sites);
    if (n_sites) memset(dbl, 0, sizeof(uint32_t) * part_off[n_sites]);
    for (int64_t i = 0; i < n_reads; ++i) {
        const int32_t pos = r_pos[i];
        const uint32_t flag = r_flag[i];
        const uint32_t *ops = cigar + cig_off[i];
        const uint32_t n_ops = cig_off[i + 1] - cig_off[i];
        int64_t ref_len;
        bool has_n;
        spl_read_extent(ops, n_ops, &ref_len, &has_n);
        const int64_t end = (int64_t)pos + spl_fetch_len(flag, ref_len) - 1;
        const uint8_t rs = stranded ? spl_read_strand(flag, stranded) : 0;
        int64_t s = 0;
        while (s < n_sites && site_pos[s] < pos) ++s; // plain scan: this harness tests arithmetic, not indexing
        for (; s < n_sites && site_pos[s] <= end; ++s) {
            const bool strand_ok = !stranded || site_strand[s] == rs;
            const int32_t *part = nullptr, *comp = nullptr;
            uint32_t n_part = 0, n_comp = 0;
            if (has_n && comp_off[s + 1] != comp_off[s]) {
                part = part_pos + part_off[s]; n_part = part_off[s + 1] - part_off[s];
                comp = comp_pos + comp_off[s]; n_comp = comp_off[s + 1] - comp_off[s];
            }
            const spl_pair r = spl_classify_pair(pos, ops, n_ops, site_pos[s], part, n_part, comp, n_comp, strand_ok);
            switch (r.cls) {
            case SPL_CLS_BETA1: beta1[s]++; break;
            case SPL_CLS_ME: b2s[s]++; break;
            case SPL_CLS_FLANK: if (combine_mode) b2s[s]++; break;
            case SPL_CLS_B1TYPE: b2s[s]++; /* fall through */
            case SPL_CLS_ALPHA_COMP:
                for (uint32_t e = 0; e < n_part; ++e) {
                    if (r.cls == SPL_CLS_ALPHA_COMP && r.has_partner_used && part[e] == r.partner_used) continue;
                    if (spl_read_splices_at(pos, ops, n_ops, part[e])) dbl[part_off[s] + e]++;
                }
                break;
            default: break;
            }
        }
    }
    return 0;
}
