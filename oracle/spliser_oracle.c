/*
 * spliser_oracle.c -- CPU restatement of SpliSER v0.1.8's `process` hot path.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle: tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py are the only callers.  The product (spliser_amd/, libspliser_hip.so) never links, loads
 * or calls it, and has no CPU fallback of its own.
 *
 * It follows the reference function by function, SITE by SITE like the reference does (the GPU path
 * is read-centric, so the two are independent formulations of the same contract):
 *
 *   orc_check_bam   = the checkBam loop of processSites     SpliSER_v0_1_8.py:686-688, :408-559
 *   orc_check_strand= check_strand                          SpliSER_v0_1_8.py:374-406
 *   orc_beta2_sse   = findBeta2Counts + calculateSSE        SpliSER_v0_1_8.py:581-639 (+ :562-579)
 *
 * Pinning: the reference ships no tests or fixtures, so this restatement is
 * pinned against outputs of the reference itself, executed in the build container by
 * tests/golden/make_golden.py (fixtures under tests/golden/, checked by tests/test_oracle_golden.py).
 * Two boundaries of that execution are stand-ins because the image has neither samtools nor HTSeq:
 * the region query (`samtools view chr:t-(t+1)`, restated in fetch_site() below from htslib's
 * documented iterator contract) and the GFF reader.  Parity is therefore pinned for everything the
 * reference's own Python computes and UNPINNED at those two third-party boundaries.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * read fetch: what `samtools view BAM chr:t-(t+1)` hands to checkBam (SpliSER_v0_1_8.py:422-435)
 * ---------------------------------------------------------------------------------------------- */

/* Threads for the site loop of orc_check_bam and the index build (default 1: the reference is single-threaded,
 * SpliSER_v0_1_8.py:681-692).  Only the all-cores CPU baseline of bench.py asks for more. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

typedef struct {
    int64_t n;
    int64_t *order;   /* read indices sorted by pos (stable)                 */
    int64_t *pos0;    /* 0-based start, in sorted order                       */
    int64_t *pmaxend; /* running maximum of endpos0 over sorted prefix        */
    int64_t *endpos0; /* htslib bam_endpos per read (file order)              */
} fetch_index;

static const int64_t *g_sort_key;
static int cmp_by_key(const void *a, const void *b)
{
    int64_t ia = *(const int64_t *)a, ib = *(const int64_t *)b;
    if (g_sort_key[ia] != g_sort_key[ib]) return g_sort_key[ia] < g_sort_key[ib] ? -1 : 1;
    return ia < ib ? -1 : (ia > ib);
}

/* htslib: rlen = 0 for unmapped (flag 0x4) records, else sum of M/D/N/=/X; endpos = pos + (rlen?rlen:1) */
static int64_t bam_endpos0(int64_t pos0, unsigned flag, const uint32_t *ops, uint32_t n_ops)
{
    int64_t rlen = 0;
    if (!(flag & 4u)) {
        for (uint32_t k = 0; k < n_ops; ++k) {
            unsigned code = ops[k] & 15u;
            if (code == 0 || code == 2 || code == 3 || code == 7 || code == 8) rlen += (int64_t)(ops[k] >> 4);
        }
    }
    if (rlen == 0) rlen = 1;
    return pos0 + rlen;
}

static int fetch_index_build(fetch_index *fi, int64_t n, const int32_t *r_pos, const uint16_t *r_flag,
                             const uint32_t *cig_off, const uint32_t *cigar)
{
    memset(fi, 0, sizeof(*fi));
    fi->n = n;
    if (n == 0) return 0;
    fi->order = (int64_t *)malloc(sizeof(int64_t) * n);
    fi->pos0 = (int64_t *)malloc(sizeof(int64_t) * n);
    fi->pmaxend = (int64_t *)malloc(sizeof(int64_t) * n);
    fi->endpos0 = (int64_t *)malloc(sizeof(int64_t) * n);
    int64_t *key = (int64_t *)malloc(sizeof(int64_t) * n);
    if (!fi->order || !fi->pos0 || !fi->pmaxend || !fi->endpos0 || !key) return -1;
    int sorted = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1) reduction(&& : sorted)
#endif
    for (int64_t i = 0; i < n; ++i) {
        key[i] = (int64_t)r_pos[i] - 1;
        fi->order[i] = i;
        fi->endpos0[i] = bam_endpos0(key[i], r_flag[i], cigar + cig_off[i], cig_off[i + 1] - cig_off[i]);
        if (i && (int64_t)r_pos[i] < (int64_t)r_pos[i - 1]) sorted = 0;
    }
    if (!sorted) {
        g_sort_key = key;
        qsort(fi->order, (size_t)n, sizeof(int64_t), cmp_by_key);
    }
    int64_t run = INT64_MIN;
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = fi->order[i];
        fi->pos0[i] = key[r];
        if (fi->endpos0[r] > run) run = fi->endpos0[r];
        fi->pmaxend[i] = run;
    }
    free(key);
    return 0;
}

static void fetch_index_free(fetch_index *fi)
{
    free(fi->order); free(fi->pos0); free(fi->pmaxend); free(fi->endpos0);
}

/* number of sorted reads with pos0 < bound */
static int64_t count_pos_below(const fetch_index *fi, int64_t bound)
{
    int64_t lo = 0, hi = fi->n;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (fi->pos0[mid] < bound) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ------------------------------------------------------------------------------------------------
 * check_strand  (SpliSER_v0_1_8.py:374-406).  stranded_type: 1 = "fr", 2 = "rf".
 * ---------------------------------------------------------------------------------------------- */
int orc_check_strand(int stranded_type, unsigned sam_flag, char site_strand)
{
    char read_strand = 0;
    int first_or_single = (sam_flag & 64u) || !(sam_flag & 1u);
    if (stranded_type == 1) {
        if (first_or_single) read_strand = (sam_flag & 16u) ? '-' : '+';
        else read_strand = (sam_flag & 16u) ? '+' : '-';
    }
    if (stranded_type == 2) {
        if (first_or_single) read_strand = (sam_flag & 16u) ? '+' : '-';
        else read_strand = (sam_flag & 16u) ? '-' : '+';
    }
    return read_strand == site_strand;
}

static int in_list(const int32_t *list, uint32_t n, int64_t v)
{
    for (uint32_t i = 0; i < n; ++i) if ((int64_t)list[i] == v) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * One SAM line against one site: SpliSER_v0_1_8.py:436-559.
 * Counters are updated in place exactly where the reference calls the Site adders.
 * ---------------------------------------------------------------------------------------------- */
static void check_one_read(int64_t target, char site_strand, const int32_t *partners, uint32_t n_partners,
                           const int32_t *competitors, uint32_t n_competitors, int64_t left_bound, unsigned flag,
                           const uint32_t *ops, uint32_t n_ops, int is_stranded, int stranded_type, int combine_mode,
                           uint32_t *beta1_count, uint32_t *beta2simple_count, uint32_t *double_counts /* per partner */)
{
    /* spliceSites list (:442, :484-485) */
    int64_t stack_sites[64];
    int64_t *splice_sites = stack_sites;
    uint32_t n_splice = 0, cap_splice = 64;

    int64_t partner_used = 0;
    int have_partner_used = 0;      /* partnerUsed = "" (:443): never equal to an int */
    int comp_splicing = 0;          /* :444 */
    int alpha_read = 0, beta1_read = 0, flanking_read = 0, beta1type_read = 0, mutex_read = 0; /* :446-451 */
    int mapped_region = 0, progression = 0;
    int64_t current = left_bound;   /* :452 */

    for (uint32_t i = 0; i < n_ops; ++i) {
        int64_t d = (int64_t)(ops[i] >> 4);
        unsigned c = ops[i] & 15u;
        if (c == 0 || c == 8 || c == 7) { mapped_region = 1; progression = 1; }  /* M X = (:457-459) */
        else if (c == 3 || c == 2) { mapped_region = 0; progression = 1; }        /* N D   (:460-462) */
        else { progression = 0; }   /* I S H P (:463-464); codes > 8 cannot be printed by samtools */

        if (!progression) continue;
        current += d;                                                             /* :467 */
        if (target >= current - d && current > target && current > target + 1) { /* :469 */
            if (mapped_region) {
                if (is_stranded) { if (orc_check_strand(stranded_type, flag, site_strand)) beta1_read = 1; }
                else beta1_read = 1;
            }
        }
        if (c == 3) {                                                             /* :480 */
            int64_t l_site = current - d - 1, r_site = current - 1;               /* :482-483 */
            if (n_splice + 2 > cap_splice) {
                cap_splice *= 2;
                int64_t *grown = (int64_t *)malloc(sizeof(int64_t) * cap_splice);
                memcpy(grown, splice_sites, sizeof(int64_t) * n_splice);
                if (splice_sites != stack_sites) free(splice_sites);
                splice_sites = grown;
            }
            splice_sites[n_splice++] = l_site;
            splice_sites[n_splice++] = r_site;
            if (l_site == target) { partner_used = r_site; have_partner_used = 1; alpha_read = 1; } /* :487-489 */
            if (r_site == target) { partner_used = l_site; have_partner_used = 1; alpha_read = 1; } /* :490-492 */
            if (in_list(competitors, n_competitors, r_site) && in_list(partners, n_partners, l_site)) comp_splicing = 1;
            if (in_list(competitors, n_competitors, l_site) && in_list(partners, n_partners, r_site)) comp_splicing = 1;
            if (comp_splicing && target > l_site && target < r_site) flanking_read = 1;             /* :503-505 */
            if (!alpha_read && !comp_splicing && target > l_site && target < r_site) {              /* :507 */
                if (is_stranded) { if (orc_check_strand(stranded_type, flag, site_strand)) mutex_read = 1; }
                else mutex_read = 1;
            }
        }
    }

    if (beta1_read && comp_splicing) beta1type_read = 1;                                            /* :516-517 */

    if (alpha_read && comp_splicing) {                                                              /* :519 */
        for (uint32_t p = 0; p < n_partners; ++p) {      /* set(partners) & set(spliceSites): keys are unique */
            int hit = 0;
            for (uint32_t k = 0; k < n_splice; ++k) if (splice_sites[k] == (int64_t)partners[p]) hit = 1;
            if (hit && !(have_partner_used && (int64_t)partners[p] == partner_used)) double_counts[p] += 1;
        }
    } else if (flanking_read) {                                                                     /* :529 */
        if (combine_mode) *beta2simple_count += 1;                                                  /* :531-532 */
    } else if (mutex_read) {                                                                        /* :540 */
        *beta2simple_count += 1;
    } else if (beta1type_read) {                                                                    /* :544 */
        for (uint32_t p = 0; p < n_partners; ++p) {
            int hit = 0;
            for (uint32_t k = 0; k < n_splice; ++k) if (splice_sites[k] == (int64_t)partners[p]) hit = 1;
            if (hit) double_counts[p] += 1;
        }
        *beta2simple_count += 1;
    } else if (beta1_read && !beta1type_read) {                                                     /* :558 */
        *beta1_count += 1;
    }
    if (splice_sites != stack_sites) free(splice_sites);
}

/* ------------------------------------------------------------------------------------------------
 * processSites, first loop (SpliSER_v0_1_8.py:686-688): checkBam for every site of one chromosome.
 * stranded: 0 = unstranded, 1 = fr, 2 = rf.  Outputs are overwritten.
 * Returns 0, or -1 on allocation failure.
 * ---------------------------------------------------------------------------------------------- */
int orc_check_bam(int64_t n_sites, const int32_t *site_pos, const uint8_t *site_strand, const uint32_t *part_off,
                  const int32_t *part_pos, const uint32_t *comp_off, const int32_t *comp_pos, int64_t n_reads,
                  const int32_t *r_pos, const uint16_t *r_flag, const uint32_t *cig_off, const uint32_t *cigar,
                  int stranded, int combine_mode, uint32_t *beta1, uint32_t *beta2s_reads, uint32_t *dbl)
{
    fetch_index fi;
    if (fetch_index_build(&fi, n_reads, r_pos, r_flag, cig_off, cigar) != 0) { fetch_index_free(&fi); return -1; }
    memset(beta1, 0, sizeof(uint32_t) * (size_t)n_sites);
    memset(beta2s_reads, 0, sizeof(uint32_t) * (size_t)n_sites);
    if (n_sites > 0) memset(dbl, 0, sizeof(uint32_t) * (size_t)part_off[n_sites]);

    /* The CPU-baseline leg of bench.py may ask for several threads (orc_set_threads): the sites of a chromosome are
     * independent of each other (every output below belongs to one site), so the loop is simply shared out. */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 128) num_threads(g_threads > 0 ? g_threads : 1)
#endif
    for (int64_t s = 0; s < n_sites; ++s) {
        const int64_t t = site_pos[s];
        /* region chr:t-(t+1)  ->  0-based half-open [t-1, t+1) */
        const int64_t beg0 = t - 1, end0 = t + 1;
        int64_t hi = count_pos_below(&fi, end0);
        for (int64_t i = hi - 1; i >= 0 && fi.pmaxend[i] > beg0; --i) {
            const int64_t r = fi.order[i];
            if (!(fi.endpos0[r] > beg0)) continue;          /* not overlapping: samtools does not print it */
            const int64_t left_bound = r_pos[r];
            if (!(left_bound <= t)) continue;                /* :435 */
            check_one_read(t, (char)site_strand[s], part_pos + part_off[s], part_off[s + 1] - part_off[s],
                           comp_pos + comp_off[s], comp_off[s + 1] - comp_off[s], left_bound, r_flag[r],
                           cigar + cig_off[r], cig_off[r + 1] - cig_off[r], stranded != 0, stranded, combine_mode,
                           &beta1[s], &beta2s_reads[s], dbl + part_off[s]);
        }
    }
    fetch_index_free(&fi);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * processSites, second loop (SpliSER_v0_1_8.py:690-692): findBeta2Counts + calculateSSE per site,
 * one sample.  part_site[e] = row of the partner Site (Site.Partners entry) or -1.
 * Python ints are unbounded and int/int is correctly rounded; here counts are int64 and every value
 * met in practice is < 2^53, where (double)a/(double)b is the same correctly rounded quotient.
 * ---------------------------------------------------------------------------------------------- */
int orc_beta2_sse(int64_t n_sites, const int32_t *site_pos, const uint32_t *part_off, const int32_t *part_pos,
                  const int32_t *part_site, const int64_t *alpha, const int64_t *edge_cnt, const uint32_t *beta1,
                  const uint32_t *beta2s_reads, const uint32_t *dbl, int beta2_cryptic, int64_t *beta2_simple,
                  int64_t *beta2_cryptic_count, double *beta2_weighted, double *sse)
{
    (void)part_pos;
    for (int64_t s = 0; s < n_sites; ++s) {
        const int64_t t = site_pos[s];
        int64_t b2simple = beta2s_reads[s];
        int64_t cryptic = 0;          /* beta2CrypticCounts (:585) */
        double weighted = 0.00;       /* beta2CrypticWeighted (:586) */
        const int64_t total_alpha = alpha[s];

        for (uint32_t e = part_off[s]; e < part_off[s + 1]; ++e) {   /* for pSite in Partners (:590) */
            const int32_t p = part_site[e];
            if (p < 0) continue;
            const int64_t ppos = site_pos[p];
            int64_t doubles = dbl[e];
            int have_double_key = dbl[e] != 0;                        /* dict key created by checkBam adders */
            /* PartnerBeta2DoubleCounts is keyed by the partner's POSITION (:598): what an earlier partner at the same position
             * added to it (two Site objects at one position: strands '+' and '?', a junction whose ends coincide) is there when
             * this one is looked at (:608) */
            for (uint32_t e2 = part_off[s]; e2 <= e; ++e2) {
                const int32_t q = part_site[e2];
                if (q < 0 || site_pos[q] != ppos) continue;
                for (uint32_t f = part_off[q]; f < part_off[q + 1]; ++f) { /* pSite.getPartnerCounts().items() (:592) */
                    const int64_t cpos = part_pos[f];
                    const int64_t cnt = edge_cnt[f];
                    if (f > part_off[q]) { /* (a dict: a position listed twice by q's own edges is one item) */
                        int seen = 0;
                        for (uint32_t f2 = part_off[q]; f2 < f; ++f2) seen |= part_pos[f2] == part_pos[f];
                        if (seen) continue;
                    }
                    if ((ppos > t && cpos < t) || (ppos < t && cpos > t)) { /* :594-599 */
                        if (e2 == e) b2simple += cnt;
                        doubles += cnt;
                        have_double_key = 1;
                    }
                }
            }
            const int64_t p_alpha = alpha[p];                          /* :602 */
            const int64_t shared = edge_cnt[e];                         /* PartnerCounts[pSite.pos] (:604) */
            int64_t b2 = p_alpha - shared;                              /* :606 */
            if (have_double_key) { b2 = b2 - doubles; if (b2 < 0) b2 = 0; } /* :608-611, :574-579 */
            cryptic += b2;                                              /* :613 */
            double w = 0.0;                                             /* trueDivCatchZero (:562-572) */
            if ((double)total_alpha > 0.0) w = (double)shared / (double)total_alpha;
            double wb2 = (double)b2 * w;                                /* :618 */
            weighted = weighted + wb2;                                  /* :619 */
        }
        beta2_simple[s] = b2simple;
        beta2_cryptic_count[s] = cryptic;
        beta2_weighted[s] = weighted;

        /* calculateSSE (:626-639) */
        const int64_t betas_int = (int64_t)beta1[s] + b2simple;         /* :631 */
        double value = 0.0;
        if (beta2_cryptic) {
            double betas = (double)betas_int + weighted;                /* :635 */
            double denom = (double)total_alpha + betas;                 /* :637 */
            if (denom > 0.0) value = (double)total_alpha / denom;       /* :639 via :570-571 */
        } else {
            int64_t denom = total_alpha + betas_int;
            if ((double)denom > 0.0) value = (double)total_alpha / (double)denom;
        }
        sse[s] = value;
    }
    return 0;
}
