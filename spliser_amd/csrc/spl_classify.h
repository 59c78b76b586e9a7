// spl_classify.h -- the per-(read, site) decision of SpliSER's checkBam as straight-line C++.
//
// This header is the arithmetic core of the classification kernel (spl_kernels.hip includes it with
// SPL_HD = __device__ __forceinline__).  It contains no HIP intrinsics so that tests/ can also
// compile it with g++ and drive it pair by pair against the oracle in the build container, where no
// GPU exists; the product never calls it on the host.
//
// Reference: SpliSER_v0_1_8.py:436-559 (CIGAR walk + classification) and :374-406 (check_strand).
#ifndef SPL_CLASSIFY_H
#define SPL_CLASSIFY_H

#include <stdint.h>

#ifndef SPL_HD
#define SPL_HD inline
#endif

// BAM CIGAR op codes (SAM spec 4.2): MIDNSHP=X -> 0..8
enum { SPL_OP_M = 0, SPL_OP_I = 1, SPL_OP_D = 2, SPL_OP_N = 3, SPL_OP_S = 4, SPL_OP_H = 5, SPL_OP_P = 6, SPL_OP_EQ = 7, SPL_OP_X = 8 };

// Outcome of the exclusive if/elif chain at SpliSER_v0_1_8.py:519-559.
enum {
    SPL_CLS_NONE = 0,       // no counter changes
    SPL_CLS_ALPHA_COMP = 1, // :519  alpha read that also shows competitive splicing -> double counts (p != partnerUsed)
    SPL_CLS_FLANK = 2,      // :529  flanking read: counts toward beta2Simple only in combine mode
    SPL_CLS_ME = 3,         // :540  mutually-exclusive splicing -> beta2Simple += 1
    SPL_CLS_B1TYPE = 4,     // :544  unspliced at the site + competitive splicing -> double counts, beta2Simple += 1
    SPL_CLS_BETA1 = 5       // :558  plain beta1 -> beta1 += 1
};

// Ops that consume the reference ("progression", :457-462) and, of those, the ones that are aligned
// bases ("mappedRegion").  Bit i of the mask = op code i.
#define SPL_PROG_MASK ((1u << SPL_OP_M) | (1u << SPL_OP_D) | (1u << SPL_OP_N) | (1u << SPL_OP_EQ) | (1u << SPL_OP_X))
#define SPL_MAPPED_MASK ((1u << SPL_OP_M) | (1u << SPL_OP_EQ) | (1u << SPL_OP_X))

// check_strand (:374-406): '+' or '-' the read is taken to originate from.  stranded: 1 = fr, 2 = rf.
SPL_HD uint8_t spl_read_strand(uint32_t flag, int stranded)
{
    const bool first = (flag & 64u) || !(flag & 1u);
    const bool rev = (flag & 16u) != 0;
    bool minus = first ? rev : !rev; // "fr"
    if (stranded == 2) minus = !minus; // "rf"
    return minus ? (uint8_t)'-' : (uint8_t)'+';
}

SPL_HD bool spl_contains(const int32_t *list, uint32_t n, int32_t v)
{
    bool hit = false;
    for (uint32_t i = 0; i < n; ++i) hit |= (list[i] == v);
    return hit;
}

// Reference length and N-op presence of one read (one pass over its ops).
SPL_HD void spl_read_extent(const uint32_t *ops, uint32_t n_ops, int64_t *ref_len, bool *has_n)
{
    int64_t len = 0;
    bool hn = false;
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if ((SPL_PROG_MASK >> code) & 1u) len += (int64_t)(op >> 4);
        hn |= (code == SPL_OP_N);
    }
    *ref_len = len;
    *has_n = hn;
}

// Number of sites a read is evaluated against: `samtools view chr:t-(t+1)` returns it when it overlaps
// [t, t+1] (htslib: pos0 < t+1 && bam_endpos > t-1, with bam_endpos = pos0 + (rlen ? rlen : 1) and
// rlen = 0 for flag 0x4), and checkBam keeps it when POS <= t (:435).  Together: POS <= t <= POS+L-1.
SPL_HD int64_t spl_fetch_len(uint32_t flag, int64_t ref_len)
{
    if (flag & 4u) return 1;
    return ref_len > 0 ? ref_len : 1;
}

struct spl_pair {
    int cls;             // SPL_CLS_*
    int32_t partner_used; // :488/:491, meaningful when has_partner_used
    bool has_partner_used;
};

// The CIGAR walk of checkBam for ONE read against ONE site t.
//   pos        SAM POS of the read (1-based)
//   ops/n_ops  BAM-native CIGAR
//   t          Site.pos
//   part/n_part  keys of the site's PartnerCounts (:417-419);  comp/n_comp  its CompetitorPos (:414)
//   strand_ok  !isStranded || check_strand(strandedType, flag, siteStrand)
// Positions are int32; the caller guarantees pos + ref_len fits (SPL_ERR_RANGE otherwise).
SPL_HD spl_pair spl_classify_pair(int32_t pos, const uint32_t *ops, uint32_t n_ops, int32_t t,
                                  const int32_t *part, uint32_t n_part, const int32_t *comp, uint32_t n_comp,
                                  bool strand_ok)
{
    bool alpha = false, beta1 = false, comp_spl = false, flank = false, me = false, has_pu = false;
    int32_t pu = 0;
    int32_t cur = pos;
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if (!((SPL_PROG_MASK >> code) & 1u)) continue; // I,S,H,P: no progression (:463-464)
        const int32_t d = (int32_t)(op >> 4);
        const int32_t start = cur;
        cur += d;
        // :469  one single op covers both t and t+1
        if (t >= start && cur > t && cur > t + 1) {
            if (((SPL_MAPPED_MASK >> code) & 1u) && strand_ok) beta1 = true;
        }
        if (code == SPL_OP_N) {
            const int32_t l = start - 1; // :482
            const int32_t r = cur - 1;   // :483
            if (l == t) { pu = r; has_pu = true; alpha = true; }
            if (r == t) { pu = l; has_pu = true; alpha = true; }
            if (n_comp != 0u) { // both tests need a competitor position
                if (spl_contains(comp, n_comp, r) && spl_contains(part, n_part, l)) comp_spl = true; // :494-497
                if (spl_contains(comp, n_comp, l) && spl_contains(part, n_part, r)) comp_spl = true; // :498-501
            }
            const bool inside = (t > l) && (t < r);
            if (comp_spl && inside) flank = true;                                // :503-505
            if (!alpha && !comp_spl && inside && strand_ok) me = true;           // :507-512
        }
    }
    spl_pair out;
    out.partner_used = pu;
    out.has_partner_used = has_pu;
    if (alpha && comp_spl) out.cls = SPL_CLS_ALPHA_COMP;
    else if (flank) out.cls = SPL_CLS_FLANK;
    else if (me) out.cls = SPL_CLS_ME;
    else if (beta1 && comp_spl) out.cls = SPL_CLS_B1TYPE;
    else if (beta1) out.cls = SPL_CLS_BETA1;
    else out.cls = SPL_CLS_NONE;
    return out;
}

// spliceSites membership (:484-485, :522, :547): is v the lSite or rSite of any N op of the read?
SPL_HD bool spl_read_splices_at(int32_t pos, const uint32_t *ops, uint32_t n_ops, int32_t v)
{
    bool hit = false;
    int32_t cur = pos;
    for (uint32_t k = 0; k < n_ops; ++k) {
        const uint32_t op = ops[k];
        const uint32_t code = op & 15u;
        if (!((SPL_PROG_MASK >> code) & 1u)) continue;
        const int32_t d = (int32_t)(op >> 4);
        cur += d;
        if (code == SPL_OP_N) hit |= (cur - d - 1 == v) | (cur - 1 == v);
    }
    return hit;
}

#endif // SPL_CLASSIFY_H
