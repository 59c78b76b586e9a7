"""Steps 1-2 of ``process`` for one sample, array-at-a-time.

``sites.SiteTable`` restates findAlphaCounts / findCompetitorPos line by line on Python objects (it is what the goldens
pin, and ``combine`` needs its objects).  For a single sample the same table follows from a few sorts:

  * a site is a distinct (position[, strand]) among the junction ends, in first-appearance order for everything the
    reference decides at creation (strand text, gene) and in ``Site.__lt__`` order for the rows;
  * alpha is a sum over the lines that touch the site, PartnerCounts a sum per (site, partner position) in
    first-appearance order, Competitors the partner positions of the partners other than the site's own.

``build`` returns None whenever the input leaves that regime (a gene query, strands other than + / - in a stranded run):
the caller then takes ``sites.SiteTable``.  tests/test_fast_sites.py holds both builders to identical arrays on every
golden input and on random tables.

Reference map: findAlphaCounts SpliSER_v0_1_8.py:227-362, binary_site_search :175-225, binary_gene_search :118-173
(native: spl_gene_search), findCompetitorPos :364-372.
"""
import numpy as np

from . import native
from .sites import ChromArrays


class FastSiteTable(object):
    """The slice of SiteTable's interface that ``process`` uses."""

    def __init__(self, bins, is_stranded):
        self.bins = bins
        self.is_stranded = bool(is_stranded)
        self.chrom_index = bins.chrom_index
        self.assessed = self.created = self.assigned = 0
        self._arrays = {}

    def n_sites(self):
        return sum(a.n for a in self._arrays.values())

    def chrom_arrays(self, chrom):
        arr = self._arrays.get(chrom)
        if arr is None:
            arr = _empty(chrom)
        return arr

    def find_competitors(self):
        """Competitors are part of the arrays already (kept so that ``process`` reads like the reference)."""


def _empty(chrom):
    out = ChromArrays()
    out.chrom, out.n = chrom, 0
    out.pos = np.zeros(0, np.int64)
    out.strand = np.zeros(0, np.uint8)
    out.alpha = np.zeros(0, np.int64)
    out.gene_idx, out.gene_names = np.zeros(0, np.int64), []
    out.part_off = np.zeros(1, np.uint32)
    out.part_pos = np.zeros(0, np.int64)
    out.part_site = np.zeros(0, np.int32)
    out.edge_cnt = np.zeros(0, np.int64)
    out.comp_off = np.zeros(1, np.uint32)
    out.comp_pos = np.zeros(0, np.int64)
    return out


def _strand_code(s):
    """What spl_gene_search compares: '+' and '-' as they are, anything else 0."""
    return 43 if s == "+" else (45 if s == "-" else 0)


def build(bins, is_stranded, bed_path, q_chrom="All", q_gene="All", max_intron=0):
    """-> FastSiteTable, or None when the line-by-line builder has to be used."""
    if q_gene != "All":
        return None
    cols = native.read_bed_columns(bed_path)     # the 12-column lines as findAlphaCounts reads them (:257-277), natively
    if cols is None:
        return None                               # (a line the native reader will not vouch for)
    table = FastSiteTable(bins, is_stranded)
    for chrom in cols.chrom_names:                # :265-268 (also for chromosomes the -c filter drops)
        bins.ensure_chrom(chrom)
    for k, chrom in enumerate(cols.chrom_names):
        if not (q_chrom == chrom or q_chrom == "All"):
            continue
        rows = np.flatnonzero(cols.chrom == k)    # line order
        arr = _chrom(table, chrom, cols.left[rows], cols.right[rows], cols.strand[rows], cols.alpha[rows])
        if arr is None:
            return None
        table._arrays[chrom] = arr
    return table


def _chrom(table, chrom, left, right, strand, alpha):
    """One chromosome from its lines (arrays in line order; ``strand`` = the strand column's byte, 0 when empty)."""
    is_stranded = table.is_stranded
    k = left.shape[0]
    if is_stranded and bool(((strand != 43) & (strand != 45)).any()):
        return None          # strand-free look-ups depend on what has been inserted so far: line by line
    if k and (min(int(left.min()), int(right.min())) < 0 or max(int(left.max()), int(right.max())) >= (1 << 61)):
        return None
    if bool((left == right).any()):
        return None          # both look-ups precede both insertions (:291-292): such a line creates TWO sites at one position
    minus = strand == 45
    # junction ends in the order the reference looks them up: line 0 left, line 0 right, line 1 left, ...
    pos = np.empty(2 * k, np.int64)
    pos[0::2], pos[1::2] = left, right
    line = np.repeat(np.arange(k, dtype=np.int64), 2)
    key = pos * 2 + np.repeat(minus, 2) if is_stranded else pos      # Site.__lt__: position, then '+' before '-'
    ukey, first, inv = np.unique(key, return_index=True, return_inverse=True)
    n = ukey.shape[0]
    table.assessed += 2 * k
    table.created += n
    out = ChromArrays()
    out.chrom, out.n = chrom, n
    out.pos = (ukey >> 1) if is_stranded else ukey.copy()
    creator = line[first]                                   # the line whose look-up created the site
    out.strand = strand[creator].astype(np.uint8)           # (first byte of the creating line's strand column, 0 = empty;
    #                                                          strand_text follows from it when somebody asks)
    out.alpha = np.zeros(n, np.int64)
    np.add.at(out.alpha, inv, np.repeat(alpha, 2))           # :341
    # genes (:313): one bisection per created site, with the creating line's strand
    g_left, g_right, g_strand, names = table.bins.gene_arrays(chrom)
    if len(names) and n:
        q_strand = np.where((out.strand == 43) | (out.strand == 45), out.strand, 0).astype(np.uint8)
        gi = native.gene_search(g_left, g_right, g_strand, out.pos, q_strand, is_stranded)
        table.assigned += int((gi >= 0).sum())
    else:
        gi = np.full(n, -1, np.int64)
    out.gene_idx, out.gene_names = np.asarray(gi, np.int64), names   # (the Gene column; a list of str only on demand)
    # partner edges (:352-355): site of an end -> position of the other end, summed per (site, position), listed in
    # first-appearance order
    src = inv
    dst = np.empty(2 * k, np.int64)
    dst[0::2], dst[1::2] = inv[1::2], inv[0::2]
    dst_pos = out.pos[dst]
    w = np.repeat(alpha, 2)
    order = np.lexsort((np.arange(2 * k), dst_pos, src))     # by site, partner position, appearance
    s_src, s_pos = src[order], dst_pos[order]
    head = np.ones(2 * k, bool)
    head[1:] = (s_src[1:] != s_src[:-1]) | (s_pos[1:] != s_pos[:-1])
    starts = np.flatnonzero(head)
    e_src = s_src[starts]
    e_pos = s_pos[starts]
    e_cnt = np.add.reduceat(w[order], starts) if 2 * k else np.zeros(0, np.int64)
    e_first = order[starts]                                   # appearance of the edge = its first line end
    e_site = dst[e_first]                                     # "first Partners entry wins for a given position"
    by_seen = np.lexsort((e_first, e_src))
    e_src, e_pos, e_cnt, e_site = e_src[by_seen], e_pos[by_seen], e_cnt[by_seen], e_site[by_seen]
    deg = np.bincount(e_src, minlength=n)
    out.part_off = np.zeros(n + 1, np.uint32)
    np.cumsum(deg, out=out.part_off[1:])
    out.part_pos = e_pos.astype(np.int64)
    out.part_site = e_site.astype(np.int32)
    out.edge_cnt = e_cnt.astype(np.int64)
    # competitors (:364-372): positions of my partners' partners other than my own position, sorted, unique
    p_off = out.part_off.astype(np.int64)
    fan = deg[e_site]                                         # partners of each partner
    total = int(fan.sum())
    owner = np.repeat(e_src, fan)
    base = np.repeat(p_off[e_site], fan)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(fan) - fan, fan)
    cand = out.part_pos[base + within] if total else np.zeros(0, np.int64)
    keep = cand != out.pos[owner] if total else np.zeros(0, bool)
    owner, cand = owner[keep], cand[keep]
    o2 = np.lexsort((cand, owner))
    owner, cand = owner[o2], cand[o2]
    uniq = np.ones(owner.shape[0], bool)
    uniq[1:] = (owner[1:] != owner[:-1]) | (cand[1:] != cand[:-1])
    owner, cand = owner[uniq], cand[uniq]
    out.comp_off = np.zeros(n + 1, np.uint32)
    np.cumsum(np.bincount(owner, minlength=n), out=out.comp_off[1:])
    out.comp_pos = cand.astype(np.int64)
    return out
