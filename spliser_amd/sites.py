"""Host-side site table: Steps 0-2 of SpliSER ``process`` restated for the MI355X build.

What the reference keeps as lists of ``Gene`` / ``Site`` objects in module globals
(SpliSER_v0_1_8.py:23-25) is built here once per run and then flattened, per chromosome, into the
SoA + CSR arrays the HIP kernels consume (``include/spliser.h``: ``spl_sites``).

Reference map
    GeneBins.from_annotation   createGenes          SpliSER_v0_1_8.py:50-116
    gene_search                binary_gene_search   SpliSER_v0_1_8.py:118-173
    SiteTable.add_bed          findAlphaCounts      SpliSER_v0_1_8.py:227-362 (+ binary_site_search :175-225)
    SiteTable.find_competitors findCompetitorPos    SpliSER_v0_1_8.py:364-372
    SiteTable.chrom_arrays     Site fields          Gene_Site_Iter_Graph_v0_1_8.py:98-120

Everything here is O(junctions); none of it is on the GPU hot path, but it defines row order, the
alpha / Partners / Competitors columns and the Gene column, so it follows the reference's rules
(including the ones that look accidental) exactly.
"""
import bisect
import os

import numpy as np

__all__ = ["Gene", "GeneBins", "gene_search", "SiteTable", "ChromArrays"]


class Gene(object):
    """Gene_Site_Iter_Graph_v0_1_8.py:10-78 -- only the fields ``process`` reads."""
    __slots__ = ("name", "left", "right", "strand")

    def __init__(self, name, left, right, strand):
        self.name = str(name)
        self.left = int(left)
        self.right = int(right)
        self.strand = str(strand)


NA_GENE = Gene("NA", -1, -1, None)  # SpliSER_v0_1_8.py:40-47
USE_NATIVE_TEXT = True              # tests switch it off to hold the native readers to the line-by-line ones


def _first_attribute(field):
    """HTSeq names a GFF feature after the value of the FIRST attribute of column 9
    (HTSeq.parse_GFF_attribute_string(..., extra_return_first_value=True)); quotes are stripped."""
    first = field.strip().split(";")[0].strip()
    if "=" in first:
        value = first.split("=", 1)[1]
    elif " " in first:
        value = first.split(" ", 1)[1]
    else:
        value = first
    return value.strip().strip('"')


class GeneBins(object):
    """Per-chromosome gene lists ordered by left boundary (bisect.insort on Gene.__lt__,
    Gene_Site_Iter_Graph_v0_1_8.py:27-28), plus the chromosome order they induce.

    Two forms of the same lists: ``genes`` (Gene objects: what the line-by-line builders walk) and ``gene_arrays`` (columns:
    what the array-at-a-time builder bisects natively).  An annotation read by the native reader starts out as columns and
    grows its objects only if somebody asks for them."""

    def __init__(self):
        self.chrom_index = []   # SpliSER_v0_1_8.py:23
        self._genes = {}        # chrom -> list[Gene]      (gene2D_array, :24)
        self._lefts = {}        # chrom -> list[int] parallel to genes[chrom], for insort
        self._columns = None    # chrom -> (left, right, strand code, names) in list order, when read natively
        self._chroms = set()
        self.query_gene = None  # QUERY_gene, :39
        self.n_created = 0

    @property
    def genes(self):
        if self._columns is not None:      # objects on demand
            for chrom, (left, right, strand, names) in self._columns.items():
                self._genes[chrom] = [Gene(n, l, r, chr(s)) for n, l, r, s in zip(names, left.tolist(), right.tolist(), strand.tolist())]
                self._lefts[chrom] = left.tolist()
            self._columns = None
        return self._genes

    def has_chrom(self, chrom):
        return chrom in self._chroms

    def ensure_chrom(self, chrom):
        if chrom not in self._chroms:
            self._chroms.add(chrom)
            self.chrom_index.append(chrom)
            if self._columns is not None:
                self._columns[chrom] = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint8), [])
            else:
                self._genes[chrom] = []
                self._lefts[chrom] = []

    def gene_arrays(self, chrom):
        """-> (left, right, strand code ('+' 43, '-' 45, anything else 0), names) of a chromosome's genes, list order."""
        if self._columns is not None:
            left, right, strand, names = self._columns.get(chrom, (np.zeros(0, np.int64),) * 2 + (np.zeros(0, np.uint8), []))
            return left, right, np.where((strand == 43) | (strand == 45), strand, 0).astype(np.uint8), names
        genes = self._genes.get(chrom, [])
        k = len(genes)
        return (np.fromiter((g.left for g in genes), dtype=np.int64, count=k), np.fromiter((g.right for g in genes), dtype=np.int64, count=k),
                np.fromiter((43 if g.strand == "+" else (45 if g.strand == "-" else 0) for g in genes), dtype=np.uint8, count=k),
                [g.name for g in genes])

    @classmethod
    def from_annotation(cls, path, a_type="gene", q_gene="All", log=None):
        """createGenes.  ``a_type`` is accepted and ignored exactly like the reference does
        (it tests ``line.type == 'gene'`` literally, SpliSER_v0_1_8.py:82).  Coordinates follow
        HTSeq.GFF_Reader: start = column4 - 1, end = column5 (0-based, half-open)."""
        bins = cls()
        if q_gene == "All" and USE_NATIVE_TEXT:
            cols = None
            try:
                from . import native
                cols = native.read_gff_genes(path)
            except Exception:
                cols = None
            if cols is not None:
                bins._columns = {}
                for k, chrom in enumerate(cols.chrom_names):
                    rows = np.flatnonzero(cols.chrom == k)
                    order = rows[np.argsort(cols.left[rows], kind="stable")]   # = bisect.insort by left, later lines after equal ones
                    bins._chroms.add(chrom)
                    bins.chrom_index.append(chrom)
                    bins._columns[chrom] = (cols.left[order], cols.right[order], cols.strand[order], [cols.names[i] for i in order.tolist()])
                bins.n_created = int(cols.chrom.shape[0])
                if log:
                    log("%d Genes created in %d bins" % (bins.n_created, len(bins.chrom_index)))
                return bins
        with open(path, "r") as handle:
            for raw in handle:
                if raw.startswith("#") or not raw.strip():
                    continue
                cols = raw.rstrip("\n").split("\t")
                if len(cols) < 9 or cols[2] != "gene":
                    continue
                chrom = cols[0]
                gene = Gene(_first_attribute(cols[8]), int(cols[3]) - 1, int(cols[4]), cols[6])
                bins.ensure_chrom(chrom)
                if q_gene == "All":
                    bins.n_created += 1
                    at = bisect.bisect_right(bins._lefts[chrom], gene.left)
                    bins._lefts[chrom].insert(at, gene.left)
                    bins._genes[chrom].insert(at, gene)
                elif gene.name == q_gene:
                    if log:
                        log("Query Gene found")
                    bins.query_gene = gene
                    bins._genes[chrom].append(gene)
                    bins._lefts[chrom].append(gene.left)
        if log:
            log("%d Genes created in %d bins" % (bins.n_created, len(bins.chrom_index)))
        return bins


def _strand_free(strand, is_stranded):
    """``isStranded == False or (strand != '+' and strand != '-')`` -- the query does not care about strand."""
    return (not is_stranded) or (strand != "+" and strand != "-")


def gene_search(genes, pos, strand, is_stranded):
    """binary_gene_search (SpliSER_v0_1_8.py:118-173): index of a gene containing ``pos`` or -1.

    The bisection below keeps the reference's bookkeeping (``lo``/``hi`` trackers, the idx == 1
    special case, "stuck" detection when an index repeats) because overlapping genes make the list
    only partially ordered and the outcome depends on the exact probe sequence; so does the
    last-ditch sweep over offsets -3..2 that re-bases itself on every hit and never looks at the
    final list element.
    """
    n = len(genes)
    if n == 0:
        return -1
    free = _strand_free(strand, is_stranded)
    pos = int(pos)
    idx = n // 2
    hi, lo, last, nxt = n, 0, -1, n // 2
    while True:
        g = genes[idx]
        if g.left <= pos <= g.right and (free or strand == g.strand):
            return idx
        if pos >= g.right:
            nxt = idx + (hi - idx) // 2
            lo = idx
        elif pos <= g.left:
            nxt = idx - (idx - lo) // 2
            hi = idx
            if idx == 1:
                nxt = 0
        if idx == last:
            break
        last, idx = idx, nxt
    found = False
    for off in range(-3, 3):
        j = idx + off
        if 0 <= j < n - 1 and genes[j].left <= pos <= genes[j].right:
            if free or strand == genes[j].strand:
                found = True
                idx = j
    return idx if found else -1


class _Site(object):
    """Gene_Site_Iter_Graph_v0_1_8.py:98-120, one sample."""
    __slots__ = ("pos", "strand", "gene", "alpha", "partner_counts", "partner_sites", "competitors", "row")

    def __init__(self, pos, strand, gene):
        self.pos = pos
        self.strand = strand
        self.gene = gene
        self.alpha = 0
        self.partner_counts = {}   # PartnerCounts: partner pos -> shared alpha (insertion ordered)
        self.partner_sites = []    # Partners: Site objects, first-seen order
        self.competitors = []      # CompetitorPos (sorted unique)
        self.row = -1


_STRAND_RANK = {"+": 0, "-": 1}


def _site_before(a, b, is_stranded):
    """Site.__lt__ (Gene_Site_Iter_Graph_v0_1_8.py:123-136): by position; at one position of a stranded analysis '+' is before '-'
    and every other pair of strands answers None there -- not before."""
    if a.pos != b.pos:
        return a.pos < b.pos
    return bool(is_stranded and a.strand == "+" and b.strand == "-")


def _array_insort(arr, site, is_stranded):
    """bisect.insort(arr, site): behind everything it is not before."""
    lo, hi = 0, len(arr)
    while lo < hi:
        mid = (lo + hi) // 2
        if _site_before(site, arr[mid], is_stranded):
            hi = mid
        else:
            lo = mid + 1
    arr.insert(lo, site)


def _array_search(arr, pos, strand, is_stranded):
    """binary_site_search (SpliSER_v0_1_8.py:175-225) on the reference's own list: the index it returns, or -1.  Its bisection
    keeps two trackers and gives up when it visits an index twice in a row; a site at the position that does not suit the query
    is followed by ONE look at each neighbour (below first, the one above wins) and the search ends there."""
    n = len(arr)
    if n == 0:          # (the reference does not search an empty list, :290-296)
        return -1
    free = _strand_free(strand, is_stranded)
    pos = int(pos)
    idx = nxt = n // 2
    top, bottom, last = n, 0, -1
    while True:
        here = arr[idx].pos
        if here == pos:
            if free or arr[idx].strand == strand:
                return idx
            hit = -1
            for k in (idx - 1, idx + 1):
                if 0 <= k < n and arr[k].pos == pos and arr[k].strand == strand:
                    hit = k
            return hit
        if pos >= here:
            nxt = idx + (top - idx) // 2
            bottom = idx
        else:
            nxt = idx - (idx - bottom) // 2
            top = idx
            if idx == 1:
                nxt = 0
        if idx == last:
            return -1
        last, idx = idx, nxt


class ChromArrays(object):
    """SoA + CSR view of one chromosome's sites (the payload of ``spl_sites``).

    ``genes`` / ``strand_text`` (one str per row, what the Gene and Strand columns print) are lists when the line-by-line
    builder made the table; the array builder (fast_sites) leaves ``gene_idx`` (row -> index into ``gene_names``, -1 = NA) and
    the strand bytes instead, and the lists come into being only if somebody asks for them."""
    __slots__ = ("chrom", "n", "pos", "strand", "part_off", "part_pos", "part_site", "edge_cnt",
                 "comp_off", "comp_pos", "alpha", "_genes", "_strand_text", "gene_idx", "gene_names", "_tsv_static")

    def __init__(self):
        self._genes = self._strand_text = self.gene_idx = self.gene_names = None

    @property
    def genes(self):
        if self._genes is None and self.gene_idx is not None:
            names = self.gene_names
            self._genes = [names[i] if i >= 0 else "NA" for i in self.gene_idx.tolist()]
        return self._genes

    @genes.setter
    def genes(self, value):
        self._genes = value

    @property
    def strand_text(self):
        if self._strand_text is None and self.strand is not None:
            self._strand_text = [chr(c) if c else "" for c in self.strand.tolist()]
        return self._strand_text

    @strand_text.setter
    def strand_text(self, value):
        self._strand_text = value


class SiteTable(object):
    def __init__(self, gene_bins=None, is_stranded=False):
        self.bins = gene_bins if gene_bins is not None else GeneBins()
        self.is_stranded = bool(is_stranded)
        self.chrom_index = self.bins.chrom_index       # shared list: BED chromosomes are appended to it
        self.sites = {c: [] for c in self.chrom_index}  # site2D_array (:25), kept in Site.__lt__ order
        self._by_pos = {c: {} for c in self.chrom_index}
        self._exact = {}      # chrom -> its sites as the reference's list has them (see _find)
        self.assessed = self.created = self.assigned = 0

    # -- lookup with binary_site_search's acceptance rule (:198-207) -------------------------------
    # As long as no two sites of a chromosome can answer the same query, a dictionary by position says what the reference's
    # bisection finds.  They can when a query does not care about strand (an unstranded analysis, or a BED strand other than
    # '+' / '-') and a position holds more than one site: a line whose two ends coincide (both look-ups precede both
    # insertions, :291-292), or -- stranded -- a '?' line at a position that has its '+' and '-' site.  WHICH of them the
    # reference takes is where its bisection happens to land (:175-225), and where a new site goes among equals is where
    # bisect.insort under Site.__lt__ puts it (a comparison of '?' with '+' answers None there).  From the first line that can
    # make such a position on, the chromosome's sites are therefore kept as the reference keeps them -- a list in its order --
    # and looked up and inserted with its own steps (_array_search, _array_insort); tools/fuzz_reference.py holds that to the
    # reference itself.
    def _find(self, chrom, pos, strand):
        arr = self._exact.get(chrom)
        if arr is not None:
            k = _array_search(arr, pos, strand, self.is_stranded)
            return arr[k] if k >= 0 else None
        group = self._by_pos[chrom].get(pos)
        if not group:
            return None
        if _strand_free(strand, self.is_stranded):
            return group[0]
        for site in group:
            if site.strand == strand:
                return site
        return None

    def _insert(self, chrom, site):
        """bisect.insort under Site.__lt__ (Gene_Site_Iter_Graph_v0_1_8.py:123-136): ascending pos;
        in a stranded analysis '+' sorts before '-' at equal pos."""
        group = self._by_pos[chrom].setdefault(site.pos, [])
        group.append(site)
        arr = self._exact.get(chrom)
        if arr is not None:
            _array_insort(arr, site, self.is_stranded)
        elif self.is_stranded and len(group) > 1:
            group.sort(key=lambda s: _STRAND_RANK.get(s.strand, 2))  # stable

    def _keep_as_the_reference_does(self, chrom):
        """From here on this chromosome's sites live in a list in the reference's order (so far: ascending position, '+' before
        '-': what the groups hold)."""
        if chrom in self._exact or os.environ.get("SPL_SITES_FIRST_AT_POSITION"):   # (the switch: round 4's look-ups, for tools/fuzz_reference.py's comparison)
            return
        by_pos = self._by_pos[chrom]
        arr = []
        for pos in sorted(by_pos):
            arr.extend(by_pos[pos])
        self._exact[chrom] = arr

    def add_bed(self, bed_path, q_chrom="All", q_gene="All", max_intron=0):
        """findAlphaCounts for one sample."""
        bins = self.bins
        query = bins.query_gene
        if q_gene != "All" and query is None:
            raise ValueError("gene %r not found in the annotation (the reference dereferences None here, "
                             "SpliSER_v0_1_8.py:283)" % (q_gene,))
        max_intron = int(max_intron)
        with open(bed_path, "r") as handle:
            for line in handle:
                values = line.split("\t")
                if len(values) != 12:           # header / non-BED12 lines are skipped (:259)
                    continue
                chrom = values[0]
                if chrom not in self.sites:      # :265-268 (also for chromosomes the -c filter drops)
                    bins.ensure_chrom(chrom)
                    self.sites[chrom] = []
                    self._by_pos[chrom] = {}
                if not (q_chrom == chrom or q_chrom == "All"):
                    continue
                strand = values[5]
                flank = values[10].split(",")
                leftpos = int(values[1]) + int(flank[0])      # :275
                rightpos = int(values[2]) - int(flank[1])     # :276
                alpha = int(values[4])                         # :277
                if q_gene != "All":                            # :279-288
                    l_in = (leftpos + max_intron >= query.left) and (leftpos <= query.right)
                    r_in = (rightpos - max_intron <= query.right) and (rightpos >= query.left)
                    if not (l_in or r_in):
                        continue
                self.assessed += 2
                if leftpos == rightpos or (self.is_stranded and strand != "+" and strand != "-"):
                    self._keep_as_the_reference_does(chrom)      # (two sites may come to share a position for one query)
                # both look-ups happen before either insertion (:291-292)
                found = [self._find(chrom, leftpos, strand), self._find(chrom, rightpos, strand)]
                pair = []
                for pos, site in zip((leftpos, rightpos), found):
                    if site is None:
                        self.created += 1
                        gi = gene_search(bins.genes[chrom], pos, strand, self.is_stranded)   # :313
                        if gi >= 0:
                            gene = bins.genes[chrom][gi]
                            self.assigned += 1
                        else:
                            gene = NA_GENE
                        site = _Site(pos, strand, gene)
                        new = True
                    else:
                        new = False
                    site.alpha += alpha                        # :341
                    pair.append((site, new))
                for site, new in pair:                         # :347-350
                    if new:
                        self._insert(chrom, site)
                (left, _), (right, _) = pair
                for a, b in ((left, right), (right, left)):    # :352-355
                    if not any(self._same_site(b, q) for q in a.partner_sites):
                        a.partner_sites.append(b)
                    a.partner_counts[b.pos] = a.partner_counts.get(b.pos, 0) + alpha
        self._finalise_order()

    def _same_site(self, a, b):
        """Site.__eq__ (Gene_Site_Iter_Graph_v0_1_8.py:151-163)."""
        if a.pos != b.pos:
            return False
        if not self.is_stranded:
            return True
        return a.strand == b.strand

    def _finalise_order(self):
        for chrom, by_pos in self._by_pos.items():
            ordered = []
            if chrom in self._exact:
                ordered = list(self._exact[chrom])
            else:
                for pos in sorted(by_pos):
                    ordered.extend(by_pos[pos])
            for row, site in enumerate(ordered):
                site.row = row
            self.sites[chrom] = ordered

    def find_competitors(self):
        """findCompetitorPos: partners of my partners, other than me, as sorted unique positions."""
        for chrom in self.chrom_index:
            for site in self.sites.get(chrom, ()):
                found = set()
                for p in site.partner_sites:
                    for c in p.partner_sites:
                        if c.pos != site.pos:
                            found.add(c.pos)
                site.competitors = sorted(found)

    def n_sites(self):
        return sum(len(v) for v in self.sites.values())

    def chrom_arrays(self, chrom):
        """Flatten one chromosome to the arrays of ``spl_sites`` (all owned numpy arrays)."""
        sites = self.sites.get(chrom, [])
        n = len(sites)
        out = ChromArrays()
        out.chrom, out.n = chrom, n
        out.pos = np.fromiter((s.pos for s in sites), dtype=np.int64, count=n)
        out.strand = np.fromiter((ord(s.strand[0]) if s.strand else 0 for s in sites), dtype=np.uint8, count=n)
        out.alpha = np.fromiter((s.alpha for s in sites), dtype=np.int64, count=n)
        out.genes = [s.gene.name for s in sites]
        out.strand_text = [s.strand for s in sites]
        # A row's partner edges are its Partners LIST (the Site objects findBeta2Counts walks, :590), one edge each, with the count
        # PartnerCounts holds for the partner's position (:604).  Two partners at one position -- a '+' and a '?' site of a stranded
        # analysis, the two sites of a junction whose ends coincide -- are two edges with the same position and count: the TSV's
        # Partners column (a dict by position) shows one, the sums over partners take both.
        deg = np.fromiter((len(s.partner_sites) for s in sites), dtype=np.int64, count=n)
        out.part_off = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(deg, out=out.part_off[1:])
        n_part = int(out.part_off[-1])
        out.part_pos = np.empty(n_part, dtype=np.int64)
        out.part_site = np.empty(n_part, dtype=np.int32)
        out.edge_cnt = np.empty(n_part, dtype=np.int64)
        e = 0
        for s in sites:
            for p in s.partner_sites:
                out.part_pos[e] = p.pos
                out.part_site[e] = p.row
                out.edge_cnt[e] = s.partner_counts[p.pos]
                e += 1
        cdeg = np.fromiter((len(s.competitors) for s in sites), dtype=np.int64, count=n)
        out.comp_off = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(cdeg, out=out.comp_off[1:])
        out.comp_pos = np.fromiter((c for s in sites for c in s.competitors), dtype=np.int64,
                                   count=int(out.comp_off[-1]))
        return out
