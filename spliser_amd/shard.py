"""Shard planning: which chromosomes go to which GPU, and how they share one launch.

The hot path partitions by chromosome with no exchange step: a read only ever touches
sites of its own RNAME and findBeta2Counts only reads partner data of the same chromosome
(SpliSER_v0_1_8.py:590-604).  So

  * ``assign``  bin-packs chromosomes onto N devices by read count (longest-processing-time first);
  * ``pack``    lays the chromosomes of one device side by side in ONE int32 coordinate space, so the
                whole device share is a single kernel launch instead of one launch per chromosome
                (a 5-chromosome plant genome is five ~50 us launches otherwise).  Site and read
                coordinates of chromosome k are shifted by ``offset[k]``; offsets leave a gap wider
                than any read, so no read can reach a neighbour's sites.  When the total extent would
                exceed the int32 space (genomes > 2.1 Gbp) the device share is split into several
                shards = several launches.
"""
import numpy as np

from . import native

COORD_MAX = 2147483581
GAP = 1024


def assign(weights, n_devices):
    """LPT bin packing.  weights: {chrom: cost}.  -> list (len n_devices) of chrom lists."""
    bins = [[] for _ in range(max(1, n_devices))]
    load = [0] * len(bins)
    for chrom, w in sorted(weights.items(), key=lambda kv: (-kv[1], kv[0])):
        k = load.index(min(load))
        bins[k].append(chrom)
        load[k] += w
    return bins


class Shard(object):
    """Several chromosomes packed into one coordinate space."""

    def __init__(self):
        self.chroms = []       # chromosome names, in packing order
        self.offsets = []      # coordinate shift per chromosome
        self.limits = []       # largest coordinate (unshifted) the chromosome's slot has room for
        self.site_rows = []    # (row_begin, row_end) per chromosome in the packed table
        self.edge_rows = []    # (edge_begin, edge_end) per chromosome in the packed partner CSR
        self.sites = None      # native.SiteArrays
        self.reads = None      # native.ReadArrays of the whole shard (pack(..., concat_reads=True))
        self.read_segments = []  # [(native.ReadArrays of one chromosome, coordinate shift)]: Context.upload_read_segments


def _extent(arr, reads):
    lo, hi = 1, 1
    for a in (arr.pos, arr.part_pos, arr.comp_pos):
        if len(a):
            lo, hi = min(lo, int(a.min())), max(hi, int(a.max()))
    if reads is not None and reads.n:
        lo = min(lo, int(reads.pos.min()))
        hi = max(hi, int(reads.max_end), int(reads.pos.max()))
    return lo, hi


def pack(items, concat_reads=True, extents=None):
    """items: list of (chrom, ChromArrays, ReadSet-or-None).  -> list of Shard (usually one).

    ``concat_reads=False`` skips building the shard-wide read arrays on the host: the per-chromosome arrays stay where they
    are (e.g. in the BAM decoder's buffers) and ``read_segments`` tells the device where to put them.

    ``extents``: {chrom: (lo, hi)} coordinate range to reserve for a chromosome whose reads are not known yet (a BAM file
    still being decoded: the caller plans with the reference length of the header and checks the reads against it when
    they arrive); the sites are always looked at."""
    shards, cur, cursor = [], [], 0
    groups = []
    for chrom, arr, reads in items:
        lo, hi = _extent(arr, reads)
        if extents is not None and chrom in extents:
            lo, hi = min(lo, int(extents[chrom][0])), max(hi, int(extents[chrom][1]))
        span = hi - lo + 1 + GAP
        if span > COORD_MAX:
            raise native.SpliserNativeError(-6, "chromosome %s spans more than the int32 coordinate space" % chrom)
        if cur and cursor + span > COORD_MAX:
            groups.append(cur)
            cur, cursor = [], 0
        cur.append((chrom, arr, reads, cursor - lo + 1, hi))
        cursor += span
    if cur:
        groups.append(cur)
    for group in groups:
        sh = Shard()
        pos, strand, part_pos, part_site, comp_pos, alpha, edge_cnt = [], [], [], [], [], [], []
        part_deg, comp_deg = [], []
        r_pos, r_flag, r_ops, r_nops = [], [], [], []
        row = edge = 0
        for chrom, arr, reads, off, hi in group:
            sh.chroms.append(chrom)
            sh.offsets.append(off)
            sh.limits.append(hi)
            sh.site_rows.append((row, row + arr.n))
            n_edge = int(arr.part_off[-1]) if arr.n else 0
            sh.edge_rows.append((edge, edge + n_edge))
            pos.append(arr.pos + off)
            strand.append(arr.strand)
            part_pos.append(arr.part_pos + off)
            part_site.append(np.where(arr.part_site >= 0, arr.part_site + row, -1))
            comp_pos.append(arr.comp_pos + off)
            alpha.append(arr.alpha)
            edge_cnt.append(arr.edge_cnt)
            part_deg.append(np.diff(arr.part_off.astype(np.int64)))
            comp_deg.append(np.diff(arr.comp_off.astype(np.int64)))
            if reads is not None and reads.n:
                sh.read_segments.append((native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar), off))
                if concat_reads:
                    r_pos.append(reads.pos.astype(np.int64) + off)
                    r_flag.append(reads.flag)
                    r_ops.append(reads.cigar)
                    r_nops.append(np.diff(reads.cig_off.astype(np.int64)))
            row += arr.n
            edge += n_edge

        def cat(parts, dt):
            return np.concatenate(parts).astype(dt, copy=False) if parts else np.zeros(0, dt)

        def csr(degs):
            d = cat(degs, np.int64)
            off = np.zeros(d.shape[0] + 1, np.int64)
            np.cumsum(d, out=off[1:])
            if off[-1] > 0xfffffff0:
                raise native.SpliserNativeError(-6, "CSR too large for 32-bit offsets; use more shards")
            return off.astype(np.uint32)

        sh.sites = native.SiteArrays(cat(pos, np.int64), cat(strand, np.uint8), csr(part_deg), cat(part_pos, np.int64),
                                     csr(comp_deg), cat(comp_pos, np.int64), cat(part_site, np.int32),
                                     cat(alpha, np.int64), cat(edge_cnt, np.int64))
        if concat_reads:
            sh.reads = native.ReadArrays(cat(r_pos, np.int64), cat(r_flag, np.uint16), csr(r_nops), cat(r_ops, np.uint32))
        shards.append(sh)
    return shards
