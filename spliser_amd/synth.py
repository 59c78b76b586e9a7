"""Deterministic synthetic RNA-seq workloads: genomes with multi-isoform genes,
150 bp spliced reads as SoA, and the matching junction BED12 (+ GFF).

There is no network and the reference ships no data, so every BASELINE.json configuration is
synthesised here from a seed.  Reads are produced directly as the structure-of-arrays the GPU
consumes; ``write_inputs`` can also materialise BAM/SAM + BED + GFF files for end-to-end runs.

Site convention (SpliSER_v0_1_8.py:275-276, :482-483): for an intron between an exon ending at ``eA``
and one starting at ``sB`` (1-based, inclusive) the left site is ``eA`` and the right site ``sB - 1``.
"""
import os

import numpy as np

from . import samio

ARABIDOPSIS = [("Chr1", 30427671), ("Chr2", 19698289), ("Chr3", 23459830), ("Chr4", 18585056), ("Chr5", 26975502)]
HG38 = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
        ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
        ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
        ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
        ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]
MM39 = [("chr%d" % i, n) for i, n in enumerate(
    [195154279, 181755017, 159745316, 156860686, 151758149, 149588044, 144995196, 130127694, 124359700, 130530862,
     121973369, 120092757, 120883175, 125139656, 104073951, 98008968, 95294699, 90720763, 61420004], 1)] + \
    [("chrX", 169476592), ("chrY", 91455967)]

WORKLOADS = {
    # name: chroms, genes, reads, intron (median, min, max), exons mean, seed, paired/stranded flags
    "single_gene": dict(chroms=[("Chr1", 200000)], n_genes=12, n_reads=2000, intron=(100, 70, 6000), seed=1),
    "arabidopsis": dict(chroms=ARABIDOPSIS, n_genes=27000, n_reads=20_000_000, intron=(100, 70, 6000), seed=2),
    "human": dict(chroms=HG38, n_genes=36500, n_reads=200_000_000, intron=(1500, 70, 500000), seed=3),
    "mouse_stranded": dict(chroms=MM39, n_genes=28000, n_reads=100_000_000, intron=(1200, 70, 400000), seed=5, paired=True),
}

READ_LEN = 150
MAX_BLOCKS = 8  # a 150 bp read crosses at most MAX_BLOCKS - 1 introns (exons are >= 30 bp)


class Genome(object):
    """Genes with isoforms, flattened: isoform i owns exons [iso_off[i], iso_off[i+1])."""

    def __init__(self):
        self.chrom_names, self.chrom_lengths = [], []
        self.gene_chrom = self.gene_start = self.gene_end = self.gene_strand = None  # per gene
        self.gene_names = []
        self.iso_gene = self.iso_weight = self.iso_off = None                         # per isoform
        self.ex_start = self.ex_end = None                                             # per exon (1-based inclusive)


def make_genome(chroms, n_genes, intron=(100, 70, 6000), seed=0, exons_mean=5.0, alt_fraction=0.3, gene_prefix="G"):
    rng = np.random.default_rng(seed)
    g = Genome()
    g.chrom_names = [c for c, _ in chroms]
    g.chrom_lengths = [n for _, n in chroms]
    total = float(sum(g.chrom_lengths))
    med, imin, imax = intron
    gene_chrom, gene_start, gene_end, gene_strand, names = [], [], [], [], []
    iso_gene, iso_weight, iso_off, ex_s, ex_e = [], [], [0], [], []
    gid = 0
    for ci, (cname, clen) in enumerate(chroms):
        k = max(1, int(round(n_genes * clen / total)))
        # gene bodies first, then spread them over the chromosome with random gaps
        bodies = []
        for _ in range(k):
            n_ex = max(2, int(rng.poisson(exons_mean - 1)) + 1)
            ex_len = np.clip(rng.lognormal(np.log(170.0), 0.6, n_ex).astype(np.int64), 30, 3000)
            in_len = np.clip(rng.lognormal(np.log(med), 1.0, n_ex - 1).astype(np.int64), imin, imax)
            bodies.append((ex_len, in_len))
        span = sum(int(a.sum() + b.sum()) for a, b in bodies)
        free = max(clen - span - 2000, k * 50)
        gaps = rng.dirichlet(np.ones(k + 1)) * free
        cursor = 1000
        for gi, (ex_len, in_len) in enumerate(bodies):
            cursor += int(gaps[gi]) + 20
            starts = np.empty(len(ex_len), np.int64)
            pos = cursor
            for j in range(len(ex_len)):
                starts[j] = pos
                pos += int(ex_len[j]) + (int(in_len[j]) if j < len(in_len) else 0)
            ends = starts + ex_len - 1
            cursor = int(ends[-1]) + 1
            strand = "+" if rng.random() < 0.5 else "-"
            expr = float(rng.lognormal(0.0, 2.0))
            isoforms = [(starts, ends, 1.0)]
            n_ex = len(starts)
            if n_ex >= 3 and rng.random() < alt_fraction:      # exon skipping
                drop = int(rng.integers(1, n_ex - 1))
                keep = np.arange(n_ex) != drop
                isoforms.append((starts[keep], ends[keep], 0.25))
            if rng.random() < alt_fraction:                     # alternative 5'/3' site: move one exon edge
                j = int(rng.integers(0, n_ex - 1))
                s2, e2 = starts.copy(), ends.copy()
                shift = int(rng.integers(3, 31))
                if rng.random() < 0.5:
                    if e2[j] - shift - s2[j] >= 20:
                        e2[j] -= shift                            # alternative left (donor-side) site
                else:
                    if e2[j + 1] - (s2[j + 1] + shift) >= 20:
                        s2[j + 1] += shift                        # alternative right (acceptor-side) site
                isoforms.append((s2, e2, 0.2))
            gene_chrom.append(ci)
            gene_start.append(int(starts[0]))
            gene_end.append(int(ends[-1]))
            gene_strand.append(strand)
            names.append("%s%05d" % (gene_prefix, gid))
            for s_, e_, w in isoforms:
                iso_gene.append(gid)
                iso_weight.append(expr * w)
                ex_s.append(s_)
                ex_e.append(e_)
                iso_off.append(iso_off[-1] + len(s_))
            gid += 1
    g.gene_chrom = np.asarray(gene_chrom, np.int64)
    g.gene_start, g.gene_end = np.asarray(gene_start, np.int64), np.asarray(gene_end, np.int64)
    g.gene_strand = np.asarray([ord(s) for s in gene_strand], np.uint8)
    g.gene_names = names
    g.iso_gene = np.asarray(iso_gene, np.int64)
    g.iso_weight = np.asarray(iso_weight, np.float64)
    g.iso_off = np.asarray(iso_off, np.int64)
    g.ex_start, g.ex_end = np.concatenate(ex_s), np.concatenate(ex_e)
    return g


class ReadBatch(object):
    """Reads of all chromosomes, sorted by (chromosome, pos)."""
    __slots__ = ("chrom", "pos", "flag", "cig_off", "cigar", "junc_chrom", "junc_left", "junc_right", "junc_strand")


def _paired_flags(rng, n, gene_minus):
    """fr-firststrand-like flags: read 1 on the transcript strand, mate opposite (99/147 for '+', 83/163 for '-')."""
    first = rng.random(n) < 0.5
    flag = np.where(gene_minus, np.where(first, 83, 163), np.where(first, 99, 147)).astype(np.uint16)
    return flag


def expected_reads_per_chrom(genome, n_reads, read_len=READ_LEN, genomic_fraction=0.10):
    """What ``make_reads`` draws per chromosome ON AVERAGE (the isoform and gene weights it samples from, summed by chromosome):
    deterministic, no read is made.  What bench.py's strong-scaling plan cuts the sample by before any rank has generated it."""
    g = genome
    ex_len = g.ex_end - g.ex_start + 1
    csum = np.concatenate(([0], np.cumsum(ex_len)))
    iso_tlen = csum[g.iso_off[1:]] - csum[g.iso_off[:-1]]
    w = g.iso_weight * np.maximum(iso_tlen - read_len + 1, 0)
    n_chr = len(g.chrom_names)
    n_gen = int(round(n_reads * genomic_fraction))
    tx = np.bincount(g.gene_chrom[g.iso_gene], weights=w, minlength=n_chr)
    gw = np.bincount(g.iso_gene, weights=g.iso_weight, minlength=len(g.gene_start)) * np.maximum(g.gene_end - g.gene_start + 1, 1)
    gen = np.bincount(g.gene_chrom, weights=gw, minlength=n_chr)
    return (n_reads - n_gen) * tx / max(tx.sum(), 1e-300) + n_gen * gen / max(gen.sum(), 1e-300)


def strong_plan(expected, world):
    """ONE sample cut into ``world`` stretches of equal (expected) numbers of reads in file order -- chromosome after chromosome,
    a chromosome cut anywhere: what `process --gpus N` does to a BAM (spl_bam_share_plan), for bench.py's resident step.
    -> per rank [(chromosome index, f0, f1)]: the rank's piece of that chromosome is its reads [floor(f0 n), floor(f1 n)) of the n
    it really has.  Every rank makes the same plan; every read is in exactly one piece."""
    total = float(np.sum(expected))
    plan = [[] for _ in range(world)]
    if total <= 0:
        return plan
    at = 0.0
    for c, e in enumerate(expected):
        e = float(e)
        if e <= 0:
            continue
        for r in range(world):
            lo, hi = max(at, total * r / world), min(at + e, total * (r + 1) / world)
            if hi > lo:
                f0, f1 = (lo - at) / e, (hi - at) / e
                plan[r].append((c, 0.0 if lo <= at else f0, 1.0 if hi >= at + e else f1))
        at += e
    return plan


def make_reads(genome, n_reads, seed=0, read_len=READ_LEN, genomic_fraction=0.10, paired=False, stranded_single=False, keep_chroms=None):
    """Sample reads.  (1 - genomic_fraction) come from isoforms (spliced where they cross exon edges),
    the rest are unspliced genomic reads inside gene spans (pre-mRNA / retained introns -> beta1).

    ``keep_chroms`` (a set of chromosome indexes): the SAME sample's reads on those chromosomes only -- every random number is
    drawn as for the whole sample, the reads of other chromosomes are dropped before their CIGARs are made (most of the work):
    what a rank of a strong-scaling run generates of a sample it shares with the other ranks."""
    rng = np.random.default_rng(seed)
    g = genome
    ex_len = g.ex_end - g.ex_start + 1
    n_iso = len(g.iso_gene)
    ex_iso = np.repeat(np.arange(n_iso), np.diff(g.iso_off))
    csum = np.concatenate(([0], np.cumsum(ex_len)))
    iso_tlen = csum[g.iso_off[1:]] - csum[g.iso_off[:-1]]
    ex_tstart = csum[:-1] - csum[g.iso_off[:-1]][ex_iso]          # transcript offset of each exon's first base
    usable = np.maximum(iso_tlen - read_len + 1, 0)
    w = g.iso_weight * usable
    cw = np.cumsum(w)
    n_gen = int(round(n_reads * genomic_fraction))
    n_tx = n_reads - n_gen

    # ---- transcript reads
    iso = np.searchsorted(cw, rng.random(n_tx) * cw[-1], side="right")
    iso = np.minimum(iso, n_iso - 1)
    x = (rng.random(n_tx) * usable[iso]).astype(np.int64)           # transcript offset of the read start
    keep_tx = keep_gen = None
    if keep_chroms is not None:
        # (the genomic reads' and the flags' numbers are drawn further down from the same stream: drawn in full there too)
        in_set = np.zeros(len(g.chrom_names), bool)
        in_set[list(keep_chroms)] = True
        keep_tx = in_set[g.gene_chrom[g.iso_gene[iso]]]
        iso, x = iso[keep_tx], x[keep_tx]
        n_tx = int(keep_tx.sum())
    # first exon: largest e in the isoform with ex_tstart[e] <= x  (global searchsorted on a monotone key)
    big = int(iso_tlen.max()) + 1
    key_ex = ex_iso * big + ex_tstart
    e = np.searchsorted(key_ex, iso * big + x, side="right") - 1
    d0 = x - ex_tstart[e]
    pos = g.ex_start[e] + d0
    ops = np.zeros((n_tx, 2 * MAX_BLOCKS - 1), np.uint32)
    n_ops = np.zeros(n_tx, np.int64)
    jl = np.zeros((n_tx, MAX_BLOCKS - 1), np.int64)
    jr = np.zeros((n_tx, MAX_BLOCKS - 1), np.int64)
    remaining = np.full(n_tx, read_len, np.int64)
    avail = ex_len[e] - d0
    cur = e.copy()
    active = np.ones(n_tx, bool)
    for b in range(MAX_BLOCKS):
        take = np.minimum(avail, remaining)
        if b == MAX_BLOCKS - 1:
            take = np.where(active, remaining, take)                  # safety: dump the rest into the last block
        ops[active, 2 * b] = (take[active].astype(np.uint32) << 4)    # M
        n_ops[active] = 2 * b + 1
        remaining = remaining - np.where(active, take, 0)
        active = active & (remaining > 0)
        if b == MAX_BLOCKS - 1 or not active.any():
            break
        nxt = cur + 1
        idx = np.nonzero(active)[0]
        left = g.ex_end[cur[idx]]
        right = g.ex_start[nxt[idx]] - 1
        ops[idx, 2 * b + 1] = ((right - left).astype(np.uint32) << 4) | 3   # N
        jl[idx, b], jr[idx, b] = left, right
        cur = np.where(active, nxt, cur)
        avail = np.where(active, ex_len[cur], avail)
    gene = g.iso_gene[iso]

    # ---- genomic (unspliced) reads inside gene spans
    gw = np.bincount(g.iso_gene, weights=g.iso_weight, minlength=len(g.gene_start)) * np.maximum(g.gene_end - g.gene_start + 1, 1)
    cg = np.cumsum(gw)
    gg = np.minimum(np.searchsorted(cg, rng.random(n_gen) * cg[-1], side="right"), len(gw) - 1)
    gpos = g.gene_start[gg] - read_len // 2 + (rng.random(n_gen) * (g.gene_end[gg] - g.gene_start[gg] + 1)).astype(np.int64)
    gpos = np.maximum(gpos, 1)
    if keep_chroms is not None:
        keep_gen = in_set[g.gene_chrom[gg]]
        gg, gpos = gg[keep_gen], gpos[keep_gen]
        n_gen = int(keep_gen.sum())

    all_gene = np.concatenate((gene, gg))
    all_pos = np.concatenate((pos, gpos))
    all_nops = np.concatenate((n_ops, np.ones(n_gen, np.int64)))
    chrom = g.gene_chrom[all_gene]
    minus = g.gene_strand[all_gene] == ord("-")
    n = n_reads
    if keep_chroms is not None:     # the flags' random numbers for ALL reads of the sample, then this subset's
        keep_all = np.concatenate((keep_tx, keep_gen))
        n = int(keep_all.sum())
        if paired:
            first = (rng.random(n_reads) < 0.5)[keep_all]
            flag = np.where(minus, np.where(first, 83, 163), np.where(first, 99, 147)).astype(np.uint16)
        elif stranded_single:
            flag = np.where(minus, 16, 0).astype(np.uint16)
        else:
            flag = np.where((rng.random(n_reads) < 0.5)[keep_all], 16, 0).astype(np.uint16)
    elif paired:
        flag = _paired_flags(rng, n, minus)
    elif stranded_single:
        flag = np.where(minus, 16, 0).astype(np.uint16)
    else:
        flag = np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16)
    order = np.lexsort((all_pos, chrom))
    # ragged CIGAR in sorted order
    flat_tx = ops.reshape(-1)
    col = np.arange(ops.shape[1])[None, :]
    valid_tx = (col < n_ops[:, None]).reshape(-1)
    tx_ops = flat_tx[valid_tx]
    gen_ops = np.full(n_gen, (read_len << 4), np.uint32)
    all_ops = np.concatenate((tx_ops, gen_ops))
    src_off = np.zeros(n + 1, np.int64)
    np.cumsum(all_nops, out=src_off[1:])
    nops_sorted = all_nops[order]
    dst_off = np.zeros(n + 1, np.int64)
    np.cumsum(nops_sorted, out=dst_off[1:])
    # gather: for each destination op slot, its source slot
    rep = np.repeat(src_off[:-1][order] - dst_off[:-1], nops_sorted)
    cigar = all_ops[np.arange(dst_off[-1]) + rep]

    out = ReadBatch()
    out.chrom = chrom[order]
    out.pos = all_pos[order]
    out.flag = flag[order]
    out.cig_off = dst_off
    out.cigar = cigar
    # junction usage (what regtools/tophat would put in the BED): one entry per N op
    jmask = jr > 0
    rows = np.nonzero(jmask)[0]
    out.junc_chrom = g.gene_chrom[gene[rows]]
    out.junc_left = jl[jmask]
    out.junc_right = jr[jmask]
    out.junc_strand = g.gene_strand[gene[rows]]
    return out


def split_by_chrom(batch, n_chroms):
    """-> list of samio.ReadSet (one per chromosome index; empty ones included)."""
    bounds = np.searchsorted(batch.chrom, np.arange(n_chroms + 1), side="left")
    sets = []
    for c in range(n_chroms):
        a, b = int(bounds[c]), int(bounds[c + 1])
        off = batch.cig_off[a:b + 1] - batch.cig_off[a]
        sets.append(samio.ReadSet(batch.pos[a:b], batch.flag[a:b], off, batch.cigar[batch.cig_off[a]:batch.cig_off[b]]))
    return sets


def junction_table(batches, min_count=1):
    """Aggregate junction usage over read batches -> arrays (chrom, left, right, strand, count), sorted."""
    chrom = np.concatenate([b.junc_chrom for b in batches])
    left = np.concatenate([b.junc_left for b in batches])
    right = np.concatenate([b.junc_right for b in batches])
    strand = np.concatenate([b.junc_strand for b in batches])
    if len(chrom) == 0:
        z = np.zeros(0, np.int64)
        return z, z, z, np.zeros(0, np.uint8), z
    # pack (chrom, left, right - left, strand) into one int64 key: 6 + 29 + 27 + 1 bits
    span = right - left
    key = (((chrom << 29 | left) << 27 | span) << 1) | (strand == ord("-")).astype(np.int64)
    uniq, first, counts = np.unique(key, return_index=True, return_counts=True)
    keep = counts >= min_count
    first, counts = first[keep], counts[keep]
    return chrom[first], left[first], right[first], strand[first], counts


def write_bed(path, chrom_names, junctions, overhang=20, stranded=True):
    chrom, left, right, strand, count = junctions
    with open(path, "w") as fh:
        fh.write('track name=junctions description="synthetic junctions"\n')
        for i in range(len(chrom)):
            a = b = overhang
            start, end = int(left[i]) - a, int(right[i]) + b
            st = chr(int(strand[i])) if stranded else "?"
            fh.write("%s\t%d\t%d\tJUNC%08d\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
                chrom_names[int(chrom[i])], start, end, i + 1, int(count[i]), st, start, end, a, b, end - start - b))


def write_gff(path, genome):
    with open(path, "w") as fh:
        fh.write("##gff-version 3\n")
        for i, name in enumerate(genome.gene_names):
            fh.write("%s\tsynth\tgene\t%d\t%d\t.\t%s\t.\tID=%s;Name=%s\n" % (
                genome.chrom_names[int(genome.gene_chrom[i])], int(genome.gene_start[i]), int(genome.gene_end[i]),
                chr(int(genome.gene_strand[i])), name, name))


_POOL_STATE = None


class _Junctions(object):
    __slots__ = ("junc_chrom", "junc_left", "junc_right", "junc_strand")


def _batch_job(job):
    m, seed = job
    genome, paired, nchr, keep = _POOL_STATE
    rb = make_reads(genome, m, seed=seed, paired=paired, keep_chroms=keep)
    j = _Junctions()
    j.junc_chrom, j.junc_left, j.junc_right, j.junc_strand = rb.junc_chrom, rb.junc_left, rb.junc_right, rb.junc_strand
    return split_by_chrom(rb, nchr), j


class Workload(object):
    """A genome + reads per chromosome + the site table inputs derived from them."""

    def __init__(self, name, n_reads=None, seed=None, scale=1.0, batch=2_500_000, workers=None, genome=None, read_seed=None,
                 silence=0.0, keep_chroms=None, **over):
        """``workers`` processes (fork) generate ``batch``-read slices in parallel; call this BEFORE the
        process touches the GPU (a forked child must not inherit an initialised HIP runtime).

        ``genome`` / ``read_seed`` / ``silence``: another SAMPLE of a genome that exists already (BASELINE config 4: six samples of
        one genome, seeds 11-16, 15 % sample-specific junctions) -- the reads drawn with ``read_seed``, and a fraction
        ``silence`` of the isoforms, picked by that seed, not expressed in this sample: their junctions are what the other
        samples have and this one does not (`combine` fills those gaps from this sample's BAM).

        ``keep_chroms`` (chromosome indexes): only those chromosomes' reads (and junctions) of the sample are made -- the same
        reads the whole sample has there (``make_reads``); the other chromosomes stay empty."""
        if workers is None:
            workers = min(8, os.cpu_count() or 1)
        cfg = dict(WORKLOADS[name])
        cfg.update(over)
        self.name = name
        self.seed = cfg["seed"] if seed is None else seed
        n_reads = int((cfg["n_reads"] if n_reads is None else n_reads) * scale)
        n_genes = max(2, int(cfg["n_genes"] * (scale if scale < 1.0 else 1.0)))
        self.paired = bool(cfg.get("paired"))
        self.genome = genome if genome is not None else make_genome(cfg["chroms"], n_genes, cfg["intron"], seed=self.seed,
                                                                    alt_fraction=cfg.get("alt_fraction", 0.3))
        rseed = self.seed if read_seed is None else int(read_seed)
        expressed = self.genome
        if silence > 0.0:
            import copy
            expressed = copy.copy(self.genome)
            off = np.random.default_rng(rseed * 7919 + 13).random(len(expressed.iso_weight)) < silence
            expressed.iso_weight = np.where(off, 0.0, expressed.iso_weight)
        nchr = len(self.genome.chrom_names)
        per_chrom = [[] for _ in range(nchr)]
        jobs = []
        done = k = 0
        while done < n_reads:
            m = min(batch, n_reads - done)
            jobs.append((m, rseed * 1000 + k))
            done += m
            k += 1
        global _POOL_STATE
        _POOL_STATE = (expressed, self.paired, nchr, None if keep_chroms is None else set(int(c) for c in keep_chroms))
        if workers > 1 and len(jobs) > 1:
            import multiprocessing
            with multiprocessing.get_context("fork").Pool(min(workers, len(jobs))) as pool:
                results = pool.map(_batch_job, jobs)
        else:
            results = [_batch_job(j) for j in jobs]
        _POOL_STATE = None
        batches = []
        for sets, junc in results:
            for c, rs in enumerate(sets):
                per_chrom[c].append(rs)
            batches.append(junc)
        self.reads = [_merge_sorted(parts) for parts in per_chrom]
        self.junctions = junction_table(batches)
        self.n_reads = n_reads

    # -- on-disk cache (profiling runs must not fork generator workers under rocprofv3) ----------------
    def save(self, path):
        arrays = {"junc%d" % i: a for i, a in enumerate(self.junctions)}
        for c, rs in enumerate(self.reads):
            arrays.update({"pos%d" % c: rs.pos, "flag%d" % c: rs.flag, "off%d" % c: rs.cig_off, "cig%d" % c: rs.cigar,
                           "end%d" % c: np.asarray([rs.max_end])})
        arrays["chrom_lengths"] = np.asarray(self.genome.chrom_lengths, np.int64)
        arrays["chrom_names"] = np.asarray(self.genome.chrom_names)
        arrays["meta"] = np.asarray([self.n_reads, self.seed, int(self.paired)], np.int64)
        np.savez(path, **arrays)

    @classmethod
    def load(cls, path, name):
        z = np.load(path)
        self = cls.__new__(cls)
        self.name = name
        self.n_reads, self.seed, paired = (int(v) for v in z["meta"])
        self.paired = bool(paired)
        self.genome = Genome()
        self.genome.chrom_names = [str(c) for c in z["chrom_names"]]
        self.genome.chrom_lengths = [int(v) for v in z["chrom_lengths"]]
        self.junctions = tuple(z["junc%d" % i] for i in range(5))
        self.reads = [samio.ReadSet(z["pos%d" % c], z["flag%d" % c], z["off%d" % c], z["cig%d" % c], int(z["end%d" % c][0]))
                      for c in range(len(self.genome.chrom_names))]
        return self

    def write_inputs(self, prefix, bam=True, sam=False, gff=True, level=1):
        names, lens = self.genome.chrom_names, self.genome.chrom_lengths
        write_bed(prefix + ".bed", names, self.junctions)
        if gff:
            write_gff(prefix + ".gff", self.genome)
        pairs = [(names[c], self.reads[c]) for c in range(len(names))]
        if bam:
            samio.write_bam(prefix + ".bam", names, lens, pairs, level=level, with_seq=True)
        if sam:
            samio.write_sam(prefix + ".sam", names, lens, pairs)
        return prefix


def add_soft_clips(reads, fraction, seed=0):
    """A copy of ``reads`` in which ``fraction`` of the records carry a leading and / or trailing soft clip (the aligned part
    and POS stay as they are: what a local aligner reports for reads with adapter or low-quality ends)."""
    from .samio import ReadSet
    rng = np.random.default_rng(seed)
    n = reads.n
    pick = rng.random(n) < fraction
    side = rng.integers(1, 4, n)                      # 1 leading, 2 trailing, 3 both
    lead = (pick & ((side & 1) != 0)).astype(np.int64)
    trail = (pick & ((side & 2) != 0)).astype(np.int64)
    off = reads.cig_off.astype(np.int64)
    nops = np.diff(off)
    new_off = np.zeros(n + 1, np.int64)
    np.cumsum(nops + lead + trail, out=new_off[1:])
    out = np.empty(int(new_off[-1]), np.uint32)
    # original ops, shifted behind the leading clip of their read
    owner = np.repeat(np.arange(n), nops)
    within = np.arange(int(off[-1])) - np.repeat(off[:-1], nops)
    out[new_off[:-1][owner] + lead[owner] + within] = reads.cigar
    clip = ((rng.integers(1, 25, n).astype(np.uint32)) << 4) | 4
    out[new_off[:-1][lead == 1]] = clip[lead == 1]
    out[new_off[1:][trail == 1] - 1] = clip[trail == 1]
    return ReadSet(reads.pos, reads.flag, new_off, out, reads.max_end)


def _merge_sorted(parts):
    """Merge per-batch ReadSets of one chromosome into one coordinate-sorted ReadSet."""
    parts = [p for p in parts if p.n]
    if not parts:
        return samio.ReadSet.empty()
    if len(parts) == 1:
        return parts[0]
    pos = np.concatenate([p.pos for p in parts])
    flag = np.concatenate([p.flag for p in parts])
    nops = np.concatenate([np.diff(p.cig_off.astype(np.int64)) for p in parts])
    ops = np.concatenate([p.cigar for p in parts])
    order = np.argsort(pos, kind="stable")
    src_off = np.zeros(len(pos) + 1, np.int64)
    np.cumsum(nops, out=src_off[1:])
    nops_s = nops[order]
    dst_off = np.zeros(len(pos) + 1, np.int64)
    np.cumsum(nops_s, out=dst_off[1:])
    rep = np.repeat(src_off[:-1][order] - dst_off[:-1], nops_s)
    cigar = ops[np.arange(dst_off[-1]) + rep]
    return samio.ReadSet(pos[order], flag[order], dst_off, cigar, max_end=max(p.max_end for p in parts))
