"""``combine`` -- merge per-sample ``.SpliSER.tsv`` files and fill the gaps from each sample's BAM.

Reference: ``combine`` SpliSER_v0_1_8.py:742-917, ``outputCombinedLines`` :722-740, region ordering through
``Graph.topologicalSort`` Gene_Site_Iter_Graph_v0_1_8.py:358-396.

The reference walks all sample files in lock-step and, for every site a sample does not list, calls ``checkBam`` on
that sample's BAM (one ``samtools view`` per missing (site, sample), :903).  Here the walk is done once on the host
without touching any BAM and emits the missing (site, sample) pairs as *queries*; each sample's BAM is then decoded
once and all of its queries are answered by ONE kernel launch per shard in ``combine_mode`` (a flanking read counts
toward beta2Simple, :529-536).  Query tables have one-way partner lists (no mutual links), so the library answers them
with its pair kernel -- same counters.

Order dependence that is kept: the gap-fill of sample ``idx`` sees the strand, partners and competitors
contributed only by samples with a LOWER index that list the site (:869-904); a stranded gap-fill with no such sample
has site strand '' and therefore counts nothing.
"""
import os
import sys
import threading
import time
from ast import literal_eval

import numpy as np

from . import native

HEADER = ("Sample\tRegion\tSite\tStrand\tGene\tSSE\talpha_count\tbeta1_count\tbeta2Simple_count\tbeta2Cryptic_count\t"
          "beta2_weighted\tPartners\tCompetitors\n")


def _log(msg):
    print(msg)
    sys.stdout.flush()


def read_samples_file(path, strict=True):
    """Three tab-separated columns per line: title, .SpliSER.tsv path, BAM path (:750-759).  ``combine`` rejects
    any other line; ``combineShallow`` skips it (:929-935)."""
    titles, tsvs, bams = [], [], []
    with open(path, "r") as fh:
        for line in fh:
            values = line.split("\t")
            if len(values) == 3:
                titles.append(values[0])
                tsvs.append(values[1])
                bams.append(values[2].rstrip())
            elif strict:
                raise Exception("Samples File contains lines that do not have exactly 3 tab-separated columns")
    return titles, tsvs, bams


class _Row(object):
    __slots__ = ("chrom", "pos", "strand", "gene", "sse", "alpha", "beta1", "b2s", "b2c", "b2w", "partners", "competitors")


def _parse_tsv(path):
    rows = []
    with open(path, "r") as fh:
        for i, line in enumerate(fh):
            if i == 0:
                continue
            v = line.rstrip().split("\t")
            r = _Row()
            r.chrom, r.pos, r.strand, r.gene = v[0], int(v[1]), v[2], v[3]
            r.sse = float(v[4])
            r.alpha, r.beta1, r.b2s = int(v[5]), int(v[6]), int(v[7])
            r.b2c = None if v[8] == "NA" else int(v[8])
            r.b2w = None if v[8] == "NA" else float(v[9])
            r.partners = literal_eval(v[10])
            r.competitors = literal_eval(v[11])
            rows.append(r)
    return rows


def region_order(per_sample_rows):
    """Deduce one order of regions consistent with every file (:761-790): edges between consecutive regions of each
    file (from an artificial first region), depth-first topological sort in the reference's visiting order."""
    runs = []
    for rows in per_sample_rows:
        prev, mine = "-1", []
        for r in rows:
            if r.chrom != prev:
                mine.append(r.chrom)
                prev = r.chrom
        runs.append(mine)
    return region_order_from_runs(runs)


def region_order_from_runs(runs):
    """``region_order`` from what it reads of a file: its regions in order, one entry per run of lines."""
    nodes, before, after = [], [], []
    for mine in runs:
        prev = "-1"
        for chrom in mine:
            before.append(prev)
            after.append(chrom)
            prev = chrom
            if chrom not in nodes:
                nodes.insert(0, chrom)
    if not nodes:
        return []
    nodes.insert(0, "-1")
    adj = {}
    for b, a in zip(before, after):
        lst = adj.setdefault(b, [])
        if a not in lst:
            lst.append(a)
    visited, order = set(), []

    def visit(n):            # recursive like the reference; region counts are small
        visited.add(n)
        for m in adj.get(n, ()):
            if m not in visited:
                visit(m)
        order.insert(0, n)
    for n in nodes:
        if n not in visited:
            visit(n)
    return order[1:]


class _Merged(object):
    __slots__ = ("chrom", "pos", "strand", "gene", "has_row", "alpha", "beta1", "b2s", "b2c", "b2w", "partner_keys",
                 "partner_counts", "competitors", "queries")


def merge_sites(per_sample_rows, chroms, n_samples, is_stranded, q_gene, shallow=None, log=None):
    """The lock-step walk of ``combine`` (:820-915) -- or, with ``shallow = (minSamples, minReads, minSSE)``, of
    ``combineShallow`` (:1007-1166) -- without the BAM access.  -> list of _Merged in output order."""
    cursor = [0] * n_samples
    out = []
    for chrom in chroms:
        while True:
            lowest, lowest_strand, gene, seen = -1, "?", "", 0
            for idx in range(n_samples):
                rows = per_sample_rows[idx]
                if cursor[idx] < len(rows) and rows[cursor[idx]].chrom == chrom:
                    r = rows[cursor[idx]]
                    if shallow is None:
                        if r.pos < lowest or lowest == -1 or (is_stranded and r.pos == lowest and r.strand == "+"):   # :847
                            lowest, lowest_strand, gene = r.pos, r.strand, r.gene
                    else:
                        good = (r.alpha + r.beta1 + r.b2s) >= shallow[1] and r.sse >= shallow[2]
                        if r.pos < lowest or lowest == -1 or (is_stranded and r.pos == lowest and r.strand != lowest_strand and r.strand == "+"):   # :1066
                            lowest, lowest_strand, gene = r.pos, r.strand, r.gene
                            seen = 1 if good else 0      # a new lowest position restarts the tally (:1071-1077)
                        elif r.pos == lowest and good:   # whatever the strand (:1079-1084)
                            seen += 1
            if lowest == -1:
                break
            if shallow is not None and seen < shallow[0]:
                # not enough samples with evidence: drop it from every file whose next line has this POSITION -- the
                # reference compares the number only, not region or strand (:1150-1156)
                if log:
                    log("Skipped site {} for insufficient evidence, only {} samples with Site using minimum reads".format(lowest, seen))
                for idx in range(n_samples):
                    rows = per_sample_rows[idx]
                    if cursor[idx] < len(rows) and rows[cursor[idx]].pos == lowest:
                        cursor[idx] += 1
                continue
            m = _Merged()
            m.chrom, m.pos, m.strand, m.gene = chrom, lowest, "", gene
            m.has_row = [False] * n_samples
            m.alpha, m.beta1, m.b2s, m.b2c = [0] * n_samples, [0] * n_samples, [0] * n_samples, [0] * n_samples
            m.b2w = [0.0] * n_samples
            m.partner_keys, m.partner_counts, m.competitors, m.queries = [], {}, [], []
            wanted = (q_gene == "All" or q_gene == gene)
            for idx in range(n_samples):
                rows = per_sample_rows[idx]
                r = rows[cursor[idx]] if cursor[idx] < len(rows) else None
                if r is not None and r.chrom == chrom and r.pos == lowest and (not is_stranded or r.strand == lowest_strand):   # :870
                    cursor[idx] += 1
                    m.has_row[idx] = True
                    m.strand = str(r.strand)
                    m.alpha[idx] += r.alpha
                    m.beta1[idx] += r.beta1
                    m.b2s[idx] += r.b2s
                    if r.b2c is not None:
                        m.b2c[idx] += r.b2c
                        m.b2w[idx] += r.b2w
                    for key, val in r.partners.items():
                        if key not in m.partner_counts:
                            m.partner_counts[key] = [0] * n_samples
                            m.partner_keys.append(key)
                        m.partner_counts[key][idx] += val
                    for c in r.competitors:
                        if c not in m.competitors:
                            m.competitors.append(int(c))
                            m.competitors.sort()
                elif wanted:
                    # gap: checkBam on this sample's BAM with the site as it stands NOW (:899-904)
                    m.queries.append((idx, m.strand, list(m.partner_keys), list(m.competitors)))
            if wanted:
                out.append(m)
    return out


def _sse(alpha, beta1, b2s, b2w, cryptic):
    """calculateSSE (:626-639) for one sample."""
    betas = beta1 + b2s
    if cryptic:
        den = alpha + (betas + b2w)
    else:
        den = alpha + betas
    return (alpha / den) if den > 0.0 else 0.0


def gap_queries(merged):
    """{sample idx: [(site index, chrom, pos, strand, partner keys, competitors), ...]}"""
    per_sample = {}
    for si, m in enumerate(merged):
        for (idx, strand, pkeys, comps) in m.queries:
            per_sample.setdefault(idx, []).append((si, m.chrom, m.pos, strand, pkeys, comps))
    return per_sample


class _QueryTable(object):
    """One sample's gap-fill queries in the shape ``process.process_sites`` counts: a table per chromosome (rows sorted by
    position; strand, partners and competitors of a row as the walk had them when it reached that sample), and for every row
    the index of the merged site it answers."""

    def __init__(self, queries):
        by_chrom = {}
        for q in queries:
            by_chrom.setdefault(q[1], []).append(q)
        self.chrom_index = list(by_chrom)
        self._arrays, self.site_index = {}, {}
        for chrom, qs in by_chrom.items():
            qs.sort(key=lambda q: q[2])
            self._arrays[chrom] = _query_arrays(chrom, qs)
            self.site_index[chrom] = [q[0] for q in qs]

    def chrom_arrays(self, chrom):
        return self._arrays[chrom]


def fill_gaps(merged, bam_paths, is_stranded, stranded_type, devices=(0,), threads=0, log=_log, kept_reads=None):
    """Answer every (site, sample) query of ``merge_sites``'s result on the GPUs (``fill_tables``).
    -> {(site index, sample idx): (beta1, beta2Simple)}"""
    tables = {idx: _QueryTable(qs) for idx, qs in gap_queries(merged).items()}
    results = {}
    lock = threading.Lock()

    def take(idx, site, beta1, b2):
        with lock:
            for si, x, y in zip(site, beta1.tolist(), b2.tolist()):
                results[(int(si), idx)] = (int(x), int(y))
    fill_tables(tables, take, bam_paths, is_stranded, stranded_type, devices=devices, threads=threads, log=log, kept_reads=kept_reads)
    return results


def open_kept_reads(kept_reads, bam_paths):
    """The samples' kept reads (``process --keepReads``), opened on a few threads while the caller does something else (the walk
    over the sample files does not need them; opening one maps 0.4 GB and checks it against its checksum): -> a function
    ``idx -> ReadStore or None`` that waits for that sample's.  Samples without such a file, or with a stale one, give None."""
    from concurrent.futures import ThreadPoolExecutor
    from . import readstore
    if not kept_reads:
        return lambda idx: None
    pool = ThreadPoolExecutor(max_workers=3)
    futures = {idx: pool.submit(readstore.open_if_fresh, path, bam_paths[idx]) for idx, path in enumerate(kept_reads) if path}
    pool.shutdown(wait=False)

    def get(idx):
        f = futures.pop(idx, None)
        return f.result() if f is not None else None
    get.close = lambda: [f.result().close() for f in list(futures.values()) if f.result() is not None] and futures.clear()   # (what nobody took)
    return get


def fill_tables(tables, take, bam_paths, is_stranded, stranded_type, devices=(0,), threads=0, log=_log, kept_reads=None, opened=None):
    """Answer the query tables ``{sample idx: table}`` on the GPUs.  Each sample's BAM is decoded once, in the background (on
    the GPU when the call has one device, like ``process``); its chromosomes are dealt to the devices
    (``process.process_sites``: one context per device, a chromosome goes to its GPU as soon as the decoder has it complete) and
    counted in ``combine_mode`` (a flanking read counts toward beta2Simple, :529-536); several samples are in flight at a time,
    each starting on another device.  ``take(idx, merged-site indexes, beta1, beta2Simple)`` gets a region's answers."""
    from concurrent.futures import ThreadPoolExecutor
    from . import process as _process, readstore
    devices = tuple(devices)

    def one(idx):
        table = tables[idx]
        devs = devices[idx % len(devices):] + devices[:idx % len(devices)]
        # what `process --keepReads` left of this very BAM (readstore: keyed by the BAM's size, time and edges), or the BAM again
        if opened is not None:
            source = opened(idx)          # (opened while the files were walked: open_kept_reads)
        else:
            source = readstore.open_if_fresh(kept_reads[idx], bam_paths[idx]) if kept_reads and kept_reads[idx] else None
        if source is not None:
            log("  ({}: reads kept by process, {} not decoded again)".format(os.path.basename(kept_reads[idx]), os.path.basename(bam_paths[idx])))
        else:
            source = _process.open_and_decode(bam_paths[idx], devs, None, threads)   # (on the sample's first device when it has one device)
        try:
            out = _process.process_sites(table, source, "All", is_stranded, stranded_type, False, devices=devs, combine_mode=1,
                                         log=lambda m: None)
            if isinstance(source, native.BamFile) and not source.wait_all():   # not sorted by reference: again, from the whole decode
                out = _process.process_sites(table, source, "All", is_stranded, stranded_type, False, devices=devs, combine_mode=1,
                                             log=lambda m: None)
        finally:
            if hasattr(source, "close"):
                source.close()
        for chrom, (_, r) in out.items():
            take(idx, table.site_index[chrom], r["beta1"], r["beta2s_reads"])

    order = sorted(tables)
    n_flight = max(2, len(devices))       # samples decoded and counted side by side
    if os.environ.get("SPL_COMBINE_WORKERS"):
        n_flight = max(1, int(os.environ["SPL_COMBINE_WORKERS"]))
    with ThreadPoolExecutor(max_workers=max(1, min(len(order), n_flight))) as pool:
        for _ in pool.map(one, order):
            pass


class _NativeQueryTable(object):
    """``_QueryTable`` from the tables of the native walk (``native.Combine.tables``)."""

    def __init__(self, tabs):
        self.chrom_index = [c for c, _ in tabs]
        self._arrays, self.site_index = {}, {}
        for chrom, t in tabs:
            a = _QueryArrays()
            a.chrom, a.n = chrom, int(t["pos"].shape[0])
            a.pos, a.strand = t["pos"], t["strand"]
            a.part_off, a.part_pos, a.comp_off, a.comp_pos = t["part_off"], t["part_pos"], t["comp_off"], t["comp_pos"]
            a.part_site = np.full(a.part_pos.shape[0], -1, np.int32)
            a.edge_cnt = np.zeros(a.part_pos.shape[0], np.int64)
            a.alpha = np.zeros(a.n, np.int64)
            self._arrays[chrom] = a
            self.site_index[chrom] = t["site"]

    def chrom_arrays(self, chrom):
        return self._arrays[chrom]


class _QueryArrays(object):
    __slots__ = ("chrom", "n", "pos", "strand", "part_off", "part_pos", "part_site", "edge_cnt", "comp_off", "comp_pos", "alpha")


def _query_arrays(chrom, qs):
    a = _QueryArrays()
    a.chrom, a.n = chrom, len(qs)
    a.pos = np.array([q[2] for q in qs], np.int64)
    a.strand = np.array([ord(q[3][0]) if q[3] else 0 for q in qs], np.uint8)
    pdeg = np.array([len(q[4]) for q in qs], np.int64)
    cdeg = np.array([len(q[5]) for q in qs], np.int64)
    a.part_off = np.zeros(a.n + 1, np.uint32)
    a.comp_off = np.zeros(a.n + 1, np.uint32)
    np.cumsum(pdeg, out=a.part_off[1:])
    np.cumsum(cdeg, out=a.comp_off[1:])
    a.part_pos = np.array([p for q in qs for p in q[4]], np.int64)
    a.comp_pos = np.array([c for q in qs for c in q[5]], np.int64)
    a.part_site = np.full(a.part_pos.shape[0], -1, np.int32)     # one-way lists (the range kernel takes them: its junction table is built per row)
    a.edge_cnt = np.zeros(a.part_pos.shape[0], np.int64)
    a.alpha = np.zeros(a.n, np.int64)
    return a


def write_combined(path, merged, titles, results, cryptic):
    """outputCombinedLines (:722-740)."""
    with open(path, "w") as fh:
        fh.write(HEADER)
        for si, m in enumerate(merged):
            comp_txt = "[" + ", ".join(str(c) for c in m.competitors) + "]"
            for idx, title in enumerate(titles):
                if m.has_row[idx]:
                    alpha, beta1, b2s = m.alpha[idx], m.beta1[idx], m.b2s[idx]
                    sse = _sse(alpha, beta1, b2s, m.b2w[idx], cryptic)
                else:
                    alpha = 0
                    beta1, b2s = results.get((si, idx), (0, 0))
                    sse = 0.0
                mid = ("%d\t%s" % (m.b2c[idx], str(m.b2w[idx]))) if cryptic else "NA\tNA"
                part_txt = "{" + ", ".join("%d: %d" % (k, m.partner_counts[k][idx]) for k in m.partner_keys) + "}"
                fh.write("%s\t%s\t%d\t%s\t%s\t%s\t%d\t%d\t%d\t%s\t%s\t%s\n" % (
                    title, m.chrom, m.pos, m.strand, m.gene, "{0:.3f}".format(sse), alpha, beta1, b2s, mid, part_txt, comp_txt))


def combine(samplesFile, outputPath, qGene="All", isStranded=False, strandedType="fr", isbeta2Cryptic=False,
            devices=(0,), threads=0, log=_log, shallow=None, native_walk=None):
    """-> timings {parse_s, merge_s, gapfill_s, write_s, total_s, walk: "native" | "python", sites, gap_sites, queries}.

    The walk over the files runs on columns in the native library (``native.Combine``); files its parsers do not take -- and
    ``native_walk=False`` or SPL_COMBINE_PYTHON=1 -- are walked by the Python statement of the same loop below."""
    t_all = time.perf_counter()
    from . import process as _proc
    _proc.wait_deferred_close()       # (`process --keepReads` calls of this very interpreter may still be writing what this call is about to look for)
    log("Combining samples...")
    titles, tsvs, bams = read_samples_file(samplesFile, strict=shallow is None)
    if native_walk is None:
        native_walk = not os.environ.get("SPL_COMBINE_PYTHON")
    # what `process --keepReads` may have left beside each sample's .SpliSER.tsv (taken only if it is still its BAM's: readstore)
    from . import readstore
    kept = None if os.environ.get("SPL_IGNORE_KEPT_READS") else [readstore.path_for_tsv(p) for p in tsvs]
    tm = None
    if native_walk:
        # Only the OPENING of the sample files decides between the two walks (-5 there: a file that is not the plain text `process`
        # writes -- Python reads it its own way).  -5 is also what a damaged BAM gives during the gap fill: that is the run's
        # error, not a reason to parse, merge and fill everything a second time in Python and fail again.
        walk = None
        try:
            t_open = time.perf_counter()
            if qGene is not None and any(ord(ch) > 127 for ch in str(qGene)):
                raise native.SpliserNativeError(-5, "a gene name outside ASCII is the Python walk's")
            walk = native.Combine(tsvs)
        except native.SpliserNativeError as exc:
            if exc.code != -5:
                raise
        if walk is not None:
            tm = _combine_native(walk, time.perf_counter() - t_open, titles, bams, outputPath, qGene, isStranded, strandedType, isbeta2Cryptic, devices, threads, log, shallow, kept)
    if tm is None:
        tm = _combine_python(titles, tsvs, bams, outputPath, qGene, isStranded, strandedType, isbeta2Cryptic, devices, threads, log, shallow, kept)
    if tm:
        tm["total_s"] = time.perf_counter() - t_all
    return tm


def _combine_native(walk, t_parse, titles, bams, outputPath, qGene, isStranded, strandedType, isbeta2Cryptic, devices, threads, log, shallow, kept=None):
    with walk:
        log("Establishing order of genomic regions.")
        chroms = region_order_from_runs(walk.region_runs())
        if not chroms:
            log("No genomic regions found - EXITING")
            return {}
        log("order of genomic regions deduced: {}".format(chroms))
        if shallow is not None and qGene != "All":   # combineShallow keeps only the query gene's lines in memory (:948-955)
            walk.keep_gene(qGene)
        log("Iterating through files in parallel, to interleave lines and fill gaps.")
        t0 = time.perf_counter()
        opened = open_kept_reads(kept, bams)      # (beside the walk: the gap fill finds the samples' kept reads open)
        for pos, seen in walk.merge(chroms, isStranded, qGene, shallow):
            log("Skipped site {} for insufficient evidence, only {} samples with Site using minimum reads".format(pos, seen))
        tables = {}
        n_queries = 0
        for idx in range(len(titles)):
            tabs = walk.tables(idx)
            if tabs:
                tables[idx] = _NativeQueryTable(tabs)
                n_queries += sum(int(t["pos"].shape[0]) for _, t in tabs)
        t_merge = time.perf_counter() - t0
        n_gap_sites = walk.n_gap_sites
        t0 = time.perf_counter()
        if tables:
            fill_tables(tables, walk.answers, bams, isStranded, strandedType, devices=devices, threads=threads, log=log, kept_reads=kept, opened=opened)
        if hasattr(opened, "close"):
            opened.close()
        t_fill = time.perf_counter() - t0
        t0 = time.perf_counter()
        walk.write(outputPath + ".combined.tsv", titles, isbeta2Cryptic)
        t_write = time.perf_counter() - t0
        n_sites = walk.n_sites
    log("Filled in Beta read counts for {} Sites not detected in some samples".format(n_gap_sites))
    return dict(walk="native", parse_s=t_parse, merge_s=t_merge, gapfill_s=t_fill, write_s=t_write, sites=n_sites, gap_sites=n_gap_sites,
                queries=n_queries)


def _combine_python(titles, tsvs, bams, outputPath, qGene, isStranded, strandedType, isbeta2Cryptic, devices, threads, log, shallow, kept=None):
    t0 = time.perf_counter()
    rows = [_parse_tsv(p) for p in tsvs]
    t_parse = time.perf_counter() - t0
    log("Establishing order of genomic regions.")
    chroms = region_order(rows)
    if not chroms:
        log("No genomic regions found - EXITING")
        return {}
    log("order of genomic regions deduced: {}".format(chroms))
    if shallow is not None and qGene != "All":   # combineShallow keeps only the query gene's lines in memory (:948-955)
        rows = [[r for r in file_rows if r.gene == qGene] for file_rows in rows]
    log("Iterating through files in parallel, to interleave lines and fill gaps.")
    t0 = time.perf_counter()
    merged = merge_sites(rows, chroms, len(titles), isStranded, qGene, shallow=shallow, log=log)
    n_gap_sites = sum(1 for m in merged if m.queries)
    t_merge = time.perf_counter() - t0
    t0 = time.perf_counter()
    results = fill_gaps(merged, bams, isStranded, strandedType, devices=devices, threads=threads, log=log, kept_reads=kept) if n_gap_sites else {}
    t_fill = time.perf_counter() - t0
    t0 = time.perf_counter()
    write_combined(outputPath + ".combined.tsv", merged, titles, results, isbeta2Cryptic)
    t_write = time.perf_counter() - t0
    log("Filled in Beta read counts for {} Sites not detected in some samples".format(n_gap_sites))
    return dict(walk="python", parse_s=t_parse, merge_s=t_merge, gapfill_s=t_fill, write_s=t_write, sites=len(merged), gap_sites=n_gap_sites,
                queries=sum(len(m.queries) for m in merged))


def combineShallow(samplesFile, outputPath, qGene="All", isStranded=False, minSamples=0, minReads=10, minSSE=0.0,
                   strandedType=None, isbeta2Cryptic=False, devices=(0,), threads=0, log=_log, native_walk=None):
    """SpliSER_v0_1_8.py:920-1167: ``combine`` with the per-site evidence filter (a site is processed only when at
    least ``minSamples`` samples list it with >= ``minReads`` reads and SSE >= ``minSSE``)."""
    return combine(samplesFile, outputPath, qGene=qGene, isStranded=isStranded, strandedType=strandedType,
                   isbeta2Cryptic=isbeta2Cryptic, devices=devices, threads=threads, log=log,
                   shallow=(int(minSamples), int(minReads), float(minSSE)), native_walk=native_walk)
