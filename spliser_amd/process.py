"""``process`` -- the SpliSER sub-command whose Step 3 is the MI355X hot path.

Same call signature, same stdout banners (loosely), same ``<outputPath>.SpliSER.tsv`` as
SpliSER_v0_1_8.py:695-720.  Steps 0-2 run on the host (``sites.py``); Step 3 -- the per-site
``checkBam`` loop plus ``findBeta2Counts`` / ``calculateSSE`` (``processSites``, :681-692) -- is one
``spl_count`` + one ``spl_sse`` launch per shard on each GPU (``native.py`` -> libspliser_hip.so).

Deliberate deviations from the reference, all on the failure side:
  * an unreadable / truncated / non-BAM alignment file is an error here; the reference ignores samtools'
    exit status and silently reports zero beta counts (SpliSER_v0_1_8.py:422-427);
  * ``-g GENE`` with a gene that is not in the annotation is an error here; the reference crashes with
    AttributeError at :283.
"""
import sys
import threading
import time

from . import fast_sites, native, samio, shard, sites, tsv


def _log(msg):
    print(msg)
    sys.stdout.flush()


class _SamSource(object):
    """Reads from SAM text (small inputs / fixtures)."""

    def __init__(self, path):
        self.ref_names, self._sets = samio.read_sam(path)

    def reads(self, chrom):
        return self._sets.get(chrom)


def open_alignments(path, threads=0):
    """BAM (BGZF) through the native decoder; plain SAM text through the Python reader."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:2] == b"\x1f\x8b":
        return native.BamFile(path, threads=threads)
    if magic[:1] == b"@" or b"\t" in open(path, "rb").readline():
        return _SamSource(path)
    raise native.SpliserNativeError(-5, "%s is neither BGZF/BAM nor SAM text" % path)


def process_sites(table, source, q_chrom, is_stranded, stranded_type, is_beta2_cryptic, devices=(0,), combine_mode=0,
                  log=_log, timings=None):
    """processSites (SpliSER_v0_1_8.py:681-692) for every chromosome at once.

    -> {chrom: (ChromArrays, results dict)} with results keys beta1, beta2_simple, beta2_cryptic,
    beta2_weighted, sse (numpy arrays in table row order).
    """
    stranded = native.STRANDED_CODE[stranded_type] if is_stranded else 0
    if is_stranded and stranded == 0:
        raise ValueError("strandedType must be 'fr' or 'rf' for a stranded analysis")
    items = {}
    for chrom in table.chrom_index:
        log("Processing region " + str(chrom))
        if not (q_chrom == chrom or q_chrom == "All"):
            continue
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        reads = source.reads(chrom)
        if reads is None:
            log("  (no reference named %s in the alignment file: all beta counts are 0)" % chrom)
        items[chrom] = (arr, reads)
    weights = {c: (r.n if r is not None else 0) + a.n for c, (a, r) in items.items()}
    plan = shard.assign(weights, len(devices))
    out, errors = {}, []
    lock = threading.Lock()

    def run(device, chroms):
        try:
            if not chroms:
                return
            t0 = time.perf_counter()
            order = [c for c in table.chrom_index if c in chroms]
            shards = shard.pack([(c, items[c][0], items[c][1]) for c in order], concat_reads=False)
            t1 = time.perf_counter()
            with native.Context(device) as ctx:
                for sh in shards:
                    ds = ctx.upload_sites(sh.sites)
                    dr = ctx.upload_read_segments(sh.read_segments)  # straight from the decoder's buffers
                    ctx.count_launch(ds, dr, stranded, combine_mode)
                    ctx.sse_launch(ds, is_beta2_cryptic)
                    beta1, _, _ = ds.counters()
                    b2s, b2c, b2w, sse = ds.sse_results()
                    dr.free()
                    ds.free()
                    with lock:
                        for chrom, (r0, r1) in zip(sh.chroms, sh.site_rows):
                            out[chrom] = (items[chrom][0], dict(beta1=beta1[r0:r1], beta2_simple=b2s[r0:r1],
                                                                beta2_cryptic=b2c[r0:r1], beta2_weighted=b2w[r0:r1],
                                                                sse=sse[r0:r1]))
            if timings is not None:
                with lock:
                    timings.setdefault("pack_s", 0.0)
                    timings["pack_s"] += t1 - t0
                    timings.setdefault("gpu_s", 0.0)
                    timings["gpu_s"] += time.perf_counter() - t1
        except Exception as exc:  # surfaced after join
            with lock:
                errors.append(exc)

    threads = [threading.Thread(target=run, args=(dev, chroms)) for dev, chroms in zip(devices, plan)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def write_tsv(output_path, table, results, is_beta2_cryptic):
    """outputBedFile (SpliSER_v0_1_8.py:641-664); the rows are formatted by the native library (tsv.format_chrom is the
    Python statement of the same format and what the CPU tests compare it with)."""
    path = output_path + ".SpliSER.tsv"
    with open(path, "w") as fh:
        fh.write(tsv.HEADER)
    for chrom in table.chrom_index:
        if chrom in results:
            arr, res = results[chrom]
            native.tsv_append(path, arr, res, is_beta2_cryptic)


def process(inBAM, inBed, outputPath, qGene="All", qChrom="All", maxIntronSize=0, annotationFile=None, aType="gene",
            isStranded=False, strandedType=None, isbeta2Cryptic=False, devices=(0,), threads=0, log=_log):
    """SpliSER_v0_1_8.py:695-720, keyword-compatible with the reference's argparse dests."""
    timings = {}
    t0 = time.perf_counter()
    # The alignment file does not depend on Steps 0-2: it is decoded on native threads (GIL released) while the site table
    # is built here.  Its outcome -- reads or an error -- is picked up where the reference's Step 3 begins.
    opened = {}

    def _open():
        t = time.perf_counter()
        try:
            opened["source"] = open_alignments(inBAM, threads=threads)
        except BaseException as exc:  # re-raised on the main thread below
            opened["error"] = exc
        opened["seconds"] = time.perf_counter() - t
    opener = threading.Thread(target=_open, name="spliser-bam-decode")
    opener.start()
    try:
        table = _site_table(inBed, qGene, qChrom, maxIntronSize, annotationFile, aType, isStranded, strandedType, log)
    except BaseException:
        opener.join()
        raise
    t1 = time.perf_counter()
    log("\n\nStep 3: Finding Beta reads")
    log("Processing sample 1 out of 1")
    opener.join()
    if "error" in opened:
        raise opened["error"]
    source = opened["source"]
    t2 = time.perf_counter()
    results = process_sites(table, source, qChrom, isStranded, strandedType, isbeta2Cryptic, devices=devices, log=log,
                            timings=timings)
    t3 = time.perf_counter()
    log("\nOutputting .tsv file")
    write_tsv(outputPath, table, results, isbeta2Cryptic)
    t4 = time.perf_counter()
    timings.update(site_table_s=t1 - t0, decode_s=opened["seconds"], decode_wait_s=t2 - t1, step3_s=t3 - t2, write_s=t4 - t3)
    return timings


def _site_table(inBed, qGene, qChrom, maxIntronSize, annotationFile, aType, isStranded, strandedType, log):
    """Steps 0-2 (SpliSER_v0_1_8.py:700-712)."""
    log("Processing")
    log("Stranded Analysis {}".format(strandedType) if isStranded else "Unstranded Analysis")
    bins = sites.GeneBins()
    if annotationFile is not None:
        log("\n\nStep 0: Creating Genes from Annotation...")
        bins = sites.GeneBins.from_annotation(annotationFile, aType, qGene, log=log)
    log("\n\nPreparing Splice Site Arrays")
    log("\n\nStep 1: Finding Splice Sites / Counting Alpha reads...")
    log("Processing sample 1 out of 1")
    # one sample, no gene query, ordinary strands: the table follows from a few sorts (fast_sites.py, held to the
    # line-by-line builder by tests/test_fast_sites.py); anything else is built line by line
    table = fast_sites.build(bins, isStranded, inBed, q_chrom=qChrom, q_gene=qGene, max_intron=int(maxIntronSize))
    if table is None:
        table = sites.SiteTable(bins, is_stranded=isStranded)
        table.add_bed(inBed, q_chrom=qChrom, q_gene=qGene, max_intron=int(maxIntronSize))
    log("Sites assessed:\t" + str(table.assessed))
    log("Sites found:\t\t\t" + str(table.created))
    log("Sites assigned to a Gene:\t" + str(table.assigned))
    log("Sites:\t\t\t" + str(table.n_sites()))
    log("\n\nStep 2: Identifying Competitors of each splice site...")
    table.find_competitors()
    return table
