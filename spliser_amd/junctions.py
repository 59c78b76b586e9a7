"""``junctions`` -- the BED12 junction file ``process`` takes with ``-b``, derived from the BAM itself on the GPU.

Not part of SpliSER v0.1.8: its README (README.md:41) sends the user to ``regtools junctions extract`` for this file.  The
table comes from ``spl_junctions`` (one insert per N op into a device hash table, see spl_kernels.hip); the line layout is
the one findAlphaCounts reads (SpliSER_v0_1_8.py:259-277): ``leftpos = chromStart + blockSizes[0]``, ``rightpos =
chromEnd - blockSizes[1]``, ``alpha = score``, strand in column 6.  Defaults of the policy knobs are regtools' (-a 8 -m 70
-M 500000); strand is the read strand by check_strand's rule for a stranded library and ``?`` otherwise (regtools can also
take it from the aligner's XS tag, which this build does not decode).
"""
import sys

from . import native, process as _process


def write_junction_bed(handle, chrom, table, first_number=1):
    n = len(table["left"])
    left, right = table["left"].tolist(), table["right"].tolist()
    strand, count = table["strand"].tolist(), table["count"].tolist()
    a_left, a_right = table["anchor_left"].tolist(), table["anchor_right"].tolist()
    for i in range(n):
        start, end = left[i] - a_left[i], right[i] + a_right[i]
        handle.write("%s\t%d\t%d\tJUNC%08d\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
            chrom, start, end, first_number + i, count[i], chr(strand[i]), start, end, a_left[i], a_right[i], end - start - a_right[i]))
    return n


def junctions(inBAM, outputPath, isStranded=False, strandedType=None, minAnchor=8, minIntron=70, maxIntron=500000,
              qChrom="All", devices=(0,), threads=0, log=None):
    """Writes ``outputPath`` (a BED12 file) and returns the number of junctions."""
    log = log or (lambda msg: (print(msg), sys.stdout.flush()))
    stranded = native.STRANDED_CODE[strandedType] if isStranded else 0
    if isStranded and stranded == 0:
        raise ValueError("strandedType must be 'fr' or 'rf' for a stranded library")
    import threading
    from . import shard
    source = _process.open_and_decode(inBAM, tuple(devices), None, threads)   # (on the GPU with one device, like `process`)
    is_bam = isinstance(source, native.BamFile)
    chroms = [c for c in source.ref_names if qChrom == c or qChrom == "All"]
    lengths = dict(zip(source.ref_names, source.ref_lengths)) if is_bam else {c: (source.reads(c).n if source.reads(c) is not None else 0) for c in chroms}
    plan = shard.assign({c: int(lengths.get(c, 1)) for c in chroms}, len(devices))   # chromosomes over the devices, longest first
    tables, errors, lock = {}, [], threading.Lock()

    def run(device, mine):
        try:
            with native.Context(device) as ctx:
                for chrom in [c for c in chroms if c in mine]:     # (file order: a chromosome is complete when the next begins)
                    with ctx.begin_reads() as dr:
                        if is_bam:
                            n = dr.add_bam(source, chrom)
                        else:
                            rs = source.reads(chrom)
                            n = rs.n if rs is not None else 0
                            if n:
                                dr.add(native.ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar))
                        if n == 0:
                            continue
                        dr.finish()
                        table = dr.junctions(stranded, minAnchor, minIntron, maxIntron)
                    with lock:
                        tables[chrom] = (n, table)
        except Exception as exc:
            with lock:
                errors.append(exc)
    workers = [threading.Thread(target=run, args=(dev, set(mine))) for dev, mine in zip(devices, plan)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    try:
        if errors:
            raise errors[0]
        if is_bam and not source.wait_all():
            raise native.SpliserNativeError(-5, "%s is not sorted by reference: sort it (samtools sort) first" % inBAM)
    finally:
        if hasattr(source, "close"):
            source.close()
    total = 0
    with open(outputPath, "w") as out:
        out.write('track name=junctions description="spliser_amd junctions (a>=%d, %d<=intron<=%d)"\n' % (minAnchor, minIntron, maxIntron))
        for chrom in chroms:
            if chrom not in tables:
                continue
            n, table = tables[chrom]
            total += write_junction_bed(out, chrom, table, total + 1)
            log("%s: %d reads, %d junctions" % (chrom, n, len(table["left"])))
    log("Junctions written:\t%d" % total)
    return total
