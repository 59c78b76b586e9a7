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
    source = _process.open_alignments(inBAM, threads=threads)
    total = 0
    with native.Context(devices[0]) as ctx, open(outputPath, "w") as out:
        out.write('track name=junctions description="spliser_amd junctions (a>=%d, %d<=intron<=%d)"\n' % (minAnchor, minIntron, maxIntron))
        for chrom in source.ref_names:
            if not (qChrom == chrom or qChrom == "All"):
                continue
            reads = source.reads(chrom)
            if reads is None or reads.n == 0:
                continue
            dr = ctx.upload_reads(native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar))
            table = dr.junctions(stranded, minAnchor, minIntron, maxIntron)
            dr.free()
            total += write_junction_bed(out, chrom, table, total + 1)
            log("%s: %d reads, %d junctions" % (chrom, reads.n, len(table["left"])))
    log("Junctions written:\t%d" % total)
    return total
