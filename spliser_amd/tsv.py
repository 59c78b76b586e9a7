"""``.SpliSER.tsv`` writer -- column layout and number formats of outputBedFile
(SpliSER_v0_1_8.py:641-664): 12 tab-separated columns, SSE as ``{:.3f}``, the two cryptic columns as
an int and ``{:.5f}`` (or ``NA NA`` without --beta2Cryptic), Partners as ``str(dict)``, Competitors as
``str(list)``."""

HEADER = ("Region\tSite\tStrand\tGene\tSSE\talpha_count\tbeta1_count\tbeta2Simple_count\t"
          "beta2Cryptic_count\tbeta2Cryptic_weighted\tPartners\tCompetitors\n")


def partners_repr(positions, counts):
    return "{" + ", ".join("%d: %d" % (int(p), int(c)) for p, c in zip(positions, counts)) + "}"


def competitors_repr(positions):
    return "[" + ", ".join("%d" % int(p) for p in positions) + "]"


def format_chrom(arr, res, cryptic):
    """Rows of one chromosome.  ``arr``: sites.ChromArrays; ``res``: dict with beta1, beta2_simple,
    beta2_cryptic, beta2_weighted, sse arrays."""
    out = []
    po, co = arr.part_off, arr.comp_off
    strand, genes = arr.strand, arr.genes
    for i in range(arr.n):
        a, b = int(po[i]), int(po[i + 1])
        c, d = int(co[i]), int(co[i + 1])
        if cryptic:
            mid = "%d\t%s" % (int(res["beta2_cryptic"][i]), "{0:.5f}".format(float(res["beta2_weighted"][i])))
        else:
            mid = "NA\tNA"
        out.append("%s\t%d\t%s\t%s\t%s\t%d\t%d\t%d\t%s\t%s\t%s\n" % (
            arr.chrom, int(arr.pos[i]), arr.strand_text[i], genes[i], "{0:.3f}".format(float(res["sse"][i])),
            int(arr.alpha[i]), int(res["beta1"][i]), int(res["beta2_simple"][i]), mid,
            partners_repr(arr.part_pos[a:b], arr.edge_cnt[a:b]), competitors_repr(arr.comp_pos[c:d])))
    return out
