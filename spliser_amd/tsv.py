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
    n = arr.n
    # plain Python lists: indexing numpy scalars one by one costs more than the formatting itself
    po, co = arr.part_off.tolist(), arr.comp_off.tolist()
    pos, alpha = arr.pos.tolist(), arr.alpha.tolist()
    ppos, pcnt, cpos = arr.part_pos.tolist(), arr.edge_cnt.tolist(), arr.comp_pos.tolist()
    beta1, b2s = _ints(res["beta1"]), _ints(res["beta2_simple"])
    sse = ["%.3f" % v for v in _floats(res["sse"])]          # same digits as "{0:.3f}".format (:655)
    if cryptic:
        b2c = _ints(res["beta2_cryptic"])
        b2w = ["%.5f" % v for v in _floats(res["beta2_weighted"])]
    strand_text, genes, chrom = arr.strand_text, arr.genes, arr.chrom
    for i in range(n):
        a, b = po[i], po[i + 1]
        c, d = co[i], co[i + 1]
        mid = ("%d\t%s" % (b2c[i], b2w[i])) if cryptic else "NA\tNA"
        pairs = list(zip(ppos[a:b], pcnt[a:b]))
        if len(set(ppos[a:b])) != b - a:          # (two partner sites at one position: the dict has the position once, :652)
            pairs = list(dict(pairs).items())
        partners = "{" + ", ".join(["%d: %d" % pc for pc in pairs]) + "}"
        competitors = "[" + ", ".join(map(str, cpos[c:d])) + "]"
        out.append("%s\t%d\t%s\t%s\t%s\t%d\t%d\t%d\t%s\t%s\t%s\n" % (
            chrom, pos[i], strand_text[i], genes[i], sse[i], alpha[i], beta1[i], b2s[i], mid, partners, competitors))
    return out


def _ints(a):
    return [int(v) for v in a.tolist()] if hasattr(a, "tolist") else [int(v) for v in a]


def _floats(a):
    return [float(v) for v in a.tolist()] if hasattr(a, "tolist") else [float(v) for v in a]
