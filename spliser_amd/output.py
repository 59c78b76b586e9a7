"""``output`` -- reshape a ``.combined.tsv`` into DiffSpliSER / GWAS tables (SpliSER_v0_1_8.py:1169-1292).
Pure text reshaping on the host; no BAM, no GPU."""
import sys


def _titles(samples_file, three_columns_only):
    titles = []
    with open(samples_file, "r") as fh:
        for line in fh:
            values = line.split("\t")
            if not three_columns_only or len(values) == 3:
                titles.append(values[0])
    return titles


def _site_groups(combined_file, n):
    """Groups of n consecutive data lines (one per sample); an incomplete trailing group is dropped (:1196-1206)."""
    with open(combined_file, "r") as fh:
        next(fh, None)  # header
        while True:
            group = []
            for _ in range(n):
                line = next(fh, None)
                if line is None:
                    return
                group.append(line.rstrip().split("\t"))
            yield group


def diffspliser_output(samplesFile, combinedFile, outputPath, minReads, qGene):
    """DiffSpliSER_output (:1169-1229): one wide row per site with alpha / beta / SSE per sample."""
    titles = _titles(samplesFile, True)
    with open(outputPath + str(qGene) + ".DiffSpliSER.tsv", "w") as out:
        out.write("Region\tSite\tStrand\tGene")
        for t in titles:
            out.write("\t%s_alpha\t%s_beta\t%s_SSE" % (t, t, t))
        out.write("\n")
        if not titles:
            return
        for group in _site_groups(combinedFile, len(titles)):
            first = group[0]
            out.write("\t".join(str(first[k]) for k in (1, 2, 3, 4)))
            for v in group:
                alpha = int(v[6])
                beta = float(v[7]) + float(v[8])
                weighted = float(v[10]) if v[10] != "NA" else 0
                sse = float(v[5])
                if alpha + beta >= minReads:
                    beta = beta + weighted
                    out.write("\t" + str(alpha) + "\t" + "{0:.2f}".format(beta) + "\t" + "{0:.2f}".format(sse))
                else:
                    out.write("\tNA\tNA\tNA")
            out.write("\n")


def gwas_output(samplesFile, combinedFile, outputPath, minReads, qGene, minSamples):
    """GWAS_output (:1231-1286): one two-column phenotype file per site plus a filter log."""
    print(qGene)
    titles = _titles(samplesFile, False)
    if not titles:
        return
    for group in _site_groups(combinedFile, len(titles)):
        gene, site = str(group[0][4]), str(group[0][2])
        if not (qGene == gene or qGene == "All"):
            continue
        passing, buffered = 0, ""
        with open(str(outputPath + gene + "_" + site + "_filtered.log"), "w") as filtered:
            for t, v in zip(titles, group):
                alpha = int(v[6])
                beta = float(v[7]) + float(v[8])
                if alpha + beta >= minReads:
                    passing += 1
                    buffered += str(v[0]) + "\t" + str(v[5]) + "\n"
                else:
                    filtered.write(str(t) + " did not pass minReads for " + site + "\n")
            if passing >= minSamples:
                with open(outputPath + gene + "_" + site + ".tsv", "w") as out:
                    out.write(buffered)
            else:
                filtered.write("Site: " + site + " minSamples not met - venting buffer\n")
                filtered.write(buffered + "\n")


def output(outputType, samplesFile, combinedFile, outputPath, minReads=10, qGene="All", minSamples=50):
    """SpliSER_v0_1_8.py:1288-1292."""
    if outputType == "DiffSpliSER":
        diffspliser_output(samplesFile, combinedFile, outputPath, minReads, qGene)
    if outputType == "GWAS":
        gwas_output(samplesFile, combinedFile, outputPath, minReads, qGene, minSamples)
    sys.stdout.flush()
