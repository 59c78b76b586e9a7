"""Shared test plumbing: run the host pipeline of spliser_amd with a pluggable counting engine."""
import json
import os

import numpy as np

from spliser_amd import samio, sites, tsv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRANDED = {None: 0, "fr": 1, "rf": 2}


def build_table(case_dir, opts):
    gff = os.path.join(case_dir, "genes.gff") if opts.get("gff") else None
    q_gene = opts.get("gene") or "All"
    bins = sites.GeneBins.from_annotation(gff, "gene", q_gene) if gff else sites.GeneBins()
    table = sites.SiteTable(bins, is_stranded=bool(opts.get("stranded")))
    table.add_bed(os.path.join(case_dir, "junctions.bed"), q_chrom=opts.get("chrom") or "All", q_gene=q_gene,
                  max_intron=opts.get("max_intron") or 0)
    table.find_competitors()
    return table


def oracle_engine(oracle):
    def count(arr, reads, stranded, combine_mode):
        return oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, combine_mode)

    def sse(arr, beta1, b2s_reads, dbl, cryptic):
        return oracle.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                                beta1, b2s_reads, dbl, cryptic)
    return count, sse


def run_case(case, opts, engine):
    """-> (tsv text, per-site dict rows) for one golden case/variant using ``engine`` = (count, sse)."""
    count, sse_fn = engine
    case_dir = os.path.join(GOLDEN, case)
    table = build_table(case_dir, opts)
    _, reads = samio.read_sam(os.path.join(case_dir, "reads.sam"))
    q_chrom = opts.get("chrom") or "All"
    cryptic = bool(opts.get("cryptic"))
    text = [tsv.HEADER]
    rows = []
    for chrom in table.chrom_index:
        if not (q_chrom == chrom or q_chrom == "All"):
            continue
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        rs = reads.get(chrom, samio.ReadSet.empty())
        beta1, b2s_reads, dbl = count(arr, rs, STRANDED[opts.get("stranded")], 0)
        b2s, b2c, b2w, sse = sse_fn(arr, beta1, b2s_reads, dbl, cryptic)
        res = dict(beta1=beta1, beta2_simple=b2s, beta2_cryptic=b2c, beta2_weighted=b2w, sse=sse)
        text.extend(tsv.format_chrom(arr, res, cryptic))
        for i in range(arr.n):
            rows.append(dict(chrom=chrom, pos=int(arr.pos[i]), strand=arr.strand_text[i], beta1=int(beta1[i]),
                             beta2Simple=int(b2s[i]), beta2Cryptic=int(b2c[i]), beta2Weighted=float(b2w[i]),
                             sse=float(sse[i]), alpha=int(arr.alpha[i])))
    return "".join(text), rows


def expected(case, variant):
    d = os.path.join(GOLDEN, case)
    with open(os.path.join(d, "expected.%s.tsv" % variant)) as fh:
        text = fh.read()
    with open(os.path.join(d, "expected.%s.json" % variant)) as fh:
        rows = json.load(fh)
    return text, rows


def assert_rows_match(rows, ref_rows, cryptic):
    assert len(rows) == len(ref_rows)
    for got, ref in zip(rows, ref_rows):
        key = (ref["chrom"], ref["pos"], ref["strand"])
        assert (got["chrom"], got["pos"], got["strand"]) == key
        for f in ("alpha", "beta1", "beta2Simple", "beta2Cryptic"):
            assert got[f] == ref[f], (key, f, got[f], ref[f])
        # bit-exact doubles: the reference computes beta2Weighted always, SSE with/without it
        assert got["beta2Weighted"] == ref["beta2Weighted"], (key, got["beta2Weighted"], ref["beta2Weighted"])
        assert got["sse"] == ref["sse"], (key, got["sse"], ref["sse"])
        assert abs(got["sse"] - ref["sse"]) <= 1e-9     # the tolerance BASELINE.json states
