"""`combine` / `output` against goldens produced by the real reference (tests/golden/combine_a).

CPU: the host walk (region order, lock-step merge, query snapshots, writers) is driven with the oracle answering the
gap-fill queries in the test -- the product's fill_gaps only knows the GPU.  GPU (-m gpu): the CLI end to end."""
import json
import os

import numpy as np
import pytest

import helpers
from spliser_amd import combine as cmb
from spliser_amd import output as outp
from spliser_amd import process as proc

CASE = os.path.join(helpers.GOLDEN, "combine_a")
MANIFEST = json.load(open(os.path.join(CASE, "combine_manifest.json")))
VARIANTS = sorted(MANIFEST["variants"])


def _samples_file(tmp_path, variant):
    path = str(tmp_path / "samples.tsv")
    with open(path, "w") as fh:
        for k in range(MANIFEST["n_samples"]):
            sd = os.path.join(CASE, "sample%d" % k)
            fh.write("S%d\t%s\t%s\n" % (k, os.path.join(sd, "expected.%s.tsv" % variant), os.path.join(sd, "reads.sam")))
    return path


def _flags(variant):
    v = MANIFEST["variants"][variant]
    args = v["combine"]
    shallow = None
    if v.get("command") == "combineShallow":
        def opt(flag, default, cast):
            return cast(args[args.index(flag) + 1]) if flag in args else default
        shallow = (opt("-m", 0, int), opt("-r", 10, int), opt("-e", 0.0, float))
    return dict(q_gene=args[args.index("-g") + 1] if "-g" in args else "All", stranded="--isStranded" in args,
                stype=args[args.index("-s") + 1] if "-s" in args else "fr", cryptic="--beta2Cryptic" in args, shallow=shallow)


@pytest.mark.parametrize("variant", VARIANTS)
def test_combine_host_walk_with_oracle_gap_fill(variant, tmp_path, oracle_lib):
    f = _flags(variant)
    titles, tsvs, bams = cmb.read_samples_file(_samples_file(tmp_path, variant))
    rows = [cmb._parse_tsv(p) for p in tsvs]
    chroms = cmb.region_order(rows)
    if f["shallow"] is not None and f["q_gene"] != "All":
        rows = [[r for r in fr if r.gene == f["q_gene"]] for fr in rows]
    merged = cmb.merge_sites(rows, chroms, len(titles), f["stranded"], f["q_gene"], shallow=f["shallow"])
    stranded = {"fr": 1, "rf": 2}[f["stype"]] if f["stranded"] else 0
    results = {}
    for idx, queries in cmb.gap_queries(merged).items():
        source = proc.open_alignments(bams[idx])
        table = cmb._QueryTable(queries)
        for chrom in table.chrom_index:
            s, r = table.chrom_arrays(chrom), source.reads(chrom)
            if r is None or r.n == 0:
                b1 = b2 = [0] * s.n
            else:
                b1, b2, _ = oracle_lib.check_bam(s.pos, s.strand, s.part_off, s.part_pos, s.comp_off, s.comp_pos,
                                                 r.pos, r.flag, r.cig_off, r.cigar, stranded, 1)
            for k, si in enumerate(table.site_index[chrom]):
                results[(si, idx)] = (int(b1[k]), int(b2[k]))
    out = str(tmp_path / "all.combined.tsv")
    cmb.write_combined(out, merged, titles, results, f["cryptic"])
    assert open(out).read() == open(os.path.join(CASE, "expected.%s.combined.tsv" % variant)).read()
    assert sum(len(m.queries) for m in merged) > 0 or variant == "shallow_fr"


@pytest.mark.parametrize("variant", VARIANTS)
def test_output_diffspliser_and_gwas(variant, tmp_path):
    sfile = _samples_file(tmp_path, variant)
    combined = os.path.join(CASE, "expected.%s.combined.tsv" % variant)
    outp.output("DiffSpliSER", sfile, combined, str(tmp_path / "diff_"), minReads=5, qGene="All")
    assert open(str(tmp_path / "diff_All.DiffSpliSER.tsv")).read() == open(os.path.join(CASE, "expected.%s.DiffSpliSER.tsv" % variant)).read()
    gdir = str(tmp_path / "gwas") + os.sep
    os.makedirs(gdir)
    outp.output("GWAS", sfile, combined, gdir, minReads=5, qGene="All", minSamples=2)
    want = json.load(open(os.path.join(CASE, "expected.%s.GWAS.json" % variant)))
    got = {f: open(os.path.join(gdir, f)).read() for f in sorted(os.listdir(gdir))}
    assert got == want


def test_region_order_matches_reference_rule():
    class R(object):
        def __init__(self, c):
            self.chrom = c
    files = [[R("b"), R("b"), R("d")], [R("a"), R("b"), R("c"), R("d")], [R("e")]]
    order = cmb.region_order(files)
    assert set(order) == {"a", "b", "c", "d", "e"}
    assert order.index("a") < order.index("b") < order.index("d") and order.index("b") < order.index("c") < order.index("d")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_combine_cli_on_gpu(variant, tmp_path):
    from spliser_amd import cli
    sfile = _samples_file(tmp_path, variant)
    v = MANIFEST["variants"][variant]
    argv = [v.get("command", "combine"), "-S", sfile, "-o", str(tmp_path / "all")] + v["combine"]
    assert cli.main(argv) == 0
    assert open(str(tmp_path / "all.combined.tsv")).read() == open(os.path.join(CASE, "expected.%s.combined.tsv" % variant)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_process_reproduces_per_sample_goldens(variant, tmp_path):
    """The per-sample .SpliSER.tsv files the combine goldens were built from are process goldens too."""
    from spliser_amd import cli
    p = MANIFEST["variants"][variant]["process"]
    for k in range(MANIFEST["n_samples"]):
        sd = os.path.join(CASE, "sample%d" % k)
        argv = ["process", "-B", os.path.join(sd, "reads.sam"), "-b", os.path.join(sd, "junctions.bed"), "-o", str(tmp_path / ("s%d" % k))]
        if p.get("gff"):
            argv += ["-A", os.path.join(sd, "genes.gff")]
        if p.get("stranded"):
            argv += ["--isStranded", "-s", p["stranded"]]
        if p.get("cryptic"):
            argv += ["--beta2Cryptic"]
        assert cli.main(argv) == 0
        assert open(str(tmp_path / ("s%d.SpliSER.tsv" % k))).read() == open(os.path.join(sd, "expected.%s.tsv" % variant)).read()
