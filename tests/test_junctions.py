"""spl_junctions (junction table of a read set on the device) against the plain-Python restatement in
oracle/oracle.py and against what the synthetic generator knows it put into the reads."""
import os

import pytest

import helpers
from oracle import oracle
from spliser_amd import native, samio, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def _device_table(ctx, reads, stranded):
    dr = ctx.upload_reads(native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar))
    j = dr.junctions(stranded)
    dr.free()
    return [tuple(int(j[k][i]) for k in ("left", "right", "strand", "count", "anchor_left", "anchor_right")) for i in range(len(j["left"]))]


@pytest.mark.parametrize("case", ["cigar_corners", "random_b", "kat1", "kat2", "multichrom", "random_unstranded_q"])
@pytest.mark.parametrize("stranded", [0, 1, 2])
def test_junction_table_matches_restatement_on_golden_reads(case, stranded, ctx):
    path = os.path.join(helpers.GOLDEN, case, "reads.sam")
    if not os.path.exists(path):
        pytest.skip("no such golden case")
    _, sets = samio.read_sam(path)
    for chrom, reads in sets.items():
        want = oracle.junction_table(reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded)
        assert _device_table(ctx, reads, stranded) == want


def test_junction_table_of_adversarial_cigars(ctx):
    recs = [(0, 100, "10M0N10M"), (0, 100, "10M5N5N10M"), (16, 100, "5S10M100N2I3D7M50N8M2H"), (4, 100, "10M100N10M"),
            (0, 100, "*"), (99, 300, "10M100N10M"), (147, 300, "10M100N10M"), (0, 100, "100N10M"), (0, 100, "10M100N"),
            (0, 100, "10=1X100N3D10M"), (0, 1, "1M1N1M")] + [(0, 5000, "30M200N30M")] * 700
    reads = samio.ReadSet.from_records(recs)
    for stranded in (0, 1, 2):
        want = oracle.junction_table(reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded)
        assert _device_table(ctx, reads, stranded) == want
    assert any(row[:2] == (5029, 5229) and row[3] == 700 for row in want)
    empty = samio.ReadSet.empty()
    assert _device_table(ctx, empty, 0) == []


def test_junction_table_recovers_what_the_generator_spliced(ctx):
    wl = synth.Workload("arabidopsis", scale=0.02, seed=9)
    chrom, left, right, strand, count = wl.junctions
    for c, reads in enumerate(wl.reads):
        got = _device_table(ctx, reads, 0)
        sel = chrom == c
        want = sorted(zip(left[sel].tolist(), right[sel].tolist(), count[sel].tolist()))
        assert [(r[0], r[1], r[3]) for r in got] == want
        assert all(r[4] >= 1 and r[5] >= 1 for r in got)


def test_junctions_cli_feeds_process(ctx, tmp_path, oracle_lib):
    """BAM -> `junctions` -> BED12 -> `process`: alpha of every site = reads spliced there, counted from the same reads."""
    import numpy as np
    from spliser_amd import cli, sites
    wl = synth.Workload("arabidopsis", scale=0.005, seed=4)
    bam, bed = str(tmp_path / "s.bam"), str(tmp_path / "s.bed")
    native.write_bam(bam, wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2)
    assert cli.main(["junctions", "-B", bam, "-o", bed, "-a", "1", "-m", "1", "-M", "0"]) == 0
    table = sites.SiteTable()
    table.add_bed(bed)
    chrom, left, right, strand, count = wl.junctions
    for c, name in enumerate(wl.genome.chrom_names):
        arr = table.chrom_arrays(name)
        want = {}
        sel = chrom == c
        for l, r, n in zip(left[sel].tolist(), right[sel].tolist(), count[sel].tolist()):
            want[l] = want.get(l, 0) + n
            want[r] = want.get(r, 0) + n
        assert dict(zip(arr.pos.tolist(), arr.alpha.tolist())) == want
    # the default policy (-a 8 -m 70) drops short anchors: never more junctions, never larger counts
    bed2 = str(tmp_path / "s2.bed")
    assert cli.main(["junctions", "-B", bam, "-o", bed2]) == 0
    strict = sites.SiteTable()
    strict.add_bed(bed2)
    assert 0 < strict.n_sites() <= table.n_sites()
    assert cli.main(["process", "-B", bam, "-b", bed2, "-o", str(tmp_path / "out")]) == 0
    assert sum(1 for _ in open(str(tmp_path / "out.SpliSER.tsv"))) == strict.n_sites() + 1
    # the min-anchor rule is per read: restated on the reads of one chromosome
    reads = wl.reads[0]
    want = [row for row in oracle.junction_table(reads.pos, reads.flag, reads.cig_off, reads.cigar, 0)]
    dr = ctx.upload_reads(native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar))
    got = dr.junctions(0, 20, 0, 0)
    dr.free()
    assert len(got["left"]) <= len(want) and int(got["anchor_left"].min()) >= 20 and int(got["anchor_right"].min()) >= 20


def test_junctions_command_writes_the_golden_beds(tmp_path):
    """The BED12 files the REAL reference was run on (tests/golden/junctions_*: its findAlphaCounts read them into the sites and
    alpha counts of expected.*.tsv) are what the `junctions` command writes from the same reads today, byte for byte."""
    import sys
    sys.path.insert(0, helpers.GOLDEN)
    from make_golden import JUNCTION_CASES, JUNCTION_KNOBS
    from spliser_amd.junctions import junctions
    for name, case in JUNCTION_CASES.items():
        out = str(tmp_path / (name + ".bed"))
        junctions(os.path.join(helpers.GOLDEN, case["reads_of"], "reads.sam"), out, log=lambda m: None, **dict(JUNCTION_KNOBS, **case["junctions"]))
        assert open(out).read() == open(os.path.join(helpers.GOLDEN, name, "junctions.bed")).read()
