"""The array-at-a-time site table (spliser_amd/fast_sites.py) against the line-by-line one (sites.SiteTable), which the
reference goldens pin: identical arrays, gene names, strand text and counters on every golden input and on random BEDs."""
import glob
import os

import numpy as np
import pytest

import helpers
from spliser_amd import fast_sites, native, sites

CASES = sorted(os.path.basename(os.path.dirname(p)) for p in glob.glob(os.path.join(helpers.GOLDEN, "*", "junctions.bed")))


def _both(bed, gff, is_stranded, q_chrom="All"):
    def bins():
        return sites.GeneBins.from_annotation(gff, "gene", "All") if gff else sites.GeneBins()
    slow = sites.SiteTable(bins(), is_stranded=is_stranded)
    slow.add_bed(bed, q_chrom=q_chrom)
    slow.find_competitors()
    fast = fast_sites.build(bins(), is_stranded, bed, q_chrom=q_chrom)
    return slow, fast


def _assert_same(slow, fast):
    assert fast is not None
    assert list(fast.chrom_index) == list(slow.chrom_index)
    assert (fast.assessed, fast.created, fast.assigned, fast.n_sites()) == (slow.assessed, slow.created, slow.assigned, slow.n_sites())
    for chrom in slow.chrom_index:
        a, b = slow.chrom_arrays(chrom), fast.chrom_arrays(chrom)
        assert a.n == b.n
        for name in ("pos", "strand", "alpha", "part_off", "part_pos", "part_site", "edge_cnt", "comp_off", "comp_pos"):
            x, y = getattr(a, name), getattr(b, name)
            assert x.dtype == y.dtype and np.array_equal(x, y), (chrom, name)
        assert a.genes == b.genes and a.strand_text == b.strand_text


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("is_stranded", [False, True])
@pytest.mark.parametrize("with_gff", [False, True])
def test_fast_table_equals_line_by_line_table_on_goldens(case, is_stranded, with_gff):
    d = os.path.join(helpers.GOLDEN, case)
    gff = os.path.join(d, "genes.gff")
    if with_gff and not os.path.exists(gff):
        pytest.skip("case has no annotation")
    slow, fast = _both(os.path.join(d, "junctions.bed"), gff if with_gff else None, is_stranded)
    if fast is None:   # declined: then the input must really be outside the fast regime
        assert _outside_fast_regime(os.path.join(d, "junctions.bed"), is_stranded)
        return
    _assert_same(slow, fast)


def _outside_fast_regime(bed, is_stranded):
    for line in open(bed):
        v = line.split("\t")
        if len(v) != 12:
            continue
        flank = v[10].split(",")
        if int(v[1]) + int(flank[0]) == int(v[2]) - int(flank[1]):
            return True
        if is_stranded and v[5] not in ("+", "-"):
            return True
    return False


def _random_bed(path, gff, rng, n_lines, odd_strands=False, self_loops=False):
    chroms = ["c%d" % i for i in range(3)]
    with open(gff, "w") as fh:
        for c in chroms[:2]:    # the third chromosome has no genes
            for g in range(40):
                a = int(rng.integers(1, 90000))
                b = a + int(rng.integers(50, 20000))    # overlapping genes on purpose
                fh.write("%s\tsrc\tgene\t%d\t%d\t.\t%s\t.\tID=g%s_%d;x=1\n" % (c, a, b, rng.choice(["+", "-", "."]), c, g))
    anchors = rng.integers(100, 100000, 60)
    with open(path, "w") as fh:
        fh.write("track name=junctions\n")
        for _ in range(n_lines):
            c = chroms[int(rng.integers(0, 3))]
            a, b = sorted(int(x) for x in rng.choice(anchors, 2, replace=False))
            if self_loops and rng.random() < 0.02:
                b = a
            strand = rng.choice(["+", "-"]) if not odd_strands or rng.random() < 0.9 else rng.choice(["?", "."])
            o1, o2 = int(rng.integers(1, 30)), int(rng.integers(1, 30))
            fh.write("%s\t%d\t%d\tj\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
                c, a - o1, b + o2, int(rng.integers(1, 500)), strand, a - o1, b + o2, o1, o2, b - a + o2))


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("is_stranded", [False, True])
def test_fast_table_equals_line_by_line_table_on_random_beds(tmp_path, seed, is_stranded):
    rng = np.random.default_rng(seed)
    bed, gff = str(tmp_path / "j.bed"), str(tmp_path / "g.gff")
    _random_bed(bed, gff, rng, 600)
    for q_chrom in ("All", "c1"):
        slow, fast = _both(bed, gff, is_stranded, q_chrom)
        _assert_same(slow, fast)


def test_fast_table_declines_what_it_cannot_reproduce(tmp_path):
    rng = np.random.default_rng(3)
    bed, gff = str(tmp_path / "j.bed"), str(tmp_path / "g.gff")
    _random_bed(bed, gff, rng, 300, odd_strands=True)
    assert fast_sites.build(sites.GeneBins(), True, bed) is None          # strand-free look-ups in a stranded run
    slow, fast = _both(bed, gff, False)                                    # unstranded: strands are only text
    _assert_same(slow, fast)
    _random_bed(bed, gff, rng, 300, self_loops=True)
    assert fast_sites.build(sites.GeneBins(), False, bed) is None          # a line whose two ends are one position
    assert fast_sites.build(sites.GeneBins(), False, bed, q_gene="g1") is None


def test_native_gene_search_follows_the_reference_probe_sequence():
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(0, 12))
        genes = sorted((sites.Gene("g%d" % i, int(rng.integers(0, 300)), 0, rng.choice(["+", "-", "."])) for i in range(n)),
                       key=lambda g: g.left)
        for g in genes:
            g.right = g.left + int(rng.integers(0, 120))
        qpos = rng.integers(-5, 450, 64)
        qstr = rng.choice(["+", "-", "?"], 64)
        for is_stranded in (False, True):
            want = [sites.gene_search(genes, int(p), s, is_stranded) for p, s in zip(qpos, qstr)]
            got = native.gene_search([g.left for g in genes], [g.right for g in genes],
                                     [fast_sites._strand_code(g.strand) for g in genes], qpos,
                                     [fast_sites._strand_code(s) for s in qstr], is_stranded)
            assert got.tolist() == want
