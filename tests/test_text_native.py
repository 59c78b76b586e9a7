"""The native readers of Steps 0-1 (spl_bed_open, spl_gff_open) against the line-by-line Python they stand in for
(sites.py: add_bed / from_annotation, which the reference goldens pin): same columns on every golden input and on files with
the odd lines real files have; and files they must refuse (the caller then reads line by line)."""
import glob
import os

import numpy as np

from conftest import GOLDEN
from spliser_amd import native, sites


def _python_bed(path):
    out = []
    with open(path) as fh:
        for line in fh:
            v = line.split("\t")
            if len(v) != 12:
                continue
            flank = v[10].split(",")
            out.append((v[0], int(v[1]) + int(flank[0]), int(v[2]) - int(flank[1]), int(v[4]), v[5]))
    return out


def _native_bed(path):
    c = native.read_bed_columns(path)
    if c is None:
        return None
    return [(c.chrom_names[k], int(l), int(r), int(a), chr(s) if s else "")
            for k, l, r, a, s in zip(c.chrom.tolist(), c.left.tolist(), c.right.tolist(), c.alpha.tolist(), c.strand.tolist())]


def _bins_as_lists(bins):
    return bins.chrom_index, {c: [(g.name, g.left, g.right, g.strand) for g in bins.genes[c]] for c in bins.chrom_index}, bins.n_created


def test_native_bed_and_gff_match_python_on_every_golden_input():
    beds = sorted(glob.glob(os.path.join(GOLDEN, "**", "*.bed"), recursive=True))
    gffs = sorted(glob.glob(os.path.join(GOLDEN, "**", "*.gff"), recursive=True))
    assert len(beds) >= 10 and len(gffs) >= 3
    for bed in beds:
        assert _native_bed(bed) == _python_bed(bed), bed
    for gff in gffs:
        fast = sites.GeneBins.from_annotation(gff)
        assert fast._columns is not None          # the native reader took it
        sites.USE_NATIVE_TEXT = False
        try:
            slow = sites.GeneBins.from_annotation(gff)
        finally:
            sites.USE_NATIVE_TEXT = True
        cols = {c: fast.gene_arrays(c) for c in fast.chrom_index}
        assert _bins_as_lists(fast) == _bins_as_lists(slow), gff
        for c, (l, r, s, names) in cols.items():  # the column form, taken before the objects were asked for
            assert (l.tolist(), r.tolist(), names) == ([g.left for g in slow.genes[c]], [g.right for g in slow.genes[c]], [g.name for g in slow.genes[c]])
            assert s.tolist() == [43 if g.strand == "+" else (45 if g.strand == "-" else 0) for g in slow.genes[c]]


def test_native_readers_on_odd_lines(tmp_path):
    bed = str(tmp_path / "odd.bed")
    with open(bed, "w", newline="") as fh:
        fh.write('track name=junctions\n')
        fh.write("c1\t100\t300\tJ1\t7\t+\t100\t300\t0\t2\t10,20\t0,180\n")
        fh.write("c2\t 5 \t900\tJ2\t+3\t\t5\t900\t0\t2\t1,2,\t0,1\r\n")          # blanks, sign, empty strand, trailing comma, CRLF
        fh.write("c1\t100\t300\tJ3\t1\t-\t100\t300\t0\t2\t10,20\n")                # 11 columns: skipped
        fh.write("c1\t100\t300\tJ4\t1\t-\t100\t300\t0\t2\t10,20\t0,1\textra\n")    # 13 columns: skipped
        fh.write("c1\t10\t3000000000\tJ5\t0\t?\t1\t2\t0\t2\t0,0\t0,1")             # no newline at the end
    assert _native_bed(bed) == _python_bed(bed) == [("c1", 110, 280, 7, "+"), ("c2", 6, 898, 3, ""), ("c1", 10, 3000000000, 0, "?")]
    gff = str(tmp_path / "odd.gff")
    with open(gff, "w", newline="") as fh:
        fh.write("##gff-version 3\n\n   \n")
        fh.write("c1\tsrc\tgene\t51\t320\t.\t+\t.\tID=G1;Name=x\n")
        fh.write("c1\tsrc\tmRNA\t51\t320\t.\t+\t.\tID=G1.1\n")
        fh.write('c2\tsrc\tgene\t5000\t6000\t.\t-\t.\t gene_id "G 2" ; other "y"\r\n')    # GTF style, quotes, CRLF
        fh.write("c1\tsrc\tgene\t10\t20\t.\t.\t.\tG3\n")                                    # bare name, strand '.'
        fh.write("c1\tsrc\tgene\t10\t20\t.\t+\n")                                           # 7 columns: skipped
        fh.write("c1\tsrc\tgene\t51\t99\t.\t-\t.\tID=G4\n")                                 # same left as G1: after it
    fast = sites.GeneBins.from_annotation(gff)
    assert fast._columns is not None
    sites.USE_NATIVE_TEXT = False
    try:
        slow = sites.GeneBins.from_annotation(gff)
    finally:
        sites.USE_NATIVE_TEXT = True
    assert _bins_as_lists(fast) == _bins_as_lists(slow)
    assert [g.name for g in slow.genes["c1"]] == ["G3", "G1", "G4"] and slow.genes["c2"][0].name == "G 2"


def test_native_readers_refuse_what_they_cannot_vouch_for(tmp_path):
    def bed_with(line):
        p = str(tmp_path / "x.bed")
        with open(p, "w", newline="") as fh:
            fh.write(line)
        return p
    assert native.read_bed_columns(bed_with("c1\t1_0\t300\tJ\t7\t+\t1\t3\t0\t2\t10,20\t0,1\n")) is None      # int("1_0") is 10 in Python
    assert native.read_bed_columns(bed_with("c1\t10\t300\tJ\t7\t+-\t1\t3\t0\t2\t10,20\t0,1\n")) is None     # two-character strand
    assert native.read_bed_columns(bed_with("c1\t10\t300\tJ\t7\t+\t1\t3\t0\t2\t10\t0,1\n")) is None          # one block size (Python: IndexError)
    assert native.read_bed_columns(bed_with("c1\t10\t300\tJ\t7.0\t+\t1\t3\t0\t2\t10,20\t0,1\n")) is None     # Python: ValueError
    assert native.read_bed_columns(bed_with("c1\t10\t300\tJ\t7\t+\t1\t3\t0\t2\t10,20\t0,1\rc2\n")) is None   # a lone carriage return ends a line in Python
    empty = native.read_bed_columns(bed_with(""))
    assert empty is not None and empty.chrom.shape[0] == 0
    with np.testing.assert_raises(native.SpliserNativeError):
        native.read_bed_columns(str(tmp_path / "missing.bed"))
