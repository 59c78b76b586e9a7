"""htslib region-overlap semantics over SAM text (TEST INFRASTRUCTURE, container only).

Shared by the fake ``samtools`` executable and by the in-process replay used for larger
randomised cases.  See the docstring of ``samtools`` in this directory for the contract and
for why parity is unpinned at this boundary (no samtools / htslib in the image).
"""
import re

import numpy as np

_OP = re.compile(r"(\d+)([MIDNSHP=XB])")


def endpos0(pos0, flag, cigar):
    """htslib bam_endpos(): pos + rlen, with rlen forced to 1 for unmapped / zero-length."""
    rlen = 0
    if not (flag & 4) and cigar != "*":
        for n, op in _OP.findall(cigar):
            if op in "MDN=X":
                rlen += int(n)
    if rlen == 0:
        rlen = 1
    return pos0 + rlen


def parse_region(region):
    chrom, span = region.rsplit(":", 1)
    beg_s, end_s = span.split("-")
    return chrom, int(beg_s) - 1, int(end_s)


def iter_region(path, region):
    chrom, beg0, end0 = parse_region(region)
    with open(path, "r") as handle:
        for line in handle:
            if line.startswith("@"):
                continue
            cols = line.split("\t")
            if len(cols) < 6 or cols[2] != chrom:
                continue
            pos0 = int(cols[3]) - 1
            if pos0 < end0 and endpos0(pos0, int(cols[1]), cols[5]) > beg0:
                yield line if line.endswith("\n") else line + "\n"


class SamIndex(object):
    """All records of one SAM file, grouped per chromosome, answering region queries."""

    def __init__(self, path):
        per = {}
        with open(path, "r") as handle:
            for line in handle:
                if line.startswith("@"):
                    continue
                cols = line.split("\t")
                if len(cols) < 6 or cols[2] == "*":
                    continue
                pos0 = int(cols[3]) - 1
                rec = per.setdefault(cols[2], ([], [], []))
                rec[0].append(pos0)
                rec[1].append(endpos0(pos0, int(cols[1]), cols[5]))
                rec[2].append((line if line.endswith("\n") else line + "\n").encode("ascii"))
        self.per = {c: (np.asarray(p, dtype=np.int64), np.asarray(e, dtype=np.int64), l)
                    for c, (p, e, l) in per.items()}

    def query(self, region):
        chrom, beg0, end0 = parse_region(region)
        if chrom not in self.per:
            return []
        pos, end, lines = self.per[chrom]
        hit = np.nonzero((pos < end0) & (end > beg0))[0]
        return [lines[i] for i in hit]
