"""The junction goldens (tests/golden/junctions_u, junctions_fr): BED12 files the build's `junctions`
command wrote on the GPU from the reads of other golden cases, and what the REAL reference made of them (its findAlphaCounts
read them into sites, alpha counts and partners: expected.*.tsv, through the manifest every golden test sees).  There is no
regtools here, so this is the anchor the junction table has: here the CPU checker (oracle/oracle.py junction_table) is held to
the same files, the GPU is in tests/test_junctions.py."""
import io
import os

import pytest

import helpers
from oracle import oracle
from spliser_amd import samio
from spliser_amd.junctions import write_junction_bed

import sys
sys.path.insert(0, os.path.join(helpers.GOLDEN))
from make_golden import JUNCTION_CASES, JUNCTION_KNOBS  # noqa: E402  (the cases' definitions: data, no reference needed to import)


@pytest.mark.parametrize("name", sorted(JUNCTION_CASES))
def test_checker_writes_the_bed_the_reference_was_run_on(name):
    case = JUNCTION_CASES[name]
    names, sets = samio.read_sam(os.path.join(helpers.GOLDEN, case["reads_of"], "reads.sam"))
    stranded = {"fr": 1, "rf": 2}[case["junctions"]["strandedType"]] if case["junctions"].get("isStranded") else 0
    out = io.StringIO()
    out.write('track name=junctions description="spliser_amd junctions (a>=%d, %d<=intron<=%d)"\n' % (
        JUNCTION_KNOBS["minAnchor"], JUNCTION_KNOBS["minIntron"], JUNCTION_KNOBS["maxIntron"]))
    total = 0
    for chrom in names:
        reads = sets.get(chrom)
        if reads is None or reads.n == 0:
            continue
        rows = oracle.junction_table(reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, JUNCTION_KNOBS["minAnchor"],
                                     JUNCTION_KNOBS["minIntron"], JUNCTION_KNOBS["maxIntron"])
        import numpy as np
        table = {k: np.array([r[i] for r in rows], np.int64) for i, k in enumerate(("left", "right", "strand", "count", "anchor_left", "anchor_right"))}
        total += write_junction_bed(out, chrom, table, total + 1)
    assert out.getvalue() == open(os.path.join(helpers.GOLDEN, name, "junctions.bed")).read()
    assert total > 0
