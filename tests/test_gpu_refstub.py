"""INTEGRATION.md's stub (spliser_amd/refstub.py) executed END TO END on the GPU box: its ctypes calls -- spl_create, spl_bam_open,
spl_bam_reads, spl_count, spl_sse -- against stand-ins for the reference's Site objects that offer exactly the accessors the stub
uses (Gene_Site_Iter_Graph_v0_1_8.py:98-339; the real objects cannot travel, tests/test_refstub_reference.py runs the marshalling
on them in the build container).  What the adders receive has to be what the reference's goldens hold."""
import os

import numpy as np
import pytest

import helpers
from conftest import golden_cases
from spliser_amd import native, refstub, samio

pytestmark = pytest.mark.gpu


class _Site(object):
    """The reference's Site, as far as processSites and the stub touch it."""

    def __init__(self, pos, strand, alpha):
        self.pos, self.strand, self.alpha = pos, strand, [alpha]
        self.partners, self.partner_counts, self.competitors = [], {}, []
        self.beta1, self.b2s, self.b2c, self.b2w, self.sse = [0], [0], [0], [0.0], [0.0]

    def getPos(self): return self.pos
    def getStrand(self): return self.strand
    def getPartners(self): return self.partners
    def getPartnerCounts(self): return self.partner_counts
    def getCompetitorPos(self): return self.competitors
    def getAlphaCount(self, sample): return self.alpha[sample]
    def addBeta1Count(self, v, sample): self.beta1[sample] += v
    def addBeta2SimpleCount(self, v, sample): self.b2s[sample] += v
    def addBeta2CrypticCount(self, v, sample): self.b2c[sample] += v
    def updateBeta2Weighted(self, values): self.b2w = values
    def getBeta2WeightedCounts(self): return self.b2w
    def setSSE(self, v, sample): self.sse[sample] = v


def _sites_of(arr):
    sites = [_Site(int(arr.pos[i]), arr.strand_text[i], int(arr.alpha[i])) for i in range(arr.n)]
    for i, s in enumerate(sites):
        for e in range(int(arr.part_off[i]), int(arr.part_off[i + 1])):
            s.partner_counts[int(arr.part_pos[e])] = [int(arr.edge_cnt[e])]
            if int(arr.part_site[e]) >= 0:
                s.partners.append(sites[int(arr.part_site[e])])
        s.competitors = [int(c) for c in arr.comp_pos[int(arr.comp_off[i]):int(arr.comp_off[i + 1])]]
    return sites


@pytest.mark.parametrize("case,variant,opts", [c for c in golden_cases() if c[0] in ("kat1", "kat2", "kat5", "multichrom", "random_a", "cigar_corners", "odd_strands")])
def test_the_stub_fills_the_sites_like_the_reference(case, variant, opts, tmp_path):
    case_dir = os.path.join(helpers.GOLDEN, case)
    if not os.path.isfile(os.path.join(case_dir, "junctions.bed")) or opts.get("gene"):
        pytest.skip("not a whole-file process case")
    native.build()
    table = helpers.build_table(case_dir, opts)
    names, reads = samio.read_sam(os.path.join(case_dir, "reads.sam"))
    bam = str(tmp_path / "reads.bam")
    samio.write_bam(bam, names, [10 ** 8] * len(names), [(c, reads[c]) for c in names if c in reads], with_seq=True)
    g = {"chrom_index": list(table.chrom_index), "site2D_array": [_sites_of(table.chrom_arrays(c)) for c in table.chrom_index]}
    refstub.install(g, native.LIB_PATH)
    g["processSites"](bam, opts.get("chrom") or "All", bool(opts.get("stranded")), opts.get("stranded"), bool(opts.get("cryptic")))
    _, want = helpers.expected(case, variant)
    by_chrom = {}
    for r in want:                    # (rows in the table's order: two sites may share position AND strand -- a junction whose ends coincide)
        by_chrom.setdefault(r["chrom"], []).append(r)
    n = 0
    for chrom, sites in zip(g["chrom_index"], g["site2D_array"]):
        if not (opts.get("chrom") in (None, chrom)):
            continue
        assert len(sites) == len(by_chrom.get(chrom, []))
        for s, w in zip(sites, by_chrom.get(chrom, [])):
            assert (s.pos, s.strand) == (w["pos"], w["strand"])
            assert (s.beta1[0], s.b2s[0]) == (w["beta1"], w["beta2Simple"]), (chrom, s.pos)
            assert s.b2c[0] == w["beta2Cryptic"] and s.b2w[0] == w["beta2Weighted"]
            assert s.sse[0] == w["sse"] and abs(s.sse[0] - w["sse"]) <= 1e-9     # (bit-identical; 1e-9 is the tolerance BASELINE.json states)
            n += 1
    assert n > 0
