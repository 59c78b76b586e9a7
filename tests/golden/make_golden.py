#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Container-only script (needs /root/reference).  For every case it writes the inputs
(``reads.sam``, ``junctions.bed`` and optionally ``genes.gff``) and then runs the unmodified
``SpliSER_v0_1_8.py process`` through oracle/refharness (HTSeq stub + samtools shim, see the
docstrings there for why those two boundaries are unpinned) once per flag variant, storing the
reference's ``.SpliSER.tsv`` verbatim as ``expected.<variant>.tsv``.

Fixtures are data only: inputs we synthesise here plus outputs the reference computed.
Re-run with:  python tests/golden/make_golden.py            (rewrites every case)
              python tests/golden/make_golden.py --check    (regenerates to a temp dir and diffs)
"""
import argparse
import filecmp
import json
import os
import random
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refharness"))
import run_reference  # noqa: E402

# --------------------------------------------------------------------------------------
# input writers


def sam_line(i, flag, chrom, pos, cigar):
    return "r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\n" % (i, flag, chrom, pos, cigar)


def bed_line(chrom, left, right, score, strand, a=10, b=10, name="JUNC"):
    start, end = left - a, right + b
    return "%s\t%d\t%d\t%s\t%d\t%s\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n" % (
        chrom, start, end, name, score, strand, start, end, a, b, end - start - b)


def write_case(dirpath, reads, juncs, gff=None, bed_header=None):
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "reads.sam"), "w") as fh:
        chroms = []
        for r in reads:
            if r[1] not in chroms:
                chroms.append(r[1])
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for c in chroms:
            fh.write("@SQ\tSN:%s\tLN:100000000\n" % c)
        for i, (flag, chrom, pos, cigar) in enumerate(reads):
            fh.write(sam_line(i, flag, chrom, pos, cigar))
    with open(os.path.join(dirpath, "junctions.bed"), "w") as fh:
        if bed_header:
            fh.write(bed_header)
        for j in juncs:
            fh.write(bed_line(*j))
    if gff is not None:
        with open(os.path.join(dirpath, "genes.gff"), "w") as fh:
            fh.write("##gff-version 3\n")
            for (chrom, ftype, start, end, strand, attrs) in gff:
                fh.write("%s\tsynth\t%s\t%d\t%d\t.\t%s\t.\t%s\n" % (chrom, ftype, start, end, strand, attrs))


# --------------------------------------------------------------------------------------
# hand-written known-answer cases + CIGAR corner cases

C1 = "Chr1"


def case_kat1():
    juncs = [(C1, 100, 200, 4, "?"), (C1, 100, 300, 2, "?"), (C1, 250, 300, 1, "?")]
    reads = [(0, C1, 51, "50M100N50M")] * 3 + [(0, C1, 51, "50M200N50M")] * 2 + [
        (0, C1, 51, "50M100N50M50N50M"), (0, C1, 151, "100M"), (0, C1, 81, "50M"), (16, C1, 191, "20M")]
    return dict(reads=reads, juncs=juncs,
                variants={"default": {}, "cryptic": {"cryptic": True}})


def case_kat2():
    juncs = [(C1, 100, 200, 2, "+"), (C1, 100, 300, 1, "+"), (C1, 400, 500, 1, "-")]
    reads = [(0, C1, 81, "20M100N30M")] * 2 + [
        (0, C1, 81, "20M200N30M"), (16, C1, 381, "20M100N30M"), (0, C1, 181, "40M"), (16, C1, 181, "40M"),
        (0, C1, 91, "10M2I10M"), (0, C1, 95, "6M1D20M"), (99, C1, 385, "30M"), (147, C1, 385, "30M"),
        (0, C1, 50, "10M400N10M")]
    reads = sorted(reads, key=lambda r: r[2])
    return dict(reads=reads, juncs=juncs,
                variants={"fr_cryptic": {"stranded": "fr", "cryptic": True},
                          "rf_cryptic": {"stranded": "rf", "cryptic": True},
                          "fr": {"stranded": "fr"}, "unstranded": {}})


def case_kat3():
    juncs = [(C1, 100, 200, 4, "?"), (C1, 100, 300, 2, "?"), (C1, 250, 300, 1, "?")]
    reads = [(2048, C1, 81, "50M"), (1024, C1, 81, "50M"), (256, C1, 81, "50M"), (4, C1, 90, "*"),
             (0, C1, 101, "50M")]
    return dict(reads=reads, juncs=juncs, variants={"default": {}, "cryptic": {"cryptic": True}})


def case_kat4():
    c = case_kat1()
    c["gff"] = [(C1, "gene", 51, 320, "+", "ID=G1;Name=x"), (C1, "mRNA", 51, 320, "+", "ID=G1.1;Parent=G1"),
                (C1, "gene", 5000, 6000, "-", "ID=G2")]
    c["variants"] = {"annot": {"gff": True}, "annot_gene": {"gff": True, "chrom": C1, "gene": "G1", "max_intron": 100},
                     "annot_chrom_cryptic": {"gff": True, "chrom": C1, "cryptic": True}}
    return c


def case_kat5():
    juncs = [(C1, 100, 200, 2, "+"), (C1, 100, 300, 3, "-"), (C1, 100, 200, 1, "+")]
    reads = [(0, C1, 81, "50M"), (16, C1, 81, "50M")]
    return dict(reads=reads, juncs=juncs, bed_header='track name=junctions description="synthetic"\n',
                variants={"unstranded": {}, "fr": {"stranded": "fr"}, "rf": {"stranded": "rf"},
                          "fr_cryptic": {"stranded": "fr", "cryptic": True}})


def case_cigar_corners():
    """Every op class, unmerged blocks, zero-length ops, adjacent N ops, unmapped-with-CIGAR."""
    juncs = [(C1, 1000, 1100, 5, "+"), (C1, 1000, 1200, 3, "+"), (C1, 1150, 1200, 2, "+"),
             (C1, 1300, 1400, 7, "+"), (C1, 1350, 1400, 1, "+"), (C1, 1300, 1450, 2, "+"),
             (C1, 1500, 1600, 4, "-"), (C1, 1600, 1700, 4, "-"), (C1, 2000, 2100, 0, "+")]
    reads = [
        (0, C1, 951, "5S50M100N50M3S"), (0, C1, 951, "5H50M100N50M"), (16, C1, 951, "50=100N25=1X24="),
        (0, C1, 951, "50M200N50M"), (0, C1, 951, "50M100N50M50N50M"), (0, C1, 951, "50M100N50M50N50M100N50M"),
        (0, C1, 990, "10M1I5M"), (0, C1, 990, "11M1D9M"), (0, C1, 990, "11M1P9M"), (0, C1, 990, "5M0N20M"),
        (0, C1, 990, "11M0D30M"), (0, C1, 995, "6M100N0M50N20M"), (0, C1, 995, "6M100N50N20M"),
        (0, C1, 1000, "1M100N20M"), (0, C1, 1001, "100N20M"), (0, C1, 1001, "1S100N20M"),
        (4, C1, 1000, "60M"), (4, C1, 999, "60M"), (4, C1, 1000, "*"), (0, C1, 1000, "*"), (0, C1, 1000, "20S"),
        (0, C1, 1051, "50M"), (0, C1, 1100, "1M"), (0, C1, 1100, "2M"), (0, C1, 1099, "2M"), (0, C1, 1101, "20M"),
        (0, C1, 1140, "11M49N30M"), (0, C1, 1140, "11M49N101M99N20M"), (0, C1, 1251, "50M100N50M"),
        (0, C1, 1251, "50M150N50M"), (16, C1, 1251, "100M50N40M"), (0, C1, 1251, "50M100N30M"),
        (0, C1, 1290, "200M"), (0, C1, 1290, "11M200N10M"), (1024, C1, 1290, "11M300N10M"),
        (0, C1, 1451, "50M100N50M"), (16, C1, 1451, "50M100N50M"), (83, C1, 1451, "50M100N1M99N20M"),
        (163, C1, 1451, "50M100N1M99N20M"), (99, C1, 1451, "50M200N20M"), (147, C1, 1451, "50M200N20M"),
        (0, C1, 1451, "250M"), (16, C1, 1451, "250M"), (0, C1, 1990, "11M100N10M"), (0, C1, 1990, "30M"),
        (0, C1, 2090, "30M"), (0, C1, 900, "10M2000N10M"), (16, C1, 900, "10M2000N10M"),
    ]
    reads = sorted(reads, key=lambda r: r[2])
    return dict(reads=reads, juncs=juncs,
                variants={"unstranded": {}, "cryptic": {"cryptic": True}, "fr": {"stranded": "fr"},
                          "rf_cryptic": {"stranded": "rf", "cryptic": True}})


def case_multichrom():
    """chrom_index order = GFF first-appearance then BED first-appearance; -c filter."""
    juncs = [("ChrB", 500, 700, 3, "+"), ("ChrA", 100, 200, 4, "-"), ("ChrB", 500, 800, 1, "+"),
             ("ChrC", 50, 90, 2, "+"), ("ChrA", 150, 200, 1, "-"), ("ChrB", 650, 800, 2, "+")]
    reads = [(0, "ChrA", 51, "50M100N50M"), (16, "ChrA", 101, "50M50N10M"), (0, "ChrA", 120, "100M"),
             (0, "ChrB", 451, "50M200N50M"), (0, "ChrB", 451, "50M300N50M"), (0, "ChrB", 601, "50M150N50M"),
             (0, "ChrB", 480, "300M"), (16, "ChrB", 690, "30M"), (0, "ChrC", 41, "10M40N10M"), (0, "ChrC", 45, "30M"),
             (0, "ChrD", 45, "30M")]
    gff = [("ChrC", "gene", 1, 1000, "+", "ID=GC1"), ("ChrA", "gene", 90, 210, "-", "ID=GA1;Name=a"),
           ("ChrA", "gene", 1, 95, "+", "ID=GA0"), ("ChrB", "gene", 400, 900, "+", "ID=GB1")]
    return dict(reads=reads, juncs=juncs, gff=gff,
                variants={"noannot": {}, "annot": {"gff": True}, "annot_chrB": {"gff": True, "chrom": "ChrB"},
                          "chrA_fr_cryptic": {"chrom": "ChrA", "stranded": "fr", "cryptic": True},
                          "annot_fr": {"gff": True, "stranded": "fr"}})


# --------------------------------------------------------------------------------------
# seeded random cases: alt 5'/3' sites, exon skipping, novel junctions, indels, paired flags

FLAGS = [0, 16, 0, 16, 99, 147, 83, 163, 256, 272, 1024, 1040, 2048, 2064, 65, 129, 81, 161, 4, 20]


def random_case(seed, n_genes=6, n_reads=900, chroms=("Chr1", "Chr2"), strands="+-"):
    rng = random.Random(seed)
    juncs, reads, gff = [], [], []
    for chrom in chroms:
        cursor = 500
        for g in range(n_genes):
            strand = rng.choice(strands)
            n_exons = rng.randint(2, 6)
            exons = []
            pos = cursor
            for e in range(n_exons):
                elen = rng.randint(40, 260)
                exons.append((pos, pos + elen - 1))          # 1-based inclusive exon
                pos += elen + rng.randint(60, 900)
            cursor = pos + rng.randint(100, 2000)
            gff.append((chrom, "gene", exons[0][0], exons[-1][1], strand, "ID=G%s_%d" % (chrom, g)))
            gj = set()
            for i in range(n_exons - 1):
                left, right = exons[i][1], exons[i + 1][0] - 1      # site convention: last exonic / last intronic
                gj.add((left, right))
                if rng.random() < 0.35:                               # alternative 5' (left) site
                    gj.add((left - rng.randint(3, 30), right))
                if rng.random() < 0.35:                               # alternative 3' (right) site
                    gj.add((left, right + rng.randint(3, 30)))
                if i + 2 < n_exons and rng.random() < 0.4:            # exon skipping
                    gj.add((left, exons[i + 2][0] - 1))
            gj = sorted(gj)
            rng.shuffle(gj)
            for (left, right) in gj:
                juncs.append((chrom, left, right, rng.choice([0, 1, 2, 3, 5, 8, 13, 40]), strand))
                if rng.random() < 0.1:                                # duplicated junction line
                    juncs.append((chrom, left, right, rng.randint(1, 4), strand))
            span0, span1 = exons[0][0] - 80, exons[-1][1] + 80
            known = sorted(gj)
            for _ in range(n_reads // (n_genes * len(chroms))):
                flag = rng.choice(FLAGS)
                kind = rng.random()
                start = rng.randint(span0, span1)
                if kind < 0.35:                                       # unspliced, maybe with indel / clip
                    ln = rng.randint(20, 150)
                    c = rng.random()
                    if c < 0.7:
                        cigar = "%dM" % ln
                    elif c < 0.8:
                        a = rng.randint(1, ln - 1)
                        cigar = "%dM%dI%dM" % (a, rng.randint(1, 3), ln - a)
                    elif c < 0.9:
                        a = rng.randint(1, ln - 1)
                        cigar = "%dM%dD%dM" % (a, rng.randint(1, 3), ln - a)
                    else:
                        cigar = "%dS%dM%dS" % (rng.randint(1, 9), ln, rng.randint(1, 9))
                    reads.append((flag, chrom, start, cigar))
                else:                                                 # spliced: 1..3 junctions
                    nj = 1 if kind < 0.8 else (2 if kind < 0.95 else 3)
                    left, right = rng.choice(known)
                    if rng.random() < 0.15:                           # novel junction near a known one
                        left += rng.choice([-7, -2, 2, 5])
                    if rng.random() < 0.15:
                        right += rng.choice([-6, -1, 3, 9])
                        if right <= left:
                            right = left + 20
                    pre = rng.randint(1, 90)
                    ops = ["%dM" % pre, "%dN" % (right - left)]
                    pos1 = left - pre + 1
                    cur = right + 1
                    for extra in range(nj - 1):
                        nxt = [j for j in known if j[0] > cur + 2]
                        if not nxt:
                            break
                        l2, r2 = rng.choice(nxt[:3])
                        ops.append("%dM" % (l2 - cur + 1))
                        ops.append("%dN" % (r2 - l2))
                        cur = r2 + 1
                    ops.append("%dM" % rng.randint(1, 90))
                    if pos1 >= 1:
                        reads.append((flag, chrom, pos1, "".join(ops)))
    order = {c: i for i, c in enumerate(chroms)}
    reads.sort(key=lambda r: (order[r[1]], r[2]))
    return reads, juncs, gff


def case_random(seed, **kw):
    reads, juncs, gff = random_case(seed, **kw)
    return dict(reads=reads, juncs=juncs, gff=gff,
                variants={"unstranded": {}, "cryptic": {"cryptic": True}, "fr_cryptic": {"stranded": "fr", "cryptic": True},
                          "rf": {"stranded": "rf"}, "annot_fr": {"gff": True, "stranded": "fr"}})


def case_single_gene():
    """BASELINE config 1 shape: -c Chr1 -g <gene> -m 6000 with a one-gene query."""
    reads, juncs, gff = random_case(101, n_genes=8, n_reads=1600, chroms=("Chr1",), strands="-")
    gff = [(c, t, s, e, st, a.replace("ID=GChr1_3", "ID=AT1G01060")) for (c, t, s, e, st, a) in gff]
    return dict(reads=reads, juncs=juncs, gff=gff,
                variants={"gene": {"gff": True, "chrom": "Chr1", "gene": "AT1G01060", "max_intron": 6000},
                          "gene_fr_cryptic": {"gff": True, "chrom": "Chr1", "gene": "AT1G01060", "max_intron": 6000,
                                              "stranded": "fr", "cryptic": True},
                          "all_annot": {"gff": True}})


def case_odd_strands():
    """What a careful pipeline never feeds `process` and the reference processes all the same: BED strands that are neither '+' nor
    '-' IN A STRANDED ANALYSIS (such a junction's look-ups take whichever site the reference's bisection lands on, :198, and a
    new '?' site goes among the '+' / '-' sites of its position where bisect.insort under Site.__lt__ puts it), and junctions whose
    two ends coincide (both look-ups precede both insertions, :291-292: two sites at one position from one line) -- positions
    that hold several Site objects, partner lists that name a position twice."""
    reads, juncs, gff = random_case(13, n_genes=5, n_reads=700, chroms=("Chr1",))
    rng = random.Random(1313)
    odd = []
    for (c, l, r, sc, st) in juncs:
        odd.append((c, l, r, sc, st))
        if rng.random() < 0.45:       # the same junction, or one sharing an end, once more with a strand that is none
            odd.append((c, l, r if rng.random() < 0.6 else r + rng.choice([4, 9]), rng.randint(0, 5), rng.choice("?.")))
        if rng.random() < 0.12:
            odd.append((c, l, l, rng.randint(1, 4), rng.choice("+-?")))
    rng.shuffle(odd)
    return dict(reads=reads, juncs=odd, gff=gff,
                variants={"fr": {"stranded": "fr"}, "rf_cryptic": {"stranded": "rf", "cryptic": True}, "unstranded_cryptic": {"cryptic": True},
                          "annot_fr_cryptic": {"gff": True, "stranded": "fr", "cryptic": True}})


CASES = {
    "odd_strands": case_odd_strands,
    "kat1": case_kat1, "kat2": case_kat2, "kat3": case_kat3, "kat4": case_kat4, "kat5": case_kat5,
    "cigar_corners": case_cigar_corners, "multichrom": case_multichrom,
    "random_a": lambda: case_random(7), "random_b": lambda: case_random(8, n_genes=10, n_reads=2400, chroms=("Chr1", "Chr2", "ChrM")),
    "random_unstranded_q": lambda: case_random(9, strands="?"),
    "single_gene": case_single_gene,
}


def build(outroot, names=None, cross_check=True):
    manifest = {}
    for name, fn in CASES.items():
        if names and name not in names:
            continue
        case = fn()
        d = os.path.join(outroot, name)
        write_case(d, case["reads"], case["juncs"], gff=case.get("gff"), bed_header=case.get("bed_header"))
        manifest[name] = {}
        for vname, v in case["variants"].items():
            v = dict(v)
            kw = dict(chrom=v.get("chrom"), gene=v.get("gene"), max_intron=v.get("max_intron"),
                      stranded=v.get("stranded"), cryptic=v.get("cryptic", False),
                      gff=os.path.join(d, "genes.gff") if v.get("gff") else None)
            tmp = tempfile.mkdtemp()
            try:
                text, _ = run_reference.run_process(os.path.join(d, "reads.sam"), os.path.join(d, "junctions.bed"),
                                                    os.path.join(tmp, "out"), inprocess=True,
                                                    dump_json=os.path.join(d, "expected.%s.json" % vname), **kw)
                if cross_check and len(case["juncs"]) <= 12:
                    # the fake-samtools-on-PATH route (a real child process per site) must agree
                    text2, _ = run_reference.run_process(os.path.join(d, "reads.sam"), os.path.join(d, "junctions.bed"),
                                                         os.path.join(tmp, "out2"), inprocess=False, **kw)
                    assert text == text2, "in-process replay and subprocess shim disagree for %s/%s" % (name, vname)
            finally:
                shutil.rmtree(tmp)
            with open(os.path.join(d, "expected.%s.tsv" % vname), "w") as fh:
                fh.write(text)
            manifest[name][vname] = v
            print("golden %-22s %-22s %4d rows" % (name, vname, text.count("\n") - 1))
    return manifest


# --------------------------------------------------------------------------------------
# combine / output goldens: three samples of one gene model with sample-specific junctions and reads


def combine_case(seed, n_samples=3):
    reads, juncs, gff = random_case(seed, n_genes=7, n_reads=1500, chroms=("Chr1", "Chr2"))
    rng = random.Random(seed * 7 + 1)
    samples = []
    for k in range(n_samples):
        jk = [(c, l, r, max(0, sc + rng.randint(-2, 6)), st) for (c, l, r, sc, st) in juncs if rng.random() < 0.72]
        rk = [r for r in reads if rng.random() < 0.7]
        samples.append((rk, jk))
    return samples, gff


COMBINE_VARIANTS = {
    "unstranded": dict(process={}, combine=[]),
    "fr_cryptic": dict(process={"stranded": "fr", "cryptic": True}, combine=["--isStranded", "-s", "fr", "--beta2Cryptic"]),
    "annot_gene": dict(process={"gff": True}, combine=["-g", "GChr1_2"]),
    "rf_crypticlate": dict(process={"stranded": "rf"}, combine=["--isStranded", "-s", "rf", "--beta2Cryptic"]),
    # combineShallow: evidence filter (command name is part of the variant)
    "shallow": dict(process={}, combine=["-m", "2", "-r", "4", "-e", "0.2"], command="combineShallow"),
    "shallow_fr": dict(process={"stranded": "fr", "cryptic": True}, command="combineShallow",
                       combine=["--isStranded", "-s", "fr", "--beta2Cryptic", "-m", "3", "-r", "2"]),
    "shallow_gene": dict(process={"gff": True}, combine=["-g", "GChr2_1", "-m", "1", "-r", "1"], command="combineShallow"),
}


def build_combine(outroot, name="combine_a", seed=21):
    samples, gff = combine_case(seed)
    d = os.path.join(outroot, name)
    os.makedirs(d, exist_ok=True)
    for k, (reads, juncs) in enumerate(samples):
        write_case(os.path.join(d, "sample%d" % k), reads, juncs, gff=gff)
    manifest = {}
    for vname, v in COMBINE_VARIANTS.items():
        tmp = tempfile.mkdtemp()
        try:
            lines = []
            for k in range(len(samples)):
                sd = os.path.join(d, "sample%d" % k)
                kw = dict(stranded=v["process"].get("stranded"), cryptic=v["process"].get("cryptic", False),
                          gff=os.path.join(sd, "genes.gff") if v["process"].get("gff") else None)
                text, _ = run_reference.run_process(os.path.join(sd, "reads.sam"), os.path.join(sd, "junctions.bed"),
                                                    os.path.join(tmp, "s%d" % k), inprocess=True, **kw)
                with open(os.path.join(sd, "expected.%s.tsv" % vname), "w") as fh:
                    fh.write(text)
                lines.append("S%d\t%s\t%s\n" % (k, os.path.join(tmp, "s%d.SpliSER.tsv" % k), os.path.join(sd, "reads.sam")))
            sfile = os.path.join(tmp, "samples.tsv")
            with open(sfile, "w") as fh:
                fh.writelines(lines)
            rc, log = run_reference.run_cli([v.get("command", "combine"), "-S", sfile, "-o", os.path.join(tmp, "all")] + v["combine"], inprocess=True)
            assert rc == 0, log
            shutil.copy(os.path.join(tmp, "all.combined.tsv"), os.path.join(d, "expected.%s.combined.tsv" % vname))
            # output -t DiffSpliSER / GWAS on the reference's own combined file
            rc, log = run_reference.run_cli(["output", "-S", sfile, "-C", os.path.join(tmp, "all.combined.tsv"), "-t", "DiffSpliSER",
                                             "-o", os.path.join(tmp, "diff_"), "-r", "5"], inprocess=True)
            assert rc == 0, log
            shutil.copy(os.path.join(tmp, "diff_All.DiffSpliSER.tsv"), os.path.join(d, "expected.%s.DiffSpliSER.tsv" % vname))
            gdir = os.path.join(tmp, "gwas") + os.sep
            os.makedirs(gdir)
            rc, log = run_reference.run_cli(["output", "-S", sfile, "-C", os.path.join(tmp, "all.combined.tsv"), "-t", "GWAS",
                                             "-o", gdir, "-r", "5", "-m", "2"], inprocess=True)
            assert rc == 0, log
            files = {f: open(os.path.join(gdir, f)).read() for f in sorted(os.listdir(gdir))}
            with open(os.path.join(d, "expected.%s.GWAS.json" % vname), "w") as fh:
                json.dump(files, fh, indent=0, sort_keys=True)
            manifest[vname] = v
            print("golden %-22s %-22s %4d combined lines, %d GWAS files" % (
                name, vname, open(os.path.join(d, "expected.%s.combined.tsv" % vname)).read().count("\n") - 1, len(files)))
        finally:
            shutil.rmtree(tmp)
    with open(os.path.join(d, "combine_manifest.json"), "w") as fh:
        json.dump({"n_samples": len(samples), "variants": manifest}, fh, indent=1, sort_keys=True)


# --------------------------------------------------------------------------------------
# junction goldens: the BED12 file is the one the build's own `junctions` command wrote ON THE GPU from
# the reads of another case (there is no regtools here to hold it against) -- what pins it is the real reference: its
# findAlphaCounts must read that file (SpliSER_v0_1_8.py:259-277) into the sites, alpha counts and partners the build's table has.
#   stage 1 (GPU box):   python tests/golden/make_golden.py --junction-beds gpurun_out/junction_beds
#   stage 2 (container): copy each <case>.bed to tests/golden/<case>/junctions.bed, then the usual run (names: junctions_u junctions_fr)
JUNCTION_CASES = {
    "junctions_u": dict(reads_of="random_b", junctions={}, variants={"unstranded": {}, "cryptic": {"cryptic": True}}),
    "junctions_fr": dict(reads_of="multichrom", junctions=dict(isStranded=True, strandedType="fr"),
                         variants={"fr": {"stranded": "fr"}, "fr_cryptic": {"stranded": "fr", "cryptic": True}}),
}
JUNCTION_KNOBS = dict(minAnchor=1, minIntron=1, maxIntron=10 ** 9)   # (every N op: the goldens' reads are short)


def write_junction_beds(outdir):
    """Stage 1, on the GPU box: the `junctions` command on the reads of the source cases."""
    sys.path.insert(0, ROOT)
    from spliser_amd.junctions import junctions
    os.makedirs(outdir, exist_ok=True)
    for name, case in JUNCTION_CASES.items():
        n = junctions(os.path.join(HERE, case["reads_of"], "reads.sam"), os.path.join(outdir, name + ".bed"), log=lambda m: None,
                      **dict(JUNCTION_KNOBS, **case["junctions"]))
        print("%s: %d junctions from the reads of %s" % (name, n, case["reads_of"]))


def build_junction_cases(outroot, names=None):
    manifest = {}
    for name, case in JUNCTION_CASES.items():
        if names and name not in names:
            continue
        bed = os.path.join(HERE, name, "junctions.bed")
        if not os.path.exists(bed):
            print("junction case %s: no junctions.bed yet (stage 1 runs on the GPU box)" % name)
            continue
        d = os.path.join(outroot, name)
        os.makedirs(d, exist_ok=True)
        if os.path.abspath(d) != os.path.abspath(os.path.join(HERE, name)):
            shutil.copy(bed, os.path.join(d, "junctions.bed"))
        shutil.copy(os.path.join(HERE, case["reads_of"], "reads.sam"), os.path.join(d, "reads.sam"))
        manifest[name] = {}
        for vname, v in case["variants"].items():
            tmp = tempfile.mkdtemp()
            try:
                text, _ = run_reference.run_process(os.path.join(d, "reads.sam"), os.path.join(d, "junctions.bed"), os.path.join(tmp, "out"), inprocess=True,
                                                    dump_json=os.path.join(d, "expected.%s.json" % vname), stranded=v.get("stranded"), cryptic=v.get("cryptic", False))
            finally:
                shutil.rmtree(tmp)
            with open(os.path.join(d, "expected.%s.tsv" % vname), "w") as fh:
                fh.write(text)
            manifest[name][vname] = dict(v)
            print("golden %-22s %-22s %4d rows" % (name, vname, text.count("\n") - 1))
    return manifest


def diff_tree(fresh, committed, label):
    """Every file of a regenerated case against the committed one, sub-directories included.  -> number of differences."""
    bad = 0
    cmp = filecmp.dircmp(fresh, committed)
    if cmp.diff_files or cmp.left_only or cmp.right_only or cmp.funny_files:
        print("MISMATCH", label, cmp.diff_files, cmp.left_only, cmp.right_only)
        bad += 1
    for f in cmp.common_files:      # (dircmp compares by stat signature first: make sure of the bytes)
        if open(os.path.join(fresh, f), "rb").read() != open(os.path.join(committed, f), "rb").read():
            print("MISMATCH", label, f)
            bad += 1
    for sub in cmp.common_dirs:
        bad += diff_tree(os.path.join(fresh, sub), os.path.join(committed, sub), label + "/" + sub)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--junction-beds", metavar="DIR", help="stage 1 of the junction goldens: run on the GPU box, writes <case>.bed into DIR")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.junction_beds:
        write_junction_beds(a.junction_beds)
        return
    if not run_reference.reference_available():
        sys.exit("reference not available at %s" % run_reference.REFERENCE_DIR)
    if a.check:
        tmp = tempfile.mkdtemp()
        try:
            names = [n for n in a.names if n != "combine_a"]
            if names or not a.names:
                build(tmp, names or None, cross_check=False)
                build_junction_cases(tmp, names or None)
            if not a.names or "combine_a" in a.names:
                build_combine(tmp)          # process x 3 samples -> combine / combineShallow x 7 variants -> output x 2
            bad = 0
            for name in sorted(os.listdir(tmp)):
                bad += diff_tree(os.path.join(tmp, name), os.path.join(HERE, name), name)
            sys.exit(1 if bad else 0)
        finally:
            shutil.rmtree(tmp)
    if not a.names or "combine_a" in a.names:
        build_combine(HERE)
        if a.names == ["combine_a"]:
            return
    manifest = build(HERE, [n for n in a.names if n != "combine_a"] or None)
    manifest.update(build_junction_cases(HERE, [n for n in a.names if n != "combine_a"] or None))
    mpath = os.path.join(HERE, "manifest.json")
    old = {}
    if a.names and os.path.exists(mpath):
        with open(mpath) as fh:
            old = json.load(fh)
    old.update(manifest)
    with open(mpath, "w") as fh:
        json.dump(old, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
