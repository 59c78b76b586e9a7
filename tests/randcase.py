"""Seeded adversarial cases: small site tables built through the product's own host code from random junction lists
that force what real data makes rare -- both strands at one position, shared ends, nested / crossing junctions, 0N ops,
adjacent N ops, clipped and indel CIGARs, unmapped-but-placed records, novel junctions next to known ones."""
import os
import tempfile

import numpy as np

from spliser_amd import samio, sites


def make_case(seed, stranded, dirpath=None, odd=False):
    """-> (ChromArrays of the table built from the case's BED file, ReadSet).  With ``dirpath`` the inputs are also left there as
    ``junctions.bed`` and ``reads.sam`` (what the reference takes)."""
    rng = np.random.default_rng(seed)
    n_pos = int(rng.integers(6, 18))
    positions = np.sort(rng.choice(np.arange(100, 100 + 40 * n_pos), n_pos, replace=False))
    juncs = []
    for _ in range(int(rng.integers(4, 22))):
        a, b = np.sort(rng.choice(n_pos, 2, replace=False))
        strand = "+" if rng.random() < 0.5 else "-"
        if not stranded and rng.random() < 0.3:
            strand = "?"
        if odd and rng.random() < 0.35:      # a strand that is neither '+' nor '-' in ANY analysis (a stranded one too: such a
            strand = str(rng.choice(["?", "."]))   # query takes whichever site the reference's bisection lands on, :198)
        l, r = int(positions[a]), int(positions[b])
        if odd and rng.random() < 0.12:      # a junction whose two ends coincide: two sites at one position from one line (:291-292)
            r = l
        juncs.append(("c1", l, r, int(rng.integers(0, 9)), strand))
    tmp = dirpath or tempfile.mkdtemp(prefix="spl_rand_")
    bed = os.path.join(tmp, "junctions.bed" if dirpath else "j.bed")
    with open(bed, "w") as fh:
        for (c, l, r, sc, st) in juncs:
            fh.write("%s\t%d\t%d\tJ\t%d\t%s\t%d\t%d\t0\t2\t10,10\t0,%d\n" % (c, l - 10, r + 10, sc, st, l - 10, r + 10, r - l + 10))
    table = sites.SiteTable(is_stranded=stranded)
    table.add_bed(bed)
    table.find_competitors()
    arr = table.chrom_arrays("c1")
    # reads
    recs = []
    flags = [0, 16, 99, 147, 83, 163, 4, 20, 256, 1024]
    lo, hi = int(positions[0]) - 60, int(positions[-1]) + 60
    for _ in range(int(rng.integers(40, 160))):
        kind = rng.random()
        flag = int(rng.choice(flags))
        if kind < 0.3:
            start = int(rng.integers(lo, hi))
            ln = int(rng.integers(1, 120))
            c = rng.random()
            if c < 0.6:
                cig = "%dM" % ln
            elif c < 0.75:
                a = int(rng.integers(1, ln + 1))
                cig = "%dM%dI%dM" % (a, int(rng.integers(1, 4)), ln + 1 - a)
            elif c < 0.9:
                a = int(rng.integers(1, ln + 1))
                cig = "%dM%dD%dM" % (a, int(rng.integers(0, 4)), ln + 1 - a)
            else:
                cig = "%dS%d=%dX%dH" % (int(rng.integers(1, 5)), ln, int(rng.integers(0, 3)), 2)
            recs.append((flag, start, cig))
        else:
            nj = int(rng.integers(1, 4))
            chosen = sorted(rng.choice(n_pos, min(2 * nj, n_pos - n_pos % 2), replace=False).tolist())
            pairs = [(int(positions[chosen[2 * k]]), int(positions[chosen[2 * k + 1]])) for k in range(len(chosen) // 2)]
            if rng.random() < 0.25:      # novel end next to a known one
                k = int(rng.integers(0, len(pairs)))
                pairs[k] = (pairs[k][0] + int(rng.choice([-1, 1, 2])), pairs[k][1])
                if pairs[k][0] >= pairs[k][1]:
                    pairs[k] = (pairs[k][1] - 3, pairs[k][1])
            pre = int(rng.integers(1, 60))
            ops, cur = ["%dM" % pre], pairs[0][0] + 1
            start = pairs[0][0] - pre + 1
            ok = True
            for k, (l, r) in enumerate(pairs):
                if l + 1 < cur:
                    ok = False
                    break
                if l + 1 > cur:
                    ops.append("%dM" % (l + 1 - cur))
                ops.append("%dN" % (r - l))
                cur = r + 1
                if rng.random() < 0.1:
                    ops.append("0N")
            if not ok or start < 1:
                continue
            ops.append("%dM" % int(rng.integers(0, 60)))
            recs.append((flag, start, "".join(ops)))
    recs.sort(key=lambda r: r[1])
    if dirpath:
        with open(os.path.join(dirpath, "reads.sam"), "w") as fh:
            fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:100000000\n")
            for i, (flag, start, cig) in enumerate(recs):
                fh.write("r%d\t%d\tc1\t%d\t60\t%s\t*\t0\t0\t*\t*\n" % (i, flag, start, cig))
    return arr, samio.ReadSet.from_records(recs)


def query_table(arr, seed):
    """A table as `combine` asks about it (SpliSER_v0_1_8.py:869-904): some of the rows of ``arr``, each with the strand,
    partners and competitors "as they stand" when only some samples have contributed -- a random part of its lists, possibly
    none, possibly no strand -- and no links between rows.  -> dict of arrays (pos, strand, part_off, part_pos, comp_off,
    comp_pos)."""
    rng = np.random.default_rng(seed + 977)
    keep = np.nonzero(rng.random(arr.n) < 0.7)[0]
    pos, strand, part_off, part_pos, comp_off, comp_pos = [], [], [0], [], [0], []
    for i in keep:
        pos.append(int(arr.pos[i]))
        strand.append(int(arr.strand[i]) if rng.random() < 0.85 else 0)
        p = arr.part_pos[int(arr.part_off[i]):int(arr.part_off[i + 1])].tolist()
        c = arr.comp_pos[int(arr.comp_off[i]):int(arr.comp_off[i + 1])].tolist()
        p = p[:int(rng.integers(0, len(p) + 1))]
        c = [x for x in c if rng.random() < 0.7]
        if rng.random() < 0.15 and len(arr.pos):      # a partner / competitor that is no row of the table at all
            (p if rng.random() < 0.5 else c).append(int(arr.pos[int(rng.integers(0, arr.n))]) + int(rng.choice([-1, 1, 3])))
            c = sorted(set(c))
        part_pos.extend(p)
        comp_pos.extend(c)
        part_off.append(len(part_pos))
        comp_off.append(len(comp_pos))
    return dict(pos=np.asarray(pos, np.int64), strand=np.asarray(strand, np.uint8), part_off=np.asarray(part_off, np.uint32),
                part_pos=np.asarray(part_pos, np.int64), comp_off=np.asarray(comp_off, np.uint32), comp_pos=np.asarray(comp_pos, np.int64))
