"""Debug aid: which kind of read, repeated how often, breaks parity for a randcase seed."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import randcase
from oracle import oracle
from spliser_amd import native, samio
oracle.build()
seed, stranded = int(sys.argv[1]), int(sys.argv[2])
combine = int(sys.argv[4]) if len(sys.argv) > 4 else 0
arr, rs = randcase.make_case(seed, bool(stranded))
s = native.SiteArrays.from_chrom(arr)


def subset(idx):
    idx = np.asarray(idx, np.int64)
    nops = np.diff(rs.cig_off.astype(np.int64))[idx]
    src = np.concatenate([np.arange(rs.cig_off[i], rs.cig_off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
    off = np.concatenate(([0], np.cumsum(nops)))
    return samio.ReadSet(rs.pos[idx], rs.flag[idx], off, rs.cigar[src.astype(np.int64)])


def ok(reads, ctx):
    r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
    want = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, combine)
    got = ctx.count(s, r, stranded, combine, 0)
    d, _, _ = native.pack_host(r)
    return all(np.array_equal(w, g) for w, g in zip(want, got)), d["n"].sum(axis=0).tolist() if len(d) else []


with native.Context(0) as ctx:
    for rep in (1, 2, 5, 10, 20, 40, 70):
        good, runs = ok(subset(np.repeat(np.arange(rs.n), rep)), ctx)
        print("all reads x%d:" % rep, good, runs)
    for i in range(rs.n):
        good, runs = ok(subset(np.repeat([i], 300)), ctx)
        if not good:
            print("read %d x300 FAILS:" % i, int(rs.pos[i]), int(rs.flag[i]), samio.cigar_string(rs.cigar[int(rs.cig_off[i]):int(rs.cig_off[i + 1])]), runs)
    i = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    for n in (64, 65, 128, 192, 193, 256, 257, 300):
        reads = subset(np.repeat([i], n))
        r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
        want = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, combine)
        got = ctx.count(s, r, stranded, combine, 0)
        print("read %d x%d" % (i, n), [(g.astype(int) - w.astype(int)).tolist() for g, w in zip(got, want)])
