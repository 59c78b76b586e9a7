import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import randcase
from oracle import oracle
from spliser_amd import native, samio
oracle.build()
seed, stranded, combine = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
arr, rs = randcase.make_case(seed, bool(stranded))
s = native.SiteArrays.from_chrom(arr)
def subset(idx):
    idx = np.asarray(idx, np.int64)
    nops = np.diff(rs.cig_off.astype(np.int64))[idx]
    src = np.concatenate([np.arange(rs.cig_off[i], rs.cig_off[i + 1]) for i in idx])
    off = np.concatenate(([0], np.cumsum(nops)))
    return samio.ReadSet(rs.pos[idx], rs.flag[idx], off, rs.cigar[src.astype(np.int64)])
with native.Context(0) as ctx:
    for rep in (45, 50, 53, 54, 60, 70):
        reads = subset(np.repeat(np.arange(rs.n), rep))
        r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
        want = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, combine)
        for name, flags in (("ranges", 0), ("pairs", 1), ("agg", 2)):
            for trial in range(2):
                got = ctx.count(s, r, stranded, combine, flags)
                d = [(g.astype(np.int64) - w.astype(np.int64)) for g, w in zip(got, want)]
                print("x%d %s trial %d reads %d:" % (rep, name, trial, reads.n), [(int(i), int(v)) for i, v in enumerate(d[0]) if v], [(int(i), int(v)) for i, v in enumerate(d[1]) if v], [(int(i), int(v)) for i, v in enumerate(d[2]) if v])
