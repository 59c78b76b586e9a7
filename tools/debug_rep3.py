import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import randcase
from spliser_amd import native, samio
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
    ds = ctx.upload_sites(s)
    for rep in (10, 40, 60, 70):
        reads = subset(np.repeat(np.arange(rs.n), rep))
        r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
        dr = ctx.upload_reads(r)
        ctx.count_launch(ds, dr, stranded, combine, 0)
        ctx.sync()
        d, _, _ = native.pack_host(r)
        print("x%d queued %d per copy %.3f chunks %s" % (rep, dr.literal_queue_size(), dr.literal_queue_size() / rep, d["n"].tolist()))
        dr.free()
