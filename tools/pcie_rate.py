"""The hot path with the hand-over included (run on the GPU box): caller's host arrays -> spl_reads_upload (host packing into the
page-locked staging ring, H2D copies on the copy stream) -> one counting pass + beta2 / SSE -> results back on the host.
bench.py's `value` starts with the data resident in HBM; this is the rate a caller sees who hands over host buffers every time.
tools/pcie_rate.py [repeats]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import fast_sites, native, shard, sites, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cache = "/tmp/wl/arabidopsis_s1_seed%d.npz" % synth.WORKLOADS["arabidopsis"]["seed"]
wl = synth.Workload.load(cache, "arabidopsis")
bed = os.path.join(tempfile.mkdtemp(), "j.bed")
synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
table = fast_sites.build(sites.GeneBins(), False, bed)
items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names) if table.chrom_arrays(c).n]
shards = shard.pack(items)
n_reads = sum(sh.reads.n for sh in shards)
host_bytes = sum(sh.reads.pos.nbytes + sh.reads.flag.nbytes + sh.reads.cig_off.nbytes + sh.reads.cigar.nbytes for sh in shards)
ctx = native.Context(0)
dsites = [ctx.upload_sites(sh.sites) for sh in shards]
best = None
for rep in range(reps):
    t0 = time.perf_counter()
    for sh, ds in zip(shards, dsites):
        dr = ctx.upload_reads(sh.reads)
        t1 = time.perf_counter()
        ctx.count_launch(ds, dr, 0, 0, 0)
        ctx.sse_launch(ds, False)
        cnt = ds.counters()
        res = ds.sse_results()
        dr.free()
    t2 = time.perf_counter()
    print("pass %d: pack + upload %.4f s, count + SSE + download %.4f s, total %.4f s = %.1f M reads/s (%.1f GB/s of host arrays)"
          % (rep, t1 - t0, t2 - t1, t2 - t0, n_reads / (t2 - t0) / 1e6, host_bytes / (t2 - t0) / 1e9))
    best = min(best or 1e9, t2 - t0)
print("best: %.4f s = %.1f M reads/s for %d reads, %.1f MB of host arrays" % (best, n_reads / best / 1e6, n_reads, host_bytes / 1e6))
