"""How long does the HOST take to issue one step (run on the GPU box)?  tools/host_issue.py [steps]
Prints the issue time per step (loop without the final sync) next to the step time with the sync: when the two are
close, the step is bound by the host's launches, not by the kernels."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import fast_sites, native, shard, sites, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cache = "/tmp/wl/arabidopsis_s1_seed%d.npz" % synth.WORKLOADS["arabidopsis"]["seed"]
wl = synth.Workload.load(cache, "arabidopsis")
bed = os.path.join(tempfile.mkdtemp(), "j.bed")
synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
table = fast_sites.build(sites.GeneBins(), False, bed)
items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names) if table.chrom_arrays(c).n]
shards = shard.pack(items)
ctx = native.Context(0)
dev = [(ctx.upload_sites(sh.sites), ctx.upload_reads(sh.reads)) for sh in shards]
for _ in range(5):
    for ds, dr in dev:
        ctx.count_launch(ds, dr, 0, 0, 0); ctx.sse_launch(ds, False)
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        for ds, dr in dev:
            ctx.count_launch(ds, dr, 0, 0, 0); ctx.sse_launch(ds, False)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("steps %d: issue %.1f us/step, with sync %.1f us/step" % (steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
