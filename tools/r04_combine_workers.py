import os, sys, time, tempfile, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import argparse
import bench
from spliser_amd import synth, native, process, combine as cmb
args = argparse.Namespace(combine_scale=1.0)
samples = bench.make_combine_samples(args)
tmp = tempfile.mkdtemp(prefix="cmbw_")
noop = lambda m: None
lines = []
for k, wl in enumerate(samples):
    prefix = os.path.join(tmp, "s%d" % k)
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=False)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=0, seq_mode=1)
    process.process(prefix + ".bam", prefix + ".bed", prefix, log=noop)
    process.wait_deferred_close()
    lines.append("S%d\t%s\t%s\n" % (k, prefix + ".SpliSER.tsv", prefix + ".bam"))
sfile = os.path.join(tmp, "samples.tsv")
open(sfile, "w").writelines(lines)
for rep in range(2):
    for w in ("1", "2", "3", "4", "6"):
        os.environ["SPL_COMBINE_WORKERS"] = w
        t = time.perf_counter()
        tm = cmb.combine(sfile, os.path.join(tmp, "all" + w), log=noop)
        wall = time.perf_counter() - t
        process.wait_deferred_close()
        print("workers %s: combine %.3f s (gap fill %.3f)" % (w, wall, tm["gapfill_s"]))
shutil.rmtree(tmp, ignore_errors=True)
