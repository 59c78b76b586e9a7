#!/usr/bin/env python3
"""End-to-end numbers for DESIGN.md: BAM write, decode (threads sweep), `process` CLI wall time.  Run on the GPU box."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SPL_BAM_TIMING", "1")
from spliser_amd import native, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
workload = sys.argv[2] if len(sys.argv) > 2 else "arabidopsis"
stranded = bool(synth.WORKLOADS[workload].get("paired"))
t = time.time()
wl = synth.Workload(workload, scale=scale)
print("generate %.1f s, %d reads" % (time.time() - t, wl.n_reads))
d = tempfile.mkdtemp(prefix="spl_e2e_")
prefix = os.path.join(d, "sample")
t = time.time()
native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=0)
synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=True)
synth.write_gff(prefix + ".gff", wl.genome)
print("write BAM %.1f s, %.1f MB" % (time.time() - t, os.path.getsize(prefix + ".bam") / 1e6))
out = {"reads": wl.n_reads, "bam_mb": os.path.getsize(prefix + ".bam") / 1e6, "decode": {}}
for th in (1, 8, 32, 64, 0):
    t = time.time()
    b = native.BamFile(prefix + ".bam", threads=th)
    dt = time.time() - t
    out["decode"][str(th)] = wl.n_reads / dt
    print("decode threads=%d: %.2f s = %.1f M reads/s" % (th, dt, wl.n_reads / dt / 1e6))
    b.close()
from spliser_amd.process import process  # noqa: E402
for rep in range(2):
    t = time.time()
    timings = process(prefix + ".bam", prefix + ".bed", prefix + "_out", annotationFile=prefix + ".gff", log=lambda m: None,
                      isStranded=stranded, strandedType="fr" if stranded else None, isbeta2Cryptic=stranded)
    dt = time.time() - t
    print("process wall %.2f s = %.1f M reads/s end to end; stages %s" % (dt, wl.n_reads / dt / 1e6, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timings.items()}))
    out["process_s"] = dt
    out["stages"] = timings
from spliser_amd.junctions import junctions  # noqa: E402
t = time.time()
n_j = junctions(prefix + ".bam", prefix + "_junctions.bed", log=lambda m: None, isStranded=stranded, strandedType="fr" if stranded else None)
out["junctions_s"] = time.time() - t
print("junctions CLI wall %.2f s, %d junctions" % (out["junctions_s"], n_j))
b = native.BamFile(prefix + ".bam")
with native.Context(0) as ctx:
    segs = [native.ReadArrays(r.pos, r.flag, r.cig_off, r.cigar) for r in (b.reads(c) for c in b.ref_names) if r is not None and r.n]
    dr = ctx.upload_read_segments([(s, 0) for s in segs[:1]])
    dr.junctions()
    t = time.time()
    for _ in range(5):
        dr.junctions(0, 8, 70, 500000)
    out["junction_table_call_s"] = (time.time() - t) / 5
    print("spl_junctions on %d reads: %.4f s per call (table build + download + sort)" % (dr.n, out["junction_table_call_s"]))
    dr.free()
b.close()
print(json.dumps(out))
