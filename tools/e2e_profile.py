#!/usr/bin/env python3
"""`process` end to end on one workload's files, a few times, with the stage times of each call -- small enough to sit under
rocprofv3 (tools/e2e_timeline.sh).  Runs on the GPU box.

    tools/e2e_profile.py <workload> [--runs N] [--seq-mode 0|1] [--cache DIR] [--files DIR]
"""
import argparse
import json
import os
import sys
import time

import torch  # noqa: F401  (first: one HIP runtime per process, the first one loaded -- see bench.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import native, process, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--seq-mode", type=int, default=0)
ap.add_argument("--cache", default="/tmp/wl")
ap.add_argument("--files", default="/tmp/wl_files")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--devices", default="0", help="e.g. 0,0 = two contexts (and host threads) on GPU 0")
ap.add_argument("--gpu-decode", action="store_true", help="BGZF inflate and record extraction on the GPU (process(gpuDecode=True))")
ap.add_argument("--pause", type=float, default=0.0, help="seconds of sleep before the last run")
ap.add_argument("--auto-decode", action="store_true", help="process(gpuDecode=None): by the file's compression (the product's default)")
args = ap.parse_args()
cfg = synth.WORKLOADS[args.workload]
stranded = "fr" if cfg.get("paired") else None
os.makedirs(args.cache, exist_ok=True)
os.makedirs(args.files, exist_ok=True)
prefix = os.path.join(args.files, "%s_s%g_q%d" % (args.workload, args.scale, args.seq_mode))
n_reads = None
if not (os.path.exists(prefix + ".bam") and os.path.exists(prefix + ".n")):
    cache = os.path.join(args.cache, "%s_s%g_seed%d.npz" % (args.workload, args.scale, cfg["seed"]))
    if os.path.exists(cache):
        wl = synth.Workload.load(cache, args.workload)
    else:
        wl = synth.Workload(args.workload, scale=args.scale, workers=max(1, min(32, os.cpu_count() or 1)))
        wl.save(cache)
    t = time.perf_counter()
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions)
    synth.write_gff(prefix + ".gff", wl.genome)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=6 if args.seq_mode == 2 else 1, threads=0, seq_mode=args.seq_mode)
    n_reads = sum(r.n for r in wl.reads)
    open(prefix + ".n", "w").write(str(n_reads))
    print("files written in %.1f s: %s.bam %.1f MB" % (time.perf_counter() - t, prefix, os.path.getsize(prefix + ".bam") / 1e6), flush=True)
    del wl
n_reads = int(open(prefix + ".n").read())
for k in range(args.runs):
    if args.pause and k + 1 == args.runs and k:
        time.sleep(args.pause)       # (tools/e2e_timeline.py --last-call finds the last call behind this silence)
    t = time.perf_counter()
    tm = process.process(prefix + ".bam", prefix + ".bed", prefix + ".out", annotationFile=prefix + ".gff", isStranded=bool(stranded),
                         strandedType=stranded, isbeta2Cryptic=bool(stranded), log=lambda m: None,
                         devices=tuple(int(d) for d in args.devices.split(",")), gpuDecode=None if args.auto_decode else args.gpu_decode)
    wall = time.perf_counter() - t
    tm["deferred_close_s"] = process.wait_deferred_close()
    print(json.dumps(dict(run=k, workload=args.workload, reads=n_reads, wall_s=round(wall, 4), reads_per_sec=round(n_reads / wall),
                          bam_bytes=os.path.getsize(prefix + ".bam"), gpu_decode="auto" if args.auto_decode else args.gpu_decode, decoder=tm.pop("bam_decode", None), stages={a: round(b, 4) for a, b in tm.items()})), flush=True)
