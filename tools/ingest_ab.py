#!/usr/bin/env python3
"""A/B of the device decode's kernels inside whole `process` calls (GPU box): one file per (workload, scale, seq-mode), then for
every configuration -- a set of environment variables -- a fresh child process that calls process() a few times with the
library's own stopwatch on (spl_prof_enable) and prints the walls and the kernels' table.  Configurations run interleaved,
`--rounds` times, so that a box's drift hits them alike.

    tools/ingest_ab.py --scale 0.25 --seq-mode 2 --configs "wave:;lanes:SPL_CRC_LANES=1" --runs 4 --rounds 2
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="human")
ap.add_argument("--scale", type=float, default=0.25)
ap.add_argument("--seq-mode", type=int, default=2)
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--configs", default="default:")
ap.add_argument("--devices", default="0")
ap.add_argument("--files", default="/tmp/wl_files")
ap.add_argument("--child", default=None)
ap.add_argument("--show-stderr", action="store_true", help="print what the children wrote to stderr (SPL_BAM_TIMING=1's stamps)")
args = ap.parse_args()

prefix = os.path.join(args.files, "%s_s%g_q%d" % (args.workload, args.scale, args.seq_mode))

if args.child is None:
    from spliser_amd import native, synth      # (host side only: the parent never touches the GPU)
    os.makedirs(args.files, exist_ok=True)
    if not (os.path.exists(prefix + ".bam") and os.path.exists(prefix + ".n")):
        t = time.perf_counter()
        wl = synth.Workload(args.workload, scale=args.scale, workers=max(1, min(32, os.cpu_count() or 1)))
        synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions)
        synth.write_gff(prefix + ".gff", wl.genome)
        native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=6 if args.seq_mode == 2 else 1, threads=0, seq_mode=args.seq_mode)
        open(prefix + ".n", "w").write(str(sum(r.n for r in wl.reads)))
        print("files written in %.1f s: %s.bam %.1f MB" % (time.perf_counter() - t, prefix, os.path.getsize(prefix + ".bam") / 1e6), flush=True)
        del wl
    configs = []
    for item in args.configs.split(";"):
        name, _, envs = item.partition(":")
        configs.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    results = {name: [] for name, _ in configs}
    for rnd in range(args.rounds):
        for name, env in configs:
            e = dict(os.environ)
            e.update(env)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--workload", args.workload, "--scale", str(args.scale),
                                  "--seq-mode", str(args.seq_mode), "--runs", str(args.runs), "--devices", args.devices, "--files", args.files],
                                 env=e, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if out.returncode or not line:
                print("config %s failed (rc %d): %s" % (name, out.returncode, (out.stderr or out.stdout)[-800:]), flush=True)
                continue
            results[name].append(json.loads(line[-1]))
            r = results[name][-1]
            if args.show_stderr and out.stderr:
                print(out.stderr, flush=True)
            print("round %d %-12s first %.3f walls %s  kernels(ms/call): %s" % (rnd, name, r["first"], " ".join("%.3f" % w for w in r["walls"]),
                                                                     ", ".join("%s %.1f" % (k["kernel"].replace("spl_", "").replace("_kernel", ""), k["ms"] / r["calls"]) for k in r["kernels"][:8])), flush=True)
    print(json.dumps(results))
else:
    import torch  # noqa: F401
    from spliser_amd import native, process
    n_reads = int(open(prefix + ".n").read())
    devices = tuple(int(d) for d in args.devices.split(","))
    walls = []
    native.Context(devices[0]).close()      # (the GPU context and the code object: not the first call's to pay for, as in bench.py)
    t = time.perf_counter()
    process.process(prefix + ".bam", prefix + ".bed", prefix + ".out", annotationFile=prefix + ".gff", log=lambda m: None, devices=devices)   # (the first call: no device memory at hand yet)
    first = time.perf_counter() - t
    process.wait_deferred_close()
    native.prof_enable(True)
    for k in range(args.runs):
        t = time.perf_counter()
        process.process(prefix + ".bam", prefix + ".bed", prefix + ".out", annotationFile=prefix + ".gff", log=lambda m: None, devices=devices)
        walls.append(time.perf_counter() - t)
        process.wait_deferred_close()
    rep = native.prof_report()
    native.prof_enable(False)
    print(json.dumps(dict(config=args.child, reads=n_reads, bam_bytes=os.path.getsize(prefix + ".bam"), calls=args.runs, first=round(first, 4), walls=[round(w, 4) for w in walls], kernels=rep)))
