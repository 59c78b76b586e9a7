#!/usr/bin/env python3
"""Warm `process` calls on one file, the file image as a ring (default) against all of it on the device (SPL_IMAGE_RING_PIECES=100000),
interleaved in ONE process on one box:  tools/r04_ring_ab.py [--workload human] [--seq-mode 1] [--rounds 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="human")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seq-mode", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--configs", default=";SPL_IMAGE_RING_PIECES=100000")
    a = ap.parse_args()
    import bench
    from spliser_amd import process, synth
    wl = synth.Workload(a.workload, scale=a.scale, workers=max(1, min(32, os.cpu_count() or 1)))
    files = bench.write_e2e_files(a.workload, wl, None, a.seq_mode)
    prefix = files["prefix"]
    print("files written in %.1f s: %.2f GB" % (files["files_written_s"], os.path.getsize(prefix + ".bam") / 1e9))
    configs = a.configs.split(";")
    walls = {c: [] for c in configs}
    for r in range(a.rounds + 1):
        for cfg in configs:
            saved = {}
            for kv in [x for x in cfg.split(",") if x]:
                k, v = kv.split("=", 1)
                saved[k] = os.environ.get(k)
                os.environ[k] = v
            t = time.perf_counter()
            tm = process.process(prefix + ".bam", prefix + ".bed", prefix + ".out", annotationFile=prefix + ".gff", log=lambda m: None)
            wall = time.perf_counter() - t
            process.wait_deferred_close()
            for k, v in saved.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
            if r:       # (round 0 warms both up)
                walls[cfg].append(wall)
            print("round %d  %-40s %.4f s  gpu_s %.4f" % (r, cfg or "(default: ring)", wall, tm["gpu_s"]))
    for cfg in configs:
        w = sorted(walls[cfg])
        print("%-40s best %.4f  median %.4f  all %s" % (cfg or "(default: ring)", w[0], w[len(w) // 2], " ".join("%.3f" % x for x in walls[cfg])))
    import shutil
    shutil.rmtree(files["tmp"], ignore_errors=True)


if __name__ == "__main__":
    main()
