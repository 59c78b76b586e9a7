#!/usr/bin/env python3
"""Where a cold `process` command line spends its time (run on the GPU box, nothing of this process touches the GPU):
    tools/r04_cold_cli.py [--workload human|arabidopsis] [--seq-mode 1|2] [--runs 3] [--env K=V ...]
Writes the sample's files, then runs the CLI as a child `--runs` times: wall clock of the child split into interpreter start ->
main() entered -> main() returned -> process gone, with the library's own timelines (SPL_BAM_TIMING / SPL_PROCESS_TIMING) on stderr."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, time
t0 = time.time()
sys.path.insert(0, %r)
from spliser_amd import cli
t1 = time.time()
rc = cli.main(sys.argv[1:])
t2 = time.time()
sys.stderr.write("[child] started %%.4f imports_done %%.4f main_returned %%.4f\n" %% (t0, t1, t2))
sys.stderr.flush()
sys.stdout.flush()
import os
if os.environ.get("SPL_NO_FAST_EXIT"):
    sys.exit(rc)
os._exit(rc or 0)        # (what python -m spliser_amd does: spliser_amd/__main__.py)
""" % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="human")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seq-mode", type=int, default=1)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--configs", default="", help="several environments on the same files: 'A=1,B=2;C=3;' (an empty one = the default)")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds between two runs (the driver scrubs what the last one freed)")
    a = ap.parse_args()
    import bench
    from spliser_amd import synth
    wl = synth.Workload(a.workload, scale=a.scale, workers=max(1, min(32, os.cpu_count() or 1)))
    files = bench.write_e2e_files(a.workload, wl, None, a.seq_mode)
    prefix = files["prefix"]
    print("files written in %.1f s: %.2f GB" % (files["files_written_s"], os.path.getsize(prefix + ".bam") / 1e9))
    configs = a.configs.split(";") if a.configs else [""]
    for cfg in configs:
        env = dict(os.environ, SPL_BAM_TIMING="1", SPL_PROCESS_TIMING="1")
        for kv in a.env + [x for x in cfg.split(",") if x]:
            k, v = kv.split("=", 1)
            env[k] = v
        print("==== configuration: %s" % (cfg or "(default)"))
        for k in range(a.runs):
            argv = [sys.executable, "-c", CHILD, "process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-A", prefix + ".gff", "-o", prefix + ".cold%d" % k]
            t0 = time.time()
            r = subprocess.run(argv, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
            t1 = time.time()
            stamps = [ln for ln in r.stderr.splitlines() if ln.startswith("[child]")]
            rep = [ln for ln in r.stdout.splitlines() if "Total runtime" in ln]
            print("== run %d: rc %d, wall %.4f s; %s" % (k, r.returncode, t1 - t0, rep[-1].strip() if rep else ""))
            if stamps:
                s = stamps[-1].split()
                st, im, mr = float(s[2]), float(s[4]), float(s[6])
                print("   spawn -> interpreter running %.4f | imports %.4f | main() %.4f | main returned -> process gone %.4f" % (st - t0, im - st, mr - im, t1 - mr))
            for ln in r.stderr.splitlines():
                if ln.startswith("[") and not ln.startswith("[child]"):
                    print("   " + ln)
            if a.pause:
                time.sleep(a.pause)
    import shutil
    shutil.rmtree(files["tmp"], ignore_errors=True)


if __name__ == "__main__":
    main()
