#!/usr/bin/env python3
"""Where a fresh process spends its time before and inside its first `process` call (run on the GPU box):
tools/cli_cold.py file.bam file.bed file.gff"""
import sys
import time
if len(sys.argv) > 1 and sys.argv[1] == "--walls":
    # tools/cli_cold.py --walls file.bam file.bed file.gff [runs]: `python -m spliser_amd process` as a child, with the child's own
    # wall-clock stamps (SPL_CLI_STAMPS=1): what lies before its main() -- the interpreter, the imports -- and behind it, until the
    # parent has it back (the kernel taking the process apart: its mappings, its device memory, its queues)
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bam, bed, gff = sys.argv[2:5]
    for k in range(int(sys.argv[5]) if len(sys.argv) > 5 else 5):
        t_spawn = time.time()
        r = subprocess.run([sys.executable, "-m", "spliser_amd", "process", "-B", bam, "-b", bed, "-A", gff, "-o", "/tmp/cli_cold_walls"] + os.environ.get("SPL_COLD_EXTRA", "").split(), cwd=root,
                           env=dict(os.environ, SPL_CLI_STAMPS="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        t_back = time.time()
        st = {m.group(1): float(m.group(2)) for m in re.finditer(r"\[cli stamp\] ([a-z_ ]+) ([0-9.]+)", r.stdout)}
        rt = re.search(r"Total runtime \(s\): \t([0-9.]+)", r.stdout)
        print("wall %.3f = to __main__ %.3f + imports %.3f + main() %.3f (prints %s) + from main()'s return until the parent has it back %.3f"
              % (t_back - t_spawn, st["__main__"] - t_spawn, st["imported"] - st["__main__"], st["main returned"] - st["imported"], rt.group(1)[:5] if rt else "?",
                 t_back - st["main returned"]) + (" (%.3f of it behind os._exit)" % (t_back - st["leaving"]) if "leaving" in st else ""), flush=True)
    sys.exit(0)
t0 = time.perf_counter()
import os
if os.environ.get("SPL_COLD_IMPORT_TORCH"):     # (which HIP runtime the process ends up with: torch brings its own copy)
    import torch  # noqa: F401
    t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy  # noqa: F401
t1 = time.perf_counter()
from spliser_amd import native, process  # noqa: E402
t2 = time.perf_counter()
native.lib()
t3 = time.perf_counter()
with native.Context(0) as ctx:
    t4 = time.perf_counter()
t5 = time.perf_counter()
print("numpy %.3f, spliser_amd modules %.3f, library loaded %.3f, first context %.3f, closed %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
print("HIP runtime mapped: %s" % sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime64" in l)))
for k in range(int(os.environ.get("SPL_COLD_CALLS", "2"))):
    t = time.perf_counter()
    tm = process.process(sys.argv[1], sys.argv[2], "/tmp/cli_cold_out", annotationFile=sys.argv[3], log=lambda m: None)
    print("process call %d: %.3f s  %s" % (k, time.perf_counter() - t, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in tm.items()}))
