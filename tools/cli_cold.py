#!/usr/bin/env python3
"""Where a fresh process spends its time before and inside its first `process` call (run on the GPU box):
tools/cli_cold.py file.bam file.bed file.gff"""
import sys
import time
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
