#!/usr/bin/env python3
"""BAM decode alone (no GPU): tools/decode_rate.py file.bam [threads ...] -- best of three opens per thread count."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import native  # noqa: E402

path = sys.argv[1]
for th in [int(x) for x in sys.argv[2:]] or [0]:
    best, n = None, 0
    for rep in range(3):
        t = time.perf_counter()
        b = native.BamFile(path, threads=th)
        n = b.n_records
        dt = time.perf_counter() - t
        b.close()
        best = dt if best is None else min(best, dt)
    print("decode %s threads=%d: %.3f s = %.1f M records/s (%.2f GB/s of file)" % (os.path.basename(path), th, best, n / best / 1e6,
                                                                                    os.path.getsize(path) / best / 1e9), flush=True)
