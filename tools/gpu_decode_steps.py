#!/usr/bin/env python3
"""Where the time goes after a device decode (run on the GPU box): decode, then every reference packed and sent up, timed.
tools/gpu_decode_steps.py file.bam [--host]"""
import os
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import native  # noqa: E402

path = sys.argv[1]
host = "--host" in sys.argv
for rep in range(2):
    t0 = time.perf_counter()
    bam = native.BamFile(path, defer=not host, stream=host)
    with native.Context(0) as ctx:
        t1 = time.perf_counter()
        took = None if host else bam.decode_on_device(ctx)
        t2 = time.perf_counter()
        per = []
        for c in bam.ref_names:
            a = time.perf_counter()
            n, _ = bam.wait_ref(c)
            b = time.perf_counter()
            with ctx.begin_reads() as dr:
                if n:
                    dr.add_bam(bam, c)
                dr.finish()
                ctx.sync() if hasattr(ctx, "sync") else None
            per.append((c, n, b - a, time.perf_counter() - b))
        t3 = time.perf_counter()
    t4 = time.perf_counter()
    bam.close()
    print("rep %d: open %.3f, decode %.3f (on device: %s), references %.3f, context closed %.3f, file closed %.3f" % (
        rep, t1 - t0, t2 - t1, took, t3 - t2, t4 - t3, time.perf_counter() - t4))
    print("   slowest references (name, reads, waited, packed + sent): " + ", ".join("%s %d %.3f %.3f" % p for p in sorted(per, key=lambda p: -p[3])[:4]))
