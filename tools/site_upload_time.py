#!/usr/bin/env python3
"""What a site table costs to put on the device (run on the GPU box): spl_sites_upload of every shard of a workload, best of three.
tools/site_upload_time.py [workload]"""
import os
import sys
import tempfile
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spliser_amd import fast_sites, native, shard, sites, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "human"
cfg = synth.WORKLOADS[name]
cache = "/tmp/wl/%s_s1_seed%d.npz" % (name, cfg["seed"])
wl = synth.Workload.load(cache, name) if os.path.exists(cache) else synth.Workload(name, workers=16)
bed = os.path.join(tempfile.mkdtemp(), "j.bed")
synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
table = fast_sites.build(sites.GeneBins(), False, bed)
items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names) if table.chrom_arrays(c).n]
shards = shard.pack(items, concat_reads=False)
os.environ["SPL_DEBUG_TABLE"] = "1"
with native.Context(0) as ctx:
    for k, sh in enumerate(shards):
        best = None
        for rep in range(3):
            t = time.perf_counter()
            ds = ctx.upload_sites(sh.sites)
            dt = time.perf_counter() - t
            ds.free()
            best = dt if best is None else min(best, dt)
        print("shard %d: %d sites, %d partner edges: spl_sites_upload %.4f s" % (k, sh.sites.n, sh.sites.n_part, best), flush=True)
