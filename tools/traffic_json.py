#!/usr/bin/env python3
"""HBM bytes per launch of the range kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh, as the JSON bench.py
quotes in roofline.traffic (only for the workload and the kernel sources it was measured with).
   tools/traffic_json.py <tag> [the bench.py arguments of the profiled command]"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, args = sys.argv[1], sys.argv[2:]


def is_fused(name):   # void spl_count_ranges_kernel<STRANDED, AGG, BIG, FUSED>(spl_hot_params)
    m = re.search(r"spl_count_ranges_kernel<([^>]*)>", name)
    a = [x.strip() for x in m.group(1).split(",")] if m else []
    return len(a) == 4 and a[3] == "true"


def mean(counter, sub, kernel="spl_count_ranges_kernel", fused=None):
    vals = collections.defaultdict(list)
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"] and (fused is None or is_fused(r["Kernel_Name"]) == fused):
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        return None, 0
    # the mean over ALL launches of the kernel, whichever of its instantiations a shard's read set takes (a set of 64 M reads and
    # more is cut into chunks of twice the size: another instantiation) -- bench.py's algorithmic bytes per launch are the mean
    # over the shards' launches as well
    every = [v for name in vals for v in vals[name]]
    return sum(every) / len(every), len(every)


# a fused pass (the range kernel reads the BAM-native arrays itself): its launches are the step's; the launches of the two-kernel
# comparison leg of the same bench.py run (layout + range, records in memory) are reported beside them
fetch, n = mean("FETCH_SIZE", "fetch", fused=True)
FUSED = fetch is not None
if not FUSED:
    fetch, n = mean("FETCH_SIZE", "fetch")
write, _ = mean("WRITE_SIZE", "write", fused=True if FUSED else None)
h = hashlib.sha256(open(os.path.join(ROOT, "spliser_amd", "libspliser_hip.so"), "rb").read()).hexdigest()[:16]
import re


def strip_source(text):   # (= bench.py strip_source: comments and white space do not count)
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


kh = hashlib.sha256()
for name in ("spl_kernels.hip", "spl_device.h", "spl_pack.h", "spl_classify.h", "spl_pack.cpp", "spl_devpack.hip", "spl_devpack.h", "spl_layout_tile.h"):   # (= bench.py KERNEL_SOURCES)
    kh.update(strip_source(open(os.path.join(ROOT, "spliser_amd", "csrc", name), "rb").read().decode("utf-8", "replace")).encode("utf-8"))
workload = args[args.index("--workload") + 1] if "--workload" in args else "human"
out = {"workload": workload, "bench_args": args, "lib_sha16": h, "kernel_src_sha16": kh.hexdigest()[:16], "kernel": "spl_count_ranges_kernel", "dispatches": n,
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_pmc.sh), means per dispatch",
       "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
       "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read (MI355X_MICROARCH.md, HBM): x 2 for the "
                     "record stream; this kernel's mix of 8 / 16 / 24-byte record loads and 8-byte gathers was calibrated at 1.927 on a "
                     "stream of known size (profiles/r01e_traffic.json), which is what is applied; WRITE_SIZE is exact",
       "hbm_bytes_per_launch": None if fetch is None else int(fetch * 1024 * (2.0 if FUSED else 1.927) + (write or 0) * 1024)}
if FUSED:
    out["kernel"] = "spl_count_ranges_kernel<FUSED>"
    out["correction"] = ("FETCH_SIZE x 2 (the fused pass reads the BAM-native arrays with 16-byte coalesced loads, as the layout kernel does: "
                         "MI355X_MICROARCH.md, HBM; its 8 / 12-byte index gathers are a few per cent of the bytes), WRITE_SIZE as it is")
    rf, rn = mean("FETCH_SIZE", "fetch", fused=False)
    rw, _ = mean("WRITE_SIZE", "write", fused=False)
    out["two_kernels_range"] = {"kernel": "spl_count_ranges_kernel (records in memory: the comparison leg)", "dispatches": rn, "FETCH_SIZE_KB": rf, "WRITE_SIZE_KB": rw,
                                "hbm_bytes_per_launch": None if rf is None else int(rf * 1024 * 1.927 + (rw or 0) * 1024)}
# the layout kernel of the same step: 16-byte coalesced loads throughout (the guide's x 2 for FETCH_SIZE), 8 / 16-byte stores
lf, ln = mean("FETCH_SIZE", "fetch", "spl_layout_kernel")
lw, _ = mean("WRITE_SIZE", "write", "spl_layout_kernel")
out["layout"] = {"kernel": "spl_layout_kernel", "dispatches": ln, "FETCH_SIZE_KB": lf, "WRITE_SIZE_KB": lw,
                 "correction": "FETCH_SIZE x 2 (wide coalesced loads: MI355X_MICROARCH.md, HBM), WRITE_SIZE as it is",
                 "hbm_bytes_per_launch": None if lf is None else int(lf * 1024 * 2 + (lw or 0) * 1024)}
print(json.dumps(out, indent=1))
