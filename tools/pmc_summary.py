#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of each spl_* kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "spl_" not in k:
                continue
            print(k.split("(")[0][:64], {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "n=%d" % len(next(iter(cs.values()))))
