#!/usr/bin/env python3
"""Container-only: how long the REAL reference takes (BASELINE.md plan item 1; `Total runtime (s)`, SpliSER_v0_1_8.py:1360-1361).

The unmodified /root/reference/SpliSER_v0_1_8.py `process` on (1) BASELINE config 1 -- the single-gene case of tests/golden
(`-c Chr1 -g AT1G01060 -m 6000` with its GFF) -- and (2) a 1/1000-scale config 2 (A. thaliana-shaped sample, spliser_amd/synth.py:
20 000 reads), each twice: with a child process per splice site serving `samtools view` (oracle/refharness/samtools: the SAM-text
shim -- a real samtools adds a BAI seek and BGZF inflate per call, this one re-reads a text file), and with the in-process replay
the goldens are made with (no spawn at all: the reference's own Python per read).  INDICATIVE: samtools is a shim here and the
container's cores are not the GPU box's; what it shows is the shape of the cost -- one process spawn per site plus microseconds of
Python per (read, site) pair, on one core -- beside the cost model bench.py prints.

    python tools/time_reference.py > profiles/r03_time_reference.txt
"""
import json
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "refharness"))
import run_reference  # noqa: E402


def timed(sam, bed, out, inprocess, **kw):
    t = time.perf_counter()
    text, log = run_reference.run_process(sam, bed, out, inprocess=inprocess, **kw)
    wall = time.perf_counter() - t
    m = re.search(r"Total runtime \(s\):\s*([0-9.eE+-]+)", log)
    return dict(wall_s=round(wall, 3), total_runtime_s=float(m.group(1)) if m else None, rows=text.count("\n") - 1)


def main():
    if not run_reference.reference_available():
        sys.exit("reference not available at %s (container-only tool)" % run_reference.REFERENCE_DIR)
    from spliser_amd import samio, synth
    out = {"host": {"nproc": os.cpu_count(), "note": "build container; the reference uses one core"}, "cases": []}
    tmp = tempfile.mkdtemp(prefix="spl_timeref_")
    try:
        g = os.path.join(ROOT, "tests", "golden", "single_gene")
        n_reads = sum(1 for line in open(os.path.join(g, "reads.sam")) if not line.startswith("@"))
        case = dict(name="config 1: single gene (tests/golden/single_gene), -c Chr1 -g AT1G01060 -m 6000", reads=n_reads)
        for label, inproc in (("child process per site (samtools shim)", False), ("in-process replay", True)):
            case[label] = timed(os.path.join(g, "reads.sam"), os.path.join(g, "junctions.bed"), os.path.join(tmp, "c1"), inproc,
                                gff=os.path.join(g, "genes.gff"), chrom="Chr1", gene="AT1G01060", max_intron=6000)
        out["cases"].append(case)
        wl = synth.Workload("arabidopsis", scale=0.001, seed=2, workers=1)
        sam, bed = os.path.join(tmp, "a.sam"), os.path.join(tmp, "a.bed")
        samio.write_sam(sam, wl.genome.chrom_names, wl.genome.chrom_lengths, list(zip(wl.genome.chrom_names, wl.reads)))
        synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
        case = dict(name="config 2 at 1/1000: A. thaliana-shaped sample, whole genome", reads=sum(r.n for r in wl.reads))
        for label, inproc in (("child process per site (samtools shim)", False), ("in-process replay", True)):
            case[label] = timed(sam, bed, os.path.join(tmp, "c2"), inproc)
        case["sites"] = case["in-process replay"]["rows"]
        a = case["child process per site (samtools shim)"]
        case["per_site_ms_with_spawn"] = round(1e3 * a["wall_s"] / max(1, case["sites"]), 3)
        out["cases"].append(case)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
