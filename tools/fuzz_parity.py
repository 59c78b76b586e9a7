#!/usr/bin/env python3
"""Parity fuzzing on the GPU box: tests/randcase.py cases beyond the seeds the test suite pins, each read also repeated 70
times (so that whole waves of every odd shape occur, not only mixed ones), range kernel (+ the aggregating variant now and
then) against the oracle, all strand / combine modes, fused SSE included.   tools/fuzz_parity.py FIRST LAST"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import randcase  # noqa: E402
from oracle import oracle  # noqa: E402
from spliser_amd import native, samio  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
device_path = "--device-path" in sys.argv
odd = "--odd" in sys.argv      # every second seed: BED strands that are none in stranded analyses too, junctions whose ends coincide
import tempfile  # noqa: E402
tmpdir = tempfile.mkdtemp(prefix="spl_fuzz_")
n_device = 0
oracle.build()
t0 = time.time()
n_cases = n_reads = 0
with native.Context(0) as ctx:
    for seed in range(first, last):
        for stranded in (0, 1, 2):
            arr, rs = randcase.make_case(seed, bool(stranded), odd=odd and seed % 2 == 1)
            if arr.n == 0 or rs.n == 0:
                continue
            variants = [rs]
            nops = np.diff(rs.cig_off.astype(np.int64))
            rep = 70
            idx = np.repeat(np.arange(rs.n), rep)
            src = np.concatenate([np.arange(rs.cig_off[i], rs.cig_off[i + 1]) for i in idx]) if rs.n else np.zeros(0, np.int64)
            off = np.concatenate(([0], np.cumsum(nops[idx])))
            variants.append(samio.ReadSet(rs.pos[idx], rs.flag[idx], off, rs.cigar[src.astype(np.int64)]))
            s = native.SiteArrays.from_chrom(arr)
            ds = ctx.upload_sites(s)
            for reads in variants:
                r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
                dr = ctx.upload_reads(r)
                for combine in (0, 1):
                    want = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, reads.pos,
                                            reads.flag, reads.cig_off, reads.cigar, stranded, combine)
                    for flags in ((0, native.OPT_WAVE_AGGREGATION) if seed % 10 == 0 else (0,)):
                        ctx.count_launch(ds, dr, stranded, combine, flags)
                        cryptic = bool(seed & 1)
                        ctx.sse_launch(ds, cryptic)
                        got = ds.counters()
                        for name, w, g in zip(("beta1", "beta2s", "dbl"), want, got):
                            if not np.array_equal(w, g):
                                print("MISMATCH seed %d stranded %d combine %d flags %d reads %d: %s" % (seed, stranded, combine, flags, reads.n, name))
                                sys.exit(1)
                        ws = oracle.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt, *want, cryptic)
                        for name, w, g in zip(("b2s", "b2c", "b2w", "sse"), ws, ds.sse_results()):
                            if not np.array_equal(w, g):
                                print("MISMATCH seed %d stranded %d combine %d flags %d reads %d: %s" % (seed, stranded, combine, flags, reads.n, name))
                                sys.exit(1)
                    n_cases += 1
                    n_reads += reads.n
                dr.free()
            if device_path:
                reads = variants[-1]
                path = os.path.join(tmpdir, "f.bam")
                samio.write_bam(path, ["c1"], [10 ** 8], [("c1", reads)], level=1 + seed % 9)
                os.environ["SPL_INFLATE_WINDOW_BLOCKS"] = "3" if seed % 2 else "114688"
                os.environ["SPL_FORCE_CHUNK"] = "4096" if seed % 3 == 0 else "2048"
                bam = native.BamFile(path, threads=2, defer=True)
                if not bam.decode_on_device(ctx):
                    print("device decoder did not take the file of seed %d" % seed)
                    sys.exit(1)
                for combine in (0, 1):
                    want = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, reads.pos,
                                            reads.flag, reads.cig_off, reads.cigar, stranded, combine)
                    with ctx.begin_reads() as dr2:
                        dr2.add_bam(bam, "c1", 0)
                        dr2.finish()
                        ctx.count_launch(ds, dr2, stranded, combine)
                        got = ds.counters()
                    for name, w, g in zip(("beta1", "beta2s", "dbl"), want, got):
                        if not np.array_equal(w, g):
                            print("MISMATCH (device path) seed %d stranded %d combine %d reads %d: %s" % (seed, stranded, combine, reads.n, name))
                            sys.exit(1)
                    n_device += 1
                bam.close()
                del os.environ["SPL_FORCE_CHUNK"]
            ds.free()
            # the same reads against a query table as `combine` builds them (rows without links, partial lists)
            q = randcase.query_table(arr, seed)
            if len(q["pos"]):
                sq = native.SiteArrays(q["pos"], q["strand"], q["part_off"], q["part_pos"], q["comp_off"], q["comp_pos"],
                                       part_site=np.full(len(q["part_pos"]), -1, np.int32))
                for reads in variants:
                    r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
                    for combine in (0, 1):
                        want = oracle.check_bam(q["pos"], q["strand"], q["part_off"], q["part_pos"], q["comp_off"], q["comp_pos"], reads.pos,
                                                reads.flag, reads.cig_off, reads.cigar, stranded, combine)
                        got = ctx.count(sq, r, stranded, combine, 0)
                        for name, w, g in zip(("beta1", "beta2s", "dbl"), want, got):
                            if not np.array_equal(w, g):
                                print("MISMATCH (query table) seed %d stranded %d combine %d reads %d: %s" % (seed, stranded, combine, reads.n, name))
                                sys.exit(1)
                        n_cases += 1
                        n_reads += reads.n
print("fuzz ok: seeds %d..%d, %d (case, mode) runs, %d of them more through the device decoder and layout, %d reads, %.0f s" % (first, last, n_cases, n_device, n_reads, time.time() - t0))
