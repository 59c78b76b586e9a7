#!/usr/bin/env python3
"""Container-only: the oracle (oracle/spliser_oracle.c + the product's host Steps 0-2) against the REAL reference on the
adversarial shapes of tests/randcase.py, far beyond the committed goldens.

For every seed x {unstranded, fr, rf} x {plain, --beta2Cryptic}: the unmodified /root/reference/SpliSER_v0_1_8.py `process` is
run through oracle/refharness (in-process replay of its per-site `samtools view`, HTSeq stub: the two third-party boundaries
that stay unpinned), its .SpliSER.tsv and a full-precision dump of every Site are compared with what the oracle gives for the
same files: TSV byte for byte, counters equal, doubles bit-identical.

    python tools/fuzz_reference.py FIRST LAST [--jobs N]      -> a summary line; exit 1 on the first mismatch

``--combine``: 2-4 random samples per seed (tests/golden/make_golden.py's generator: one gene model, sample-specific junctions and
reads) through the real reference's `process`, then its `combine` and `combineShallow` (random -m / -r / -e), for {unstranded, fr,
rf} -- against the product's host walk over the same per-sample files (spliser_amd.combine: region order, lock-step merge with
its order dependence, SpliSER_v0_1_8.py:869-904, query snapshots, writers) with the oracle answering the gap-fill queries:
.combined.tsv byte for byte.

Needs /root/reference: never runs on the GPU box; nothing of the reference is copied anywhere (inputs are synthesised, outputs
compared and dropped).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refharness"))


ODD = False      # --odd-strands: BED strands '?' / '.' in stranded analyses too, junctions whose two ends coincide


def one_seed(seed):
    import numpy as np
    import randcase
    import run_reference
    from oracle import oracle
    from spliser_amd import tsv
    runs = reads = 0
    for stranded in (None, "fr", "rf"):
        tmp = tempfile.mkdtemp(prefix="spl_fuzzref_")
        try:
            arr, rs = randcase.make_case(seed, bool(stranded), dirpath=tmp, odd=ODD)
            if arr.n == 0:
                continue
            scode = {None: 0, "fr": 1, "rf": 2}[stranded]
            cnt = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, rs.pos, rs.flag,
                                   rs.cig_off, rs.cigar, scode, 0)
            for cryptic in (False, True):
                dump = os.path.join(tmp, "dump.json")
                text, _ = run_reference.run_process(os.path.join(tmp, "reads.sam"), os.path.join(tmp, "junctions.bed"),
                                                    os.path.join(tmp, "ref"), inprocess=True, dump_json=dump, stranded=stranded,
                                                    cryptic=cryptic)
                b2s, b2c, b2w, sse = oracle.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                                                      cnt[0], cnt[1], cnt[2], cryptic)
                mine = tsv.HEADER + "".join(tsv.format_chrom(arr, dict(beta1=cnt[0], beta2_simple=b2s, beta2_cryptic=b2c,
                                                                       beta2_weighted=b2w, sse=sse), cryptic))
                tag = "seed %d stranded %s cryptic %s" % (seed, stranded, cryptic)
                if mine != text:
                    return "MISMATCH (tsv) " + tag
                with open(dump) as fh:
                    ref_rows = json.load(fh)
                if len(ref_rows) != arr.n:
                    return "MISMATCH (rows) " + tag
                for i, ref in enumerate(ref_rows):
                    got = (int(arr.pos[i]), arr.strand_text[i], int(arr.alpha[i]), int(cnt[0][i]), int(b2s[i]), int(b2c[i]), float(b2w[i]),
                           float(sse[i]))
                    want = (ref["pos"], ref["strand"], ref["alpha"], ref["beta1"], ref["beta2Simple"], ref["beta2Cryptic"],
                            ref["beta2Weighted"], ref["sse"])
                    if got != want:
                        return "MISMATCH (site %d: %r != %r) %s" % (i, got, want, tag)
                    # PartnerBeta2DoubleCounts as the reference leaves it: what checkBam added (:527, :551) plus, per partner, the
                    # counts of that partner's junctions flanking the site, which findBeta2Counts adds to the same dict (:594-599)
                    # (a dict by POSITION: two partner sites at one position -- two edges -- share an entry: checkBam's count once, the
                    #  flanking junctions of both, each of a partner's junctions once however often its lists name the position)
                    dbl, from_reads = {}, {}
                    t = int(arr.pos[i])
                    for e in range(int(arr.part_off[i]), int(arr.part_off[i + 1])):
                        ps, key = int(arr.part_site[e]), int(arr.part_pos[e])
                        if key not in from_reads:
                            from_reads[key] = int(cnt[2][e])
                            if cnt[2][e]:
                                dbl[key] = int(cnt[2][e])
                        if ps >= 0:
                            pp = int(arr.pos[ps])
                            seen = set()
                            for f in range(int(arr.part_off[ps]), int(arr.part_off[ps + 1])):
                                cp = int(arr.part_pos[f])
                                if cp in seen:
                                    continue
                                seen.add(cp)
                                if (pp > t and cp < t) or (pp < t and cp > t):
                                    dbl[key] = dbl.get(key, 0) + int(arr.edge_cnt[f])
                    if dbl != {int(k): int(v) for k, v in ref["double"]}:
                        return "MISMATCH (double counts of site %d: %r != %r) %s" % (i, dbl, ref["double"], tag)
                runs += 1
                reads += rs.n
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return (runs, reads)


def one_combine_seed(seed):
    import random
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    import run_reference
    from oracle import oracle
    from spliser_amd import combine as cmb
    from spliser_amd import process as proc
    rng = random.Random(seed * 31 + 5)
    n_samples = rng.randint(2, 4)
    samples, gff = make_golden.combine_case(seed, n_samples)
    runs = 0
    tmp = tempfile.mkdtemp(prefix="spl_fuzzcmb_")
    try:
        for k, (reads, juncs) in enumerate(samples):
            make_golden.write_case(os.path.join(tmp, "sample%d" % k), reads, juncs, gff=gff)
        for stype in (None, "fr", "rf"):
            cryptic = bool(stype) and rng.random() < 0.5
            lines = []
            for k in range(n_samples):
                sd = os.path.join(tmp, "sample%d" % k)
                run_reference.run_process(os.path.join(sd, "reads.sam"), os.path.join(sd, "junctions.bed"), os.path.join(tmp, "s%d_%s" % (k, stype)),
                                          inprocess=True, stranded=stype, cryptic=cryptic)
                lines.append("S%d\t%s\t%s\n" % (k, os.path.join(tmp, "s%d_%s.SpliSER.tsv" % (k, stype)), os.path.join(sd, "reads.sam")))
            sfile = os.path.join(tmp, "samples_%s.tsv" % stype)
            with open(sfile, "w") as fh:
                fh.writelines(lines)
            base = (["--isStranded", "-s", stype] if stype else []) + (["--beta2Cryptic"] if cryptic else [])
            m, r, e = rng.randint(0, n_samples), rng.randint(0, 6), rng.choice([0.0, 0.1, 0.3])
            for command, extra, shallow in (("combine", [], None), ("combineShallow", ["-m", str(m), "-r", str(r), "-e", str(e)], (m, r, e))):
                out_ref = os.path.join(tmp, "ref_%s_%s" % (stype, command))
                rc, log = run_reference.run_cli([command, "-S", sfile, "-o", out_ref] + base + extra, inprocess=True)
                if rc != 0:
                    return "reference %s failed, seed %d: %s" % (command, seed, log[-400:])
                titles, tsvs, bams = cmb.read_samples_file(sfile)
                rows = [cmb._parse_tsv(p) for p in tsvs]
                chroms = cmb.region_order(rows)
                merged = cmb.merge_sites(rows, chroms, len(titles), bool(stype), "All", shallow=shallow)
                scode = {None: 0, "fr": 1, "rf": 2}[stype]
                results = {}
                for idx, queries in cmb.gap_queries(merged).items():
                    source = proc.open_alignments(bams[idx])
                    table = cmb._QueryTable(queries)
                    for chrom in table.chrom_index:
                        st, rd = table.chrom_arrays(chrom), source.reads(chrom)
                        if rd is None or rd.n == 0:
                            b1 = b2 = [0] * st.n
                        else:
                            b1, b2, _ = oracle.check_bam(st.pos, st.strand, st.part_off, st.part_pos, st.comp_off, st.comp_pos, rd.pos, rd.flag,
                                                         rd.cig_off, rd.cigar, scode, 1)
                        for j, si in enumerate(table.site_index[chrom]):
                            results[(si, idx)] = (int(b1[j]), int(b2[j]))
                mine = os.path.join(tmp, "mine_%s_%s.combined.tsv" % (stype, command))
                cmb.write_combined(mine, merged, titles, results, cryptic)
                if open(mine).read() != open(out_ref + ".combined.tsv").read():
                    return "MISMATCH (%s) seed %d stranded %s cryptic %s shallow %r, %d samples" % (command, seed, stype, cryptic, shallow, n_samples)
                # ... and the native walk (csrc/spl_combine.cpp) with the same answers: the same bytes once more
                from spliser_amd import native
                import numpy as np
                with native.Combine(tsvs) as walk:
                    walk.merge(cmb.region_order_from_runs(walk.region_runs()), bool(stype), "All", shallow)
                    for idx in range(len(titles)):
                        for chrom, tb in walk.tables(idx):
                            a = [results[(int(si), idx)] for si in tb["site"]]
                            walk.answers(idx, tb["site"], np.array([x for x, _ in a], np.uint32), np.array([y for _, y in a], np.uint32))
                    walk.write(mine + ".native", titles, cryptic)
                if open(mine + ".native").read() != open(out_ref + ".combined.tsv").read():
                    return "MISMATCH of the native walk (%s) seed %d stranded %s cryptic %s shallow %r, %d samples" % (command, seed, stype, cryptic, shallow, n_samples)
                runs += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (runs, n_samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--combine", action="store_true", help="the real reference's combine / combineShallow against the product's host walk")
    ap.add_argument("--odd-strands", action="store_true", help="process: BED strands other than + / - in stranded analyses as well, and junctions whose "
                    "ends coincide -- positions that hold several sites for one query (binary_site_search's landing, SpliSER_v0_1_8.py:175-225)")
    a = ap.parse_args()
    global ODD
    ODD = a.odd_strands           # (the pool forks after this: the workers see it)
    import run_reference
    if not run_reference.reference_available():
        sys.exit("reference not available at %s (container-only tool)" % run_reference.REFERENCE_DIR)
    from oracle import oracle
    oracle.build()
    t0 = time.time()
    import multiprocessing
    runs = reads = 0
    if a.combine:
        samples = 0
        with multiprocessing.get_context("fork").Pool(a.jobs) as pool:
            for res in pool.imap_unordered(one_combine_seed, range(a.first, a.last), chunksize=2):
                if isinstance(res, str):
                    print(res)
                    pool.terminate()
                    sys.exit(1)
                runs += res[0]
                samples += res[1]
        print("reference combine fuzz ok: seeds %d..%d x {unstranded, fr, rf} x {combine, combineShallow with random -m -r -e}: %d runs of "
              "SpliSER_v0_1_8.py combine / combineShallow over %d samples (2-4 per seed, each through the reference's process first), "
              ".combined.tsv byte-identical to spliser_amd.combine's host walk -- the Python statement and the native one (csrc/spl_combine.cpp) -- with the oracle's gap fill, %.0f s" % (a.first, a.last, runs, samples, time.time() - t0))
        return
    with multiprocessing.get_context("fork").Pool(a.jobs) as pool:
        for res in pool.imap_unordered(one_seed, range(a.first, a.last), chunksize=4):
            if isinstance(res, str):
                print(res)
                pool.terminate()
                sys.exit(1)
            runs += res[0]
            reads += res[1]
    print("reference fuzz ok%s: seeds %d..%d x {unstranded, fr, rf} x {plain, --beta2Cryptic}: %d runs of SpliSER_v0_1_8.py process, "
          "%d reads, TSV byte-identical, counters and doubles of every site identical, %.0f s"
          % (" (--odd-strands: BED strands '?' / '.' in stranded analyses too, junctions whose ends coincide)" if ODD else "", a.first, a.last, runs, reads, time.time() - t0))


if __name__ == "__main__":
    main()
