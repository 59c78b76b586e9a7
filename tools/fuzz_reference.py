#!/usr/bin/env python3
"""Container-only: the oracle (oracle/spliser_oracle.c + the product's host Steps 0-2) against the REAL reference on the
adversarial shapes of tests/randcase.py, far beyond the committed goldens.

For every seed x {unstranded, fr, rf} x {plain, --beta2Cryptic}: the unmodified /root/reference/SpliSER_v0_1_8.py `process` is
run through oracle/refharness (in-process replay of its per-site `samtools view`, HTSeq stub: the two third-party boundaries
that stay unpinned), its .SpliSER.tsv and a full-precision dump of every Site are compared with what the oracle gives for the
same files: TSV byte for byte, counters equal, doubles bit-identical.

    python tools/fuzz_reference.py FIRST LAST [--jobs N]      -> a summary line; exit 1 on the first mismatch

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
            arr, rs = randcase.make_case(seed, bool(stranded), dirpath=tmp)
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
                    dbl = {}
                    t = int(arr.pos[i])
                    for e in range(int(arr.part_off[i]), int(arr.part_off[i + 1])):
                        ps = int(arr.part_site[e])
                        total, have = int(cnt[2][e]), bool(cnt[2][e])
                        if ps >= 0:
                            pp = int(arr.pos[ps])
                            for f in range(int(arr.part_off[ps]), int(arr.part_off[ps + 1])):
                                cp = int(arr.part_pos[f])
                                if (pp > t and cp < t) or (pp < t and cp > t):
                                    total += int(arr.edge_cnt[f])
                                    have = True
                        if have:
                            dbl[int(arr.part_pos[e])] = total
                    if dbl != {int(k): int(v) for k, v in ref["double"]}:
                        return "MISMATCH (double counts of site %d: %r != %r) %s" % (i, dbl, ref["double"], tag)
                runs += 1
                reads += rs.n
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return (runs, reads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    import run_reference
    if not run_reference.reference_available():
        sys.exit("reference not available at %s (container-only tool)" % run_reference.REFERENCE_DIR)
    from oracle import oracle
    oracle.build()
    t0 = time.time()
    import multiprocessing
    runs = reads = 0
    with multiprocessing.get_context("fork").Pool(a.jobs) as pool:
        for res in pool.imap_unordered(one_seed, range(a.first, a.last), chunksize=4):
            if isinstance(res, str):
                print(res)
                pool.terminate()
                sys.exit(1)
            runs += res[0]
            reads += res[1]
    print("reference fuzz ok: seeds %d..%d x {unstranded, fr, rf} x {plain, --beta2Cryptic}: %d runs of SpliSER_v0_1_8.py process, "
          "%d reads, TSV byte-identical, counters and doubles of every site identical, %.0f s" % (a.first, a.last, runs, reads, time.time() - t0))


if __name__ == "__main__":
    main()
