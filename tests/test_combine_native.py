"""The native walk of `combine` / `combineShallow` (csrc/spl_combine.cpp) against its Python statement (spliser_amd/combine.py:
_parse_tsv, merge_sites, _QueryTable, write_combined -- the one the reference-generated goldens and the reference's own combine
pin): the same query tables, the same .combined.tsv bytes, on the golden sample files and on random ones that hold what the walk
is sensitive to (regions in file-specific subsets, the same position on both strands, ties between files, sparse evidence),
with arbitrary answers for the gaps.  No GPU: the answers are made up, what is compared is the host's work around them."""
import ctypes
import json
import os
import random

import numpy as np
import pytest

import helpers
from spliser_amd import combine as cmb
from spliser_amd import native

CASE = os.path.join(helpers.GOLDEN, "combine_a")


def _answers(site, idx):
    return (site * 7 + idx * 3) % 11, (site * 5 + idx) % 13


def _python_side(tsvs, titles, out, stranded, q_gene, shallow, cryptic):
    rows = [cmb._parse_tsv(p) for p in tsvs]
    chroms = cmb.region_order(rows)
    if shallow is not None and q_gene != "All":
        rows = [[r for r in fr if r.gene == q_gene] for fr in rows]
    logged = []
    merged = cmb.merge_sites(rows, chroms, len(titles), stranded, q_gene, shallow=shallow, log=logged.append)
    tables = {idx: cmb._QueryTable(qs) for idx, qs in cmb.gap_queries(merged).items()}
    results = {(si, idx): _answers(si, idx) for idx, t in tables.items() for c in t.chrom_index for si in t.site_index[c]}
    cmb.write_combined(out, merged, titles, results, cryptic)
    return chroms, merged, tables, logged


def _native_side(tsvs, titles, out, stranded, q_gene, shallow, cryptic):
    with native.Combine(tsvs) as walk:
        chroms = cmb.region_order_from_runs(walk.region_runs())
        if shallow is not None and q_gene != "All":
            walk.keep_gene(q_gene)
        skipped = walk.merge(chroms, stranded, q_gene, shallow)
        tables = {}
        for idx in range(len(titles)):
            tabs = walk.tables(idx)
            if tabs:
                tables[idx] = cmb._NativeQueryTable(tabs)
        for idx, t in tables.items():
            for c in t.chrom_index:
                site = t.site_index[c]
                a = [_answers(int(si), idx) for si in site]
                walk.answers(idx, site, np.array([x for x, _ in a], np.uint32), np.array([y for _, y in a], np.uint32))
        walk.write(out, titles, cryptic)
        return chroms, walk.n_sites, walk.n_gap_sites, tables, skipped


def _compare(tsvs, tmp_path, stranded, q_gene, shallow, cryptic):
    titles = ["S%d" % k for k in range(len(tsvs))]
    po, no = str(tmp_path / "py.combined.tsv"), str(tmp_path / "nat.combined.tsv")
    chroms, merged, ptab, logged = _python_side(tsvs, titles, po, stranded, q_gene, shallow, cryptic)
    chroms2, n_sites, n_gap, ntab, skipped = _native_side(tsvs, titles, no, stranded, q_gene, shallow, cryptic)
    assert chroms2 == chroms
    assert n_sites == len(merged) and n_gap == sum(1 for m in merged if m.queries)
    assert ["Skipped site {} for insufficient evidence, only {} samples with Site using minimum reads".format(p, s) for p, s in skipped] == logged
    assert sorted(ntab) == sorted(ptab)
    for idx in ptab:
        assert ntab[idx].chrom_index == ptab[idx].chrom_index
        for c in ptab[idx].chrom_index:
            a, b = ptab[idx].chrom_arrays(c), ntab[idx].chrom_arrays(c)
            assert list(ptab[idx].site_index[c]) == ntab[idx].site_index[c].tolist()
            for name in ("pos", "strand", "part_off", "part_pos", "comp_off", "comp_pos", "part_site", "edge_cnt", "alpha"):
                assert np.array_equal(getattr(a, name), getattr(b, name)), (idx, c, name)
            assert a.n == b.n
    assert open(no).read() == open(po).read()
    return len(merged), sum(len(m.queries) for m in merged)


MANIFEST = json.load(open(os.path.join(CASE, "combine_manifest.json")))


@pytest.mark.parametrize("variant", sorted(MANIFEST["variants"]))
def test_native_walk_on_the_golden_samples(variant, tmp_path):
    v = MANIFEST["variants"][variant]
    args = v["combine"]
    shallow = None
    if v.get("command") == "combineShallow":
        def opt(flag, default, cast):
            return cast(args[args.index(flag) + 1]) if flag in args else default
        shallow = (opt("-m", 0, int), opt("-r", 10, int), opt("-e", 0.0, float))
    tsvs = [os.path.join(CASE, "sample%d" % k, "expected.%s.tsv" % variant) for k in range(MANIFEST["n_samples"])]
    q_gene = args[args.index("-g") + 1] if "-g" in args else "All"
    n, _ = _compare(tsvs, tmp_path, "--isStranded" in args, q_gene, shallow, "--beta2Cryptic" in args)
    assert n > 0


def _random_files(seed, tmp_path):
    """2-6 sample files over shared regions: every file lists a random subset of a common pool of sites (position, strand, gene),
    in `process` order (regions in one global order, each file with a subset of them; positions ascending, '+' before '-')."""
    rng = random.Random(seed)
    n_samples = rng.randint(2, 6)
    stranded = rng.random() < 0.5
    cryptic_cols = rng.random() < 0.5
    regions = ["Chr%d" % k for k in range(1, rng.randint(2, 5))] + (["scaffold_9"] if rng.random() < 0.5 else [])
    pool = {}
    for reg in regions:
        positions = sorted(rng.sample(range(50, 4000), rng.randint(1, 60)))
        sites = []
        for p in positions:
            strands = ["+", "-"] if (stranded and rng.random() < 0.2) else [rng.choice(["+", "-", "?"] if not stranded else ["+", "-"])]
            for st in strands:
                sites.append((p, st, rng.choice(["NA", "G1", "G2", "GChr1_2"])))
        pool[reg] = sites
    paths = []
    for k in range(n_samples):
        lines = ["Region\tSite\tStrand\tGene\tSSE\talpha_count\tbeta1_count\tbeta2Simple_count\tbeta2Cryptic_count\tbeta2Cryptic_weighted\tPartners\tCompetitors\n"]
        for reg in regions:
            if rng.random() < 0.15:
                continue                     # (a sample without that region)
            for (p, st, g) in pool[reg]:
                if rng.random() < 0.35:
                    continue
                if not stranded and rng.random() < 0.1:
                    st = rng.choice(["+", "-", "?"])      # (an unstranded site keeps the strand of the sample's own first junction)
                alpha, b1, b2 = rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 9)
                den = alpha + b1 + b2
                sse = alpha / den if den else 0.0
                partners = {q: rng.randint(0, 30) for q in rng.sample(range(50, 4000), rng.randint(0, 4))}
                comps = sorted(rng.sample(range(50, 4000), rng.randint(0, 3)))
                mid = "%d\t%.5f" % (rng.randint(0, 12), rng.choice([0.0, 2.0, 0.33333, 12.5, 1.66667, 0.00012, 7.0 / 3])) if cryptic_cols else "NA\tNA"
                lines.append("%s\t%d\t%s\t%s\t%.3f\t%d\t%d\t%d\t%s\t%s\t%s\n" % (reg, p, st, g, sse, alpha, b1, b2, mid, str(partners), str(comps)))
        path = str(tmp_path / ("s%d.SpliSER.tsv" % k))
        with open(path, "w") as fh:
            fh.writelines(lines)
        paths.append(path)
    return paths, stranded, cryptic_cols, rng


@pytest.mark.parametrize("seed", range(40))
def test_native_walk_on_random_files(seed, tmp_path):
    paths, stranded, cryptic_cols, rng = _random_files(seed, tmp_path)
    total = 0
    for q_gene in ("All", "G1"):
        for shallow in (None, (rng.randint(0, len(paths)), rng.randint(0, 30), rng.choice([0.0, 0.2, 0.5]))):
            for cryptic in (False, True):
                n, _ = _compare(paths, tmp_path, stranded, q_gene, shallow, cryptic)
                total += n
    assert total > 0


def test_files_the_native_parser_leaves_to_python(tmp_path):
    good = "Region\tSite\n" + "Chr1\t100\t+\tNA\t0.500\t4\t2\t2\tNA\tNA\t{200: 4}\t[300]\n"
    odd = {
        "underscore": good.replace("\t100\t", "\t1_00\t"),
        "float_pos": good.replace("\t100\t", "\t100.0\t"),
        "dup_key": good.replace("{200: 4}", "{200: 4, 200: 5}"),
        "tuple": good.replace("[300]", "(300,)"),
        "trailing_comma": good.replace("[300]", "[300,]"),
        "short": "h\nChr1\t100\t+\n",
        "crlf": good.replace("\n", "\r\n"),
        "utf8": good.replace("NA\t0.500", "géne\t0.500"),
        "inf": good.replace("0.500", "inf"),
    }
    ok = str(tmp_path / "ok.tsv")
    with open(ok, "w") as fh:
        fh.write(good)
    with native.Combine([ok]) as walk:
        assert walk.rows(0) == 1
    for name, text in odd.items():
        path = str(tmp_path / (name + ".tsv"))
        with open(path, "w", newline="") as fh:
            fh.write(text)
        with pytest.raises(native.SpliserNativeError) as err:
            native.Combine([ok, path])
        assert err.value.code == -5, name
    with pytest.raises(native.SpliserNativeError) as err:
        native.Combine([str(tmp_path / "missing.tsv")])
    assert err.value.code == -4


def test_str_of_a_float_like_python():
    L = native.lib()
    rng = np.random.default_rng(7)
    vals = [0.0, -0.0, 2.0, 0.33333, 12.5, 1e16, 1e15, 9999999999999998.0, 1.5e-5, 1e-4, 9.999e-5, 1e22, 5e-324, 1.7976931348623157e308,
            0.1 + 0.2, 100.0, 123456.789, float("inf"), float("-inf")]
    vals += (rng.integers(0, 10 ** 7, 3000) / 1e5).tolist()                      # what "{0:.5f}" of process leaves, summed or not
    vals += (rng.integers(0, 10 ** 7, 500) / 1e5 + rng.integers(0, 10 ** 7, 500) / 1e5).tolist()
    vals += np.exp(rng.uniform(-60, 60, 3000)).tolist()
    vals += rng.standard_normal(500).tolist()
    buf = ctypes.create_string_buffer(64)
    for x in vals:
        assert L.spl_fmt_repr(ctypes.c_double(x), buf) == 0
        assert buf.value.decode() == repr(float(x)), x
    assert L.spl_fmt_repr(ctypes.c_double(float("nan")), buf) == 0 and buf.value == b"nan"
