"""spl_tsv_append (the native row formatter `process` writes its output with) against tsv.format_chrom, the Python statement
of outputBedFile's format that the goldens pin: same bytes on every golden case and on adversarial numbers."""
import numpy as np
import pytest

from conftest import golden_cases
import helpers
from spliser_amd import native, samio, sites, tsv


def _native_text(tmp_path, chunks, cryptic):
    path = str(tmp_path / "out.tsv")
    with open(path, "w") as fh:
        fh.write(tsv.HEADER)
    for arr, res in chunks:
        native.tsv_append(path, arr, res, cryptic)
    with open(path) as fh:
        one_by_one = fh.read()
    # ... and all chromosomes in one call (spl_tsv_append_many: what `process` does with a shard's): the same bytes
    with open(path, "w") as fh:
        fh.write(tsv.HEADER)
    native.tsv_append_many(path, list(chunks), cryptic)
    with open(path) as fh:
        assert fh.read() == one_by_one
    return one_by_one


@pytest.mark.parametrize("case,variant,opts", golden_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_native_rows_reproduce_the_golden_files(case, variant, opts, tmp_path, oracle_lib):
    count, sse_fn = helpers.oracle_engine(oracle_lib)
    import os
    case_dir = os.path.join(helpers.GOLDEN, case)
    table = helpers.build_table(case_dir, opts)
    _, reads = samio.read_sam(os.path.join(case_dir, "reads.sam"))
    q_chrom = opts.get("chrom") or "All"
    cryptic = bool(opts.get("cryptic"))
    chunks = []
    for chrom in table.chrom_index:
        if not (q_chrom == chrom or q_chrom == "All"):
            continue
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        rs = reads.get(chrom, samio.ReadSet.empty())
        beta1, b2s_reads, dbl = count(arr, rs, helpers.STRANDED[opts.get("stranded")], 0)
        b2s, b2c, b2w, sse = sse_fn(arr, beta1, b2s_reads, dbl, cryptic)
        chunks.append((arr, dict(beta1=beta1, beta2_simple=b2s, beta2_cryptic=b2c, beta2_weighted=b2w, sse=sse)))
    ref_text, _ = helpers.expected(case, variant)
    assert _native_text(tmp_path, chunks, cryptic) == ref_text


def test_native_rows_on_adversarial_numbers(tmp_path):
    rng = np.random.default_rng(7)
    n = 4000
    arr = sites.ChromArrays()
    arr.chrom, arr.n = "chr_Ünï", n
    arr.pos = np.sort(rng.integers(0, 2 ** 31 - 100, n)).astype(np.int64)
    arr.strand_text = [str(rng.choice(["+", "-", "?", "", "."])) for _ in range(n)]
    arr.genes = [str(rng.choice(["NA", "AT1G01010", "gene with space", "géne", ""])) for _ in range(n)]
    arr.alpha = rng.integers(0, 2 ** 40, n).astype(np.int64)
    deg = rng.integers(0, 4, n)
    arr.part_off = np.zeros(n + 1, np.uint32)
    np.cumsum(deg, out=arr.part_off[1:])
    e = int(arr.part_off[-1])
    arr.part_pos = rng.integers(0, 2 ** 31, e).astype(np.int64)
    arr.edge_cnt = rng.integers(0, 2 ** 33, e).astype(np.int64)
    cdeg = rng.integers(0, 3, n)
    arr.comp_off = np.zeros(n + 1, np.uint32)
    np.cumsum(cdeg, out=arr.comp_off[1:])
    arr.comp_pos = rng.integers(0, 2 ** 31, int(arr.comp_off[-1])).astype(np.int64)
    # doubles on rounding edges: exact ties in binary that are not ties in decimal, halves, tiny, huge, thirds
    specials = np.array([0.0, 1.0, 0.5, 0.0005, 0.0015, 0.0025, 0.1235, 0.9995, 0.99949999999999994, 1e-9, 1 / 3, 2 / 3,
                         0.000005, 0.000015, 1234567.000005, 2 ** 52 + 0.5, 1e15 + 0.3, 5e-324, 0.30000000000000004])
    sse = np.concatenate((specials, rng.random(n - len(specials))))
    b2w = np.concatenate((specials[::-1], rng.random(n - len(specials)) * 10.0 ** rng.integers(-6, 9, n - len(specials))))
    res = dict(beta1=rng.integers(0, 2 ** 32 - 1, n).astype(np.uint32), beta2_simple=rng.integers(0, 2 ** 45, n).astype(np.int64),
               beta2_cryptic=rng.integers(0, 2 ** 45, n).astype(np.int64), beta2_weighted=b2w, sse=sse)
    for cryptic in (False, True):
        want = tsv.HEADER + "".join(tsv.format_chrom(arr, res, cryptic))
        assert _native_text(tmp_path, [(arr, res)], cryptic) == want


def test_fixed_point_text_matches_python_format():
    """spl_fmt_fixed (the writer's "%.3f" / "%.5f") against Python's format over ratios of small integers (what SSE values are --
    exact ties like 1/16 included), random doubles, powers of two, tiny and large values."""
    import ctypes
    import random
    lib = native.lib()
    lib.spl_fmt_fixed.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)

    def got(x, d):
        assert lib.spl_fmt_fixed(x, d, buf) == 0
        return buf.value.decode()

    rnd = random.Random(7)
    values = [0.0, 1.0, 0.5, 0.0625, 0.1875, 0.0005, 0.00049999999999999, 0.9995, 0.99949999999999994, 2.5e-5, 5e-324, 1e-310,
              123456789.123456, 8.9e12, 4503599627370496.0, 1e40, 0.000005, 0.0000049999]
    values += [2.0 ** -k for k in range(0, 80)] + [(2 * k + 1) / 2.0 ** 12 for k in range(0, 2048, 7)]
    values += [a / b for a in range(0, 60) for b in range(1, 60)]
    values += [rnd.random() for _ in range(20000)] + [rnd.random() * 10 ** rnd.randint(-8, 12) for _ in range(20000)]
    values += [rnd.randint(0, 10 ** 6) / 10 ** rnd.randint(1, 7) for _ in range(20000)]   # decimal-looking values next to ties
    for x in values:
        for d in (3, 5, 0, 6):
            assert got(x, d) == ("{0:.%df}" % d).format(x), (x, d)
    assert got(-1.5, 3) == "-1.500" and got(float("inf"), 3) == "inf"


@pytest.mark.parametrize("stranded", [False, True])
def test_rows_of_an_array_built_table(tmp_path, stranded):
    """fast_sites leaves the Gene and Strand columns as arrays (gene index, strand byte): the writer's own way from those to the
    text columns must give what the lists give (tsv.format_chrom asks for the lists)."""
    import os
    from spliser_amd import fast_sites, synth
    wl = synth.Workload("arabidopsis", scale=0.002, seed=4)
    bed, gff = str(tmp_path / "j.bed"), str(tmp_path / "g.gff")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions, stranded=stranded)
    synth.write_gff(gff, wl.genome)
    bins = sites.GeneBins.from_annotation(gff, "gene", "All", log=lambda m: None)
    table = fast_sites.build(bins, stranded, bed)
    assert table is not None
    rng = np.random.default_rng(1)
    chunks, want = [], tsv.HEADER
    for chrom in table.chrom_index:
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        assert arr._genes is None and arr._strand_text is None
        res = dict(beta1=rng.integers(0, 50, arr.n).astype(np.uint32), beta2_simple=rng.integers(0, 50, arr.n), sse=rng.random(arr.n),
                   beta2_cryptic=rng.integers(0, 9, arr.n), beta2_weighted=rng.random(arr.n) * 7)
        chunks.append((arr, res))
    got = _native_text(tmp_path, chunks, True)     # (before anything asks for the lists)
    for arr, res in chunks:
        want += "".join(tsv.format_chrom(arr, res, True))
    assert any(g != "NA" for arr, _ in chunks for g in arr.genes) and got == want
