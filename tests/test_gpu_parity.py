"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference goldens."""
import os

import numpy as np
import pytest

from conftest import golden_cases
import helpers
from spliser_amd import native, samio, shard, sites, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


KERNELS = {"ranges": 0, "pairs": native.OPT_PAIR_KERNEL, "ranges_agg": native.OPT_WAVE_AGGREGATION}


def gpu_engine(ctx, flags=0):
    def count(arr, reads, stranded, combine_mode):
        s = native.SiteArrays.from_chrom(arr)
        r = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
        return ctx.count(s, r, stranded, combine_mode, flags)

    def sse(arr, beta1, b2s_reads, dbl, cryptic):
        return ctx.sse(native.SiteArrays.from_chrom(arr), beta1, b2s_reads, dbl, cryptic)
    return count, sse


@pytest.mark.parametrize("kernel", sorted(KERNELS))
@pytest.mark.parametrize("case,variant,opts", golden_cases(), ids=lambda v: v if isinstance(v, str) else "")
def test_gpu_reproduces_reference_goldens(case, variant, opts, kernel, ctx):
    text, rows = helpers.run_case(case, opts, gpu_engine(ctx, KERNELS[kernel]))
    ref_text, ref_rows = helpers.expected(case, variant)
    assert text == ref_text
    helpers.assert_rows_match(rows, ref_rows, bool(opts.get("cryptic")))


@pytest.mark.parametrize("case", ["cigar_corners", "random_b", "kat1", "random_unstranded_q"])
@pytest.mark.parametrize("stranded", [0, 1, 2])
@pytest.mark.parametrize("combine", [0, 1])
@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_gpu_counters_match_oracle_all_modes(case, stranded, combine, kernel, ctx, oracle_lib):
    table = helpers.build_table(os.path.join(helpers.GOLDEN, case), {"stranded": "fr" if stranded else None})
    _, reads = samio.read_sam(os.path.join(helpers.GOLDEN, case, "reads.sam"))
    ocount, _ = helpers.oracle_engine(oracle_lib)
    gcount, _ = gpu_engine(ctx, KERNELS[kernel])
    for chrom in table.chrom_index:
        arr = table.chrom_arrays(chrom)
        rs = reads.get(chrom, samio.ReadSet.empty())
        for w, g in zip(ocount(arr, rs, stranded, combine), gcount(arr, rs, stranded, combine)):
            assert np.array_equal(w, g)


def _table_for(wl, tmp_path, stranded):
    bed = str(tmp_path / "j.bed")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
    table = sites.SiteTable(is_stranded=stranded)
    table.add_bed(bed)
    table.find_competitors()
    return table


@pytest.mark.parametrize("kernel", sorted(KERNELS))
@pytest.mark.parametrize("stranded,cryptic", [(0, False), (1, True), (2, True)])
def test_gpu_matches_oracle_on_synthetic_genome(stranded, cryptic, kernel, ctx, oracle_lib, tmp_path):
    """~400k reads over 5 chromosomes, packed into one shard (one launch) vs per-chromosome oracle."""
    wl = synth.Workload("arabidopsis", scale=0.02, seed=21 + stranded)
    table = _table_for(wl, tmp_path, bool(stranded))
    names = wl.genome.chrom_names
    items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(names) if table.chrom_arrays(c).n]
    shards = shard.pack(items)
    assert len(shards) == 1
    sh = shards[0]
    ds, dr = ctx.upload_sites(sh.sites), ctx.upload_reads(sh.reads)
    ctx.count_launch(ds, dr, stranded, 0, KERNELS[kernel])
    ctx.sse_launch(ds, cryptic)
    beta1, b2s_reads, dbl = ds.counters()
    b2s, b2c, b2w, sse = ds.sse_results()
    dr.free()
    ds.free()
    total = 0
    for (chrom, arr, reads), (r0, r1), (e0, e1) in zip(items, sh.site_rows, sh.edge_rows):
        w1, w2, w3 = oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                          reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, 0)
        assert np.array_equal(beta1[r0:r1], w1) and np.array_equal(b2s_reads[r0:r1], w2) and np.array_equal(dbl[e0:e1], w3)
        ws = oracle_lib.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt, w1, w2, w3, cryptic)
        for g, w in zip((b2s[r0:r1], b2c[r0:r1], b2w[r0:r1], sse[r0:r1]), ws):
            assert np.array_equal(g, w)
        assert np.all(np.abs(sse[r0:r1] - ws[3]) <= 1e-9)
        total += int(w1.sum()) + int(w2.sum())
    assert total > 10000


def test_gpu_reads_with_non_consuming_ops(ctx, oracle_lib):
    """Soft clips, hard clips, insertions and padding change nothing for checkBam; the pack kernel drops them from short
    CIGARs (so that "5S95M100N50M" takes the once-spliced path).  Random decorations of spliced and unspliced reads on a
    table with rivals, against the oracle, all kernels, all strand modes."""
    rng = np.random.default_rng(17)
    wl = synth.Workload("arabidopsis", scale=0.004, seed=31)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        import pathlib
        table = _table_for(wl, pathlib.Path(tmp), True)
    name = wl.genome.chrom_names[0]
    arr, reads = table.chrom_arrays(name), wl.reads[0]
    n = reads.n
    cig_off = reads.cig_off.astype(np.int64)
    new_ops, new_off = [], [0]
    for i in range(n):
        ops = reads.cigar[cig_off[i]:cig_off[i + 1]].tolist()
        style = int(rng.integers(0, 8))
        out = []
        if style in (1, 3, 5):
            out.append((int(rng.integers(1, 9)) << 4) | 5)          # leading hard clip
        if style in (1, 2, 3):
            out.append((int(rng.integers(1, 30)) << 4) | 4)         # leading soft clip
        for k, op in enumerate(ops):
            if style in (4, 5) and k == 0 and (op & 15) == 0 and (op >> 4) > 20:   # an insertion splits the first block
                a = int(rng.integers(5, (op >> 4) - 5))
                out += [(a << 4) | 0, (int(rng.integers(1, 4)) << 4) | 1, (((op >> 4) - a) << 4) | 0]
            elif style == 6 and k == 0:
                out += [op, (3 << 4) | 6]                            # padding after the first op
            else:
                out.append(op)
        if style in (2, 3, 7):
            out.append((int(rng.integers(1, 30)) << 4) | 4)         # trailing soft clip
        new_ops += out
        new_off.append(len(new_ops))
    deco = native.ReadArrays(reads.pos, reads.flag, np.array(new_off, np.uint32), np.array(new_ops, np.uint32))
    sites_c = native.SiteArrays.from_chrom(arr)
    for stranded in (0, 1, 2):
        want = oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, deco.pos, deco.flag,
                                    deco.cig_off, deco.cigar, stranded, 0)
        for flags in KERNELS.values():
            got = ctx.count(sites_c, deco, stranded, 0, flags)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
    assert int(want[0].sum()) > 1000 and int(want[1].sum()) > 100


def test_gpu_twice_spliced_class_limits(ctx, oracle_lib):
    """The twice-spliced class packs five lengths into three words (aligned < 4096, introns < 2^28): reads on and just beyond
    those limits, with =/X blocks, soft clips, a deletion instead of an intron, in every neighbourhood (alone, in waves of
    other classes), all modes, against the oracle."""
    rng = np.random.default_rng(23)
    # sites: junction ends of the reads below plus alternatives sharing ends (rivals), both strands
    base = [1000, 1100, 1400, 1500, 5000, 5100, 5400, 5600, 9000, 9050, 300000000, 300000100]
    SHIFT = 10000   # (room in front of the first site for the 4095-base blocks)
    pos = np.array(sorted(set(base + [1050, 1450, 5050, 5500, 9020, 1000 + 4095, 1000 + 4096 + 50])), np.int64) + SHIFT
    n = len(pos)
    strand = np.where(np.arange(n) % 3 == 0, ord("-"), ord("+")).astype(np.uint8)
    # partners: a ring of mutual links over neighbours two apart gives everybody competitors
    part = [[] for _ in range(n)]
    for i in range(n):
        for j in (i + 1, i + 2):
            if j < n:
                part[i].append(j)
                part[j].append(i)
    part_off = np.zeros(n + 1, np.uint32)
    np.cumsum([len(x) for x in part], out=part_off[1:])
    part_site = np.array([j for x in part for j in x], np.int32)
    part_pos = pos[part_site]
    comp = [sorted({int(pos[c]) for p_ in x for c in part[p_] if c != i}) for i, x in enumerate(part)]
    comp_off = np.zeros(n + 1, np.uint32)
    np.cumsum([len(x) for x in comp], out=comp_off[1:])
    comp_pos = np.array([c for x in comp for c in x], np.int64)
    sites_c = native.SiteArrays(pos, strand, part_off, part_pos, comp_off, comp_pos, part_site=part_site)

    def rec(flag, p, *ops):
        return (flag, p, ops)
    M, I, D, N, S, EQ, X = 0, 1, 2, 3, 4, 7, 8
    shapes = [
        rec(0, 951, (50, M), (100, N), (300, M), (100, N), (60, M)),                   # 1000|1100 .. 1400|1500 ends on sites
        rec(16, 951, (50, EQ), (100, N), (300, X), (100, N), (60, EQ)),
        rec(99, 951, (3, S), (50, M), (100, N), (300, M), (100, N), (60, M), (7, S)),   # soft clips: same class after compaction
        rec(0, 951, (50, M), (100, N), (300, M), (100, D), (60, M)),                    # a deletion where the second intron was
        rec(0, 951, (50, M), (100, N), (150, M), (2, I), (150, M), (100, N), (60, M)),  # insertion splits the middle block: wide
        rec(0, 1000 - 4094, (4095, M), (100, N), (300, M), (100, N), (60, M)),          # longest block that fits 12 bits
        rec(0, 1000 - 4095, (4096, M), (100, N), (300, M), (100, N), (60, M)),          # one more: wide
        rec(0, 4951, (50, M), (100, N), (300, M), (200, N), (4095, M)),
        rec(0, 8951, (50, M), (50, N), (30, M), ((1 << 28) - 1, N), (40, M)),         # the longest intron a BAM record can hold
        rec(147, 8951, (50, M), (50, N), (0, M), (5, N), (40, M)),                    # an empty middle block
        rec(0, 951, (50, M), (100, N), (300, M)), rec(0, 951, (150, M)), rec(4, 1000, (50, M), (100, N), (300, M), (100, N), (60, M)),
    ]
    recs = []
    for k in range(3000):                        # every shape in every neighbourhood
        recs.append(shapes[int(rng.integers(0, len(shapes)))])
    recs += [shapes[0]] * 300 + [shapes[7]] * 200  # and whole waves of the class
    recs.sort(key=lambda r: r[1])
    rpos = np.array([r[1] for r in recs], np.int64) + SHIFT
    rflag = np.array([r[0] for r in recs], np.uint16)
    off = np.concatenate(([0], np.cumsum([len(r[2]) for r in recs])))
    cig = np.array([(ln << 4) | code for r in recs for ln, code in r[2]], np.uint32)
    reads_c = native.ReadArrays(rpos, rflag, off, cig)
    seen = np.zeros(3, np.int64)
    for stranded in (0, 1, 2):
        for combine in (0, 1):
            want = oracle_lib.check_bam(pos, strand, part_off, part_pos, comp_off, comp_pos, reads_c.pos, reads_c.flag, reads_c.cig_off,
                                        reads_c.cigar, stranded, combine)
            for name, flags in KERNELS.items():
                got = ctx.count(sites_c, reads_c, stranded, combine, flags)
                for g, w in zip(got, want):
                    assert np.array_equal(g, w), (stranded, combine, name)
            seen += [int(w.sum()) for w in want]
    assert seen[0] > 1000 and seen[1] > 1000 and seen[2] > 0   # beta1, beta2Simple and double counts all exercised


def test_gpu_back_to_back_passes_on_one_table(ctx, tmp_path, oracle_lib):
    """The counter region of a device table exists twice and is cleared on the side by the previous pass (or by a clearing
    launch after a pair-kernel pass): a sequence of passes with changing modes and kernels on ONE uploaded table and read set
    must give the oracle's counters every time, and the device error word must not leak from one pass to the next."""
    wl = synth.Workload("arabidopsis", scale=0.003, seed=41)
    table = _table_for(wl, tmp_path, True)
    name = wl.genome.chrom_names[1]
    arr, reads = table.chrom_arrays(name), wl.reads[1]
    sites_c = native.SiteArrays.from_chrom(arr)
    reads_c = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
    ds, dr = ctx.upload_sites(sites_c), ctx.upload_reads(reads_c)
    want = {}
    for stranded in (0, 1, 2):
        for combine in (0, 1):
            want[(stranded, combine)] = oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                                             reads.pos, reads.flag, reads.cig_off, reads.cigar, stranded, combine)
    sequence = [(0, 0, "ranges"), (1, 0, "ranges"), (1, 0, "pairs"), (0, 0, "ranges"), (2, 1, "ranges"), (2, 1, "ranges_agg"),
                (0, 1, "pairs"), (0, 1, "pairs"), (1, 1, "ranges"), (0, 0, "ranges_agg"), (0, 0, "ranges")]
    for stranded, combine, kernel in sequence:
        ctx.count_launch(ds, dr, stranded, combine, KERNELS[kernel])
        for g, w in zip(ds.counters(), want[(stranded, combine)]):
            assert np.array_equal(g, w), (stranded, combine, kernel)
    # a pass that trips the range check, then a clean one on the same table
    far = ctx.upload_reads(native.ReadArrays([2147483000], [0], [0, 1], [(1 << 20) << 4]))
    ctx.count_launch(ds, far, 0, 0)
    with pytest.raises(native.SpliserNativeError):
        ds.counters()
    far.free()
    ctx.count_launch(ds, dr, 0, 0)
    for g, w in zip(ds.counters(), want[(0, 0)]):
        assert np.array_equal(g, w)
    dr.free()
    ds.free()


@pytest.mark.parametrize("tail_stream", ["1", "0", "host-wait"])
def test_gpu_tail_stream_pipelined_passes(tmp_path, oracle_lib, monkeypatch, tail_stream):
    """The literal kernel and the scan of a pass run on a stream of their own while the next pass's range kernel is under way
    (three counter copies per table, two queue buffers per read set; SPL_TAIL_STREAM=0: everything on one stream).  Passes
    launched back to back without a download in between, on two tables and read sets taking turns, with changing modes and
    kernels: whatever is downloaded, whenever, must be the oracle's result for the last pass on that table."""
    monkeypatch.setenv("SPL_TAIL_STREAM", "0" if tail_stream == "0" else "1")
    monkeypatch.setenv("SPL_TAIL_HOST_WAIT", "1" if tail_stream == "host-wait" else "0")  # (who waits for the tail two passes back)
    wl = synth.Workload("arabidopsis", scale=0.004, seed=43)
    table = _table_for(wl, tmp_path, True)
    with native.Context(0) as piped:
        dev, want = [], []
        for k in (0, 1):
            name = wl.genome.chrom_names[k]
            arr, reads = table.chrom_arrays(name), wl.reads[k]
            dev.append((piped.upload_sites(native.SiteArrays.from_chrom(arr)),
                        piped.upload_reads(native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar))))
            want.append({(st, cb): (oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                                         reads.pos, reads.flag, reads.cig_off, reads.cigar, st, cb), arr)
                         for st in (0, 1, 2) for cb in (0, 1)})
        def check(k, st, cb, cryptic):
            ds = dev[k][0]
            cnt, arr = want[k][(st, cb)]
            for g, w in zip(ds.counters(), cnt):
                assert np.array_equal(g, w), (k, st, cb)
            piped.sse_launch(ds, cryptic)
            w_sse = oracle_lib.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                                         cnt[0], cnt[1], cnt[2], cryptic)
            for g, w in zip(ds.sse_results(), w_sse):
                assert np.array_equal(np.asarray(g), np.asarray(w)), (k, st, cb, "sse")
        # bursts of passes, nothing read back until the end of a burst
        for burst, (st, cb) in enumerate([(0, 0), (1, 0), (2, 1), (0, 1)]):
            for rep in range(7):
                for k in (0, 1):
                    piped.count_launch(dev[k][0], dev[k][1], st, cb, 0)
            check(1, st, cb, burst % 2 == 1)
            check(0, st, cb, burst % 2 == 0)
        # the other kernels in between (they run on the main stream alone), the tables and read sets crossed over
        piped.count_launch(dev[0][0], dev[0][1], 1, 0, KERNELS["pairs"])
        piped.count_launch(dev[1][0], dev[1][1], 1, 0, 0)
        piped.count_launch(dev[0][0], dev[0][1], 2, 0, KERNELS["ranges_agg"])
        piped.count_launch(dev[1][0], dev[1][1], 2, 0, 0)
        check(0, 2, 0, False)
        check(1, 2, 0, True)
        for rep in range(5):
            piped.count_launch(dev[0][0], dev[0][1], 0, 0, 0)
        check(0, 0, 0, False)
        assert dev[0][1].literal_queue_size() >= 0
        # one table, two read sets taking turns (what `combine` does with a sample's gap-fill queries): the counter copies
        # cycle with the table, the queue buffers with the read set
        arr0, r1 = table.chrom_arrays(wl.genome.chrom_names[0]), wl.reads[1]
        crossed = oracle_lib.check_bam(arr0.pos, arr0.strand, arr0.part_off, arr0.part_pos, arr0.comp_off, arr0.comp_pos,
                                       r1.pos, r1.flag, r1.cig_off, r1.cigar, 1, 0)
        for rep in range(5):
            piped.count_launch(dev[0][0], dev[1][1], 1, 0, 0)
            piped.count_launch(dev[0][0], dev[0][1], 1, 0, 0)
        check(0, 1, 0, True)
        for rep in range(4):
            piped.count_launch(dev[0][0], dev[0][1], 1, 0, 0)
            piped.count_launch(dev[0][0], dev[1][1], 1, 0, 0)
        for g, w in zip(dev[0][0].counters(), crossed):
            assert np.array_equal(g, w), "one table, the other read set"
        for ds, dr in dev:
            dr.free()
            ds.free()


def test_gpu_segment_upload_equals_packed_upload(ctx, tmp_path):
    """spl_reads_upload_segments (per-chromosome arrays shifted on the device) against the host-packed shard."""
    wl = synth.Workload("arabidopsis", scale=0.01, seed=77)
    table = _table_for(wl, tmp_path, False)
    names = wl.genome.chrom_names
    items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(names) if table.chrom_arrays(c).n]
    items.insert(2, (names[0] + "_empty", table.chrom_arrays(names[0]), None))   # a chromosome without reads
    sh, = shard.pack(items)
    assert len(sh.read_segments) == len(items) - 1
    ds = ctx.upload_sites(sh.sites)
    got = []
    for dr in (ctx.upload_reads(sh.reads), ctx.upload_read_segments(sh.read_segments)):
        assert dr.n == sh.reads.n
        ctx.count_launch(ds, dr, 0, 0)
        ctx.sse_launch(ds, True)
        got.append(ds.counters() + ds.sse_results())
        dr.free()
    ds.free()
    for a, b in zip(*got):
        assert np.array_equal(a, b)
    assert int(got[0][0].sum()) > 1000
    empty = ctx.upload_read_segments([])
    assert empty.n == 0
    empty.free()


def test_gpu_long_introns_and_hot_sites(ctx, oracle_lib):
    """Skew: reads whose introns span thousands of sites (wave-cooperative path), one site hit by 200k reads
    (LDS counter contention), sites outside the LDS window (global-atomic path), unsorted reads."""
    rng = np.random.default_rng(5)
    n_sites = 30000
    pos = np.sort(rng.choice(np.arange(1000, 3_000_000), n_sites, replace=False)).astype(np.int64)
    strand = np.where(rng.random(n_sites) < 0.5, ord("+"), ord("-")).astype(np.uint8)
    # partners: pair consecutive sites; competitors: a few
    part_off = np.arange(n_sites + 1, dtype=np.uint32)
    partner = np.arange(n_sites) ^ 1
    part_pos = pos[partner]
    comp_off = np.zeros(n_sites + 1, np.uint32)
    has_comp = rng.random(n_sites) < 0.2
    comp_off[1:] = np.cumsum(has_comp)
    comp_pos = pos[(np.arange(n_sites)[has_comp] + 2) % n_sites]
    recs = []
    for _ in range(300):   # long introns: 10 kb .. 2.5 Mb
        p = int(rng.integers(1000, 400000))
        recs.append((int(rng.choice([0, 16, 99, 147])), p, "20M%dN30M" % int(rng.integers(10000, 2_500_000))))
    hot = int(pos[1234])
    recs += [(0, hot - 40, "100M")] * 3000
    hot_reads = samio.ReadSet.from_records(recs)
    # bulk: 200k unspliced reads on the hot site + random 150M reads; then shuffle a slice to break sortedness
    bulk_pos = np.concatenate((np.full(200000, hot - 70), rng.integers(1000, 2_999_000, 300000))).astype(np.int64)
    bulk = samio.ReadSet(bulk_pos, rng.choice([0, 16], bulk_pos.shape[0]), np.arange(bulk_pos.shape[0] + 1),
                         np.full(bulk_pos.shape[0], 150 << 4, np.uint32))
    allpos = np.concatenate((hot_reads.pos, bulk.pos)).astype(np.int64)
    allflag = np.concatenate((hot_reads.flag, bulk.flag))
    nops = np.concatenate((np.diff(hot_reads.cig_off.astype(np.int64)), np.ones(bulk.n, np.int64)))
    ops = np.concatenate((hot_reads.cigar, bulk.cigar))
    order = np.argsort(allpos, kind="stable")
    order[1000:5000] = order[1000:5000][::-1]
    src = np.concatenate(([0], np.cumsum(nops)))
    cig = np.concatenate([ops[src[i]:src[i + 1]] for i in order])
    off = np.concatenate(([0], np.cumsum(nops[order])))
    sites_c = native.SiteArrays(pos, strand, part_off, part_pos, comp_off, comp_pos, part_site=partner)
    reads_c = native.ReadArrays(allpos[order], allflag[order], off, cig)
    for stranded, flags in ((0, 0), (1, 0), (0, native.OPT_PAIR_KERNEL), (2, native.OPT_PAIR_KERNEL)):
        got = ctx.count(sites_c, reads_c, stranded, 0, flags)
        want = oracle_lib.check_bam(pos, strand, part_off, part_pos, comp_off, comp_pos, reads_c.pos, reads_c.flag,
                                    reads_c.cig_off, reads_c.cigar, stranded, 0)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    assert int(want[0][1234]) >= 50000
    # a table without part_site (or with one-way partner links) must silently take the pair kernel
    one_way = native.SiteArrays(pos, strand, part_off, part_pos, comp_off, comp_pos)
    got = ctx.count(one_way, reads_c, 1, 0)
    want = oracle_lib.check_bam(pos, strand, part_off, part_pos, comp_off, comp_pos, reads_c.pos, reads_c.flag,
                                reads_c.cig_off, reads_c.cigar, 1, 0)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_gpu_read_with_more_ops_than_the_packed_count_holds(ctx, oracle_lib):
    """A read of 70 000 CIGAR ops (the packed op count saturates at 65 535) between ordinary reads: the kernels must find
    its true op count although the chunk-local partition moves it away from its place in the input."""
    # the long read ends at 105 099: a walk that borrowed the ops of the next reads would run on over the site at 105 150
    pos = np.array([150, 400, 100300, 100900, 105150, 200500], np.int64)
    strand = np.full(6, ord("+"), np.uint8)
    part_off = np.array([0, 1, 2, 3, 4, 4, 4], np.uint32)
    part_pos = np.array([400, 150, 100900, 100300], np.int64)
    part_site = np.array([1, 0, 3, 2], np.int32)
    comp_off = np.zeros(7, np.uint32)
    comp_pos = np.zeros(0, np.int64)
    sites_c = native.SiteArrays(pos, strand, part_off, part_pos, comp_off, comp_pos, part_site=part_site)
    n_pairs = 35000
    long_ops = np.empty(2 * n_pairs, np.uint32)       # 1M 2D 1M 2D ...: 70 000 ops, 105 000 reference bases
    long_ops[0::2] = (1 << 4) | 0
    long_ops[1::2] = (2 << 4) | 2
    recs_ops = [long_ops, np.array([(100 << 4) | 0], np.uint32), np.array([(50 << 4) | 0, (249 << 4) | 3, (60 << 4) | 0], np.uint32),
                np.array([(120 << 4) | 0], np.uint32)]
    rpos = np.array([100, 120, 101, 100250], np.int64)
    order = np.argsort(rpos, kind="stable")
    off = np.concatenate(([0], np.cumsum([len(recs_ops[i]) for i in order])))
    reads_c = native.ReadArrays(rpos[order], np.zeros(4, np.uint16), off, np.concatenate([recs_ops[i] for i in order]))
    want = oracle_lib.check_bam(pos, strand, part_off, part_pos, comp_off, comp_pos, reads_c.pos, reads_c.flag,
                                reads_c.cig_off, reads_c.cigar, 0, 0)
    for flags in KERNELS.values():
        got = ctx.count(sites_c, reads_c, 0, 0, flags)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    assert int(want[0].sum()) > 0


def test_gpu_empty_and_degenerate_inputs(ctx):
    empty_sites = native.SiteArrays(np.zeros(0), np.zeros(0, np.uint8), np.zeros(1), np.zeros(0), np.zeros(1), np.zeros(0))
    one_read = native.ReadArrays([100], [0], [0, 1], [50 << 4])
    b1, b2, d = ctx.count(empty_sites, one_read)
    assert len(b1) == 0 and len(b2) == 0 and len(d) == 0
    sites_c = native.SiteArrays([120], [ord("+")], [0, 1], [300], [0, 0], np.zeros(0))
    no_reads = native.ReadArrays(np.zeros(0), np.zeros(0), [0], np.zeros(0))
    b1, b2, d = ctx.count(sites_c, no_reads)
    assert b1.tolist() == [0] and b2.tolist() == [0]
    b1, _, _ = ctx.count(sites_c, one_read)
    assert b1.tolist() == [1]
    star = native.ReadArrays([120, 120], [0, 4], [0, 0, 1], [60 << 4])   # CIGAR '*' and unmapped-with-CIGAR at t
    b1, _, _ = ctx.count(sites_c, star)
    assert b1.tolist() == [1]


def test_gpu_rejects_bad_arguments(ctx):
    sites_c = native.SiteArrays([200, 100], [43, 43], [0, 0, 0], np.zeros(0), [0, 0, 0], np.zeros(0))
    reads_c = native.ReadArrays([100], [0], [0, 1], [50 << 4])
    with pytest.raises(native.SpliserNativeError):
        ctx.count(sites_c, reads_c)          # unsorted site table
    ok_sites = native.SiteArrays([100], [43], [0, 0], np.zeros(0), [0, 0], np.zeros(0))
    with pytest.raises(native.SpliserNativeError):
        ctx.count(ok_sites, reads_c, stranded=3)
    far = native.ReadArrays([2147483000], [0], [0, 1], [(1 << 20) << 4])
    with pytest.raises(native.SpliserNativeError) as err:
        ctx.count(ok_sites, far)
    assert err.value.code == -6


def test_process_cli_end_to_end_bam(ctx, tmp_path, oracle_lib):
    """BAM file -> native decoder -> packed shard -> kernels -> .SpliSER.tsv, vs the committed golden."""
    from spliser_amd import cli
    for case, variant in (("random_b", "fr_cryptic"), ("multichrom", "annot"), ("single_gene", "gene")):
        d = os.path.join(helpers.GOLDEN, case)
        opts = dict(__import__("json").load(open(os.path.join(helpers.GOLDEN, "manifest.json")))[case][variant])
        names, sets = samio.read_sam(os.path.join(d, "reads.sam"))
        bam = str(tmp_path / (case + ".bam"))
        samio.write_bam(bam, names, [10 ** 8] * len(names), [(c, sets[c]) for c in names if c in sets], with_seq=True)
        argv = ["process", "-B", bam, "-b", os.path.join(d, "junctions.bed"), "-o", str(tmp_path / case)]
        if opts.get("gff"):
            argv += ["-A", os.path.join(d, "genes.gff")]
        if opts.get("chrom"):
            argv += ["-c", opts["chrom"]]
        if opts.get("gene"):
            argv += ["-g", opts["gene"], "-m", str(opts["max_intron"])]
        if opts.get("stranded"):
            argv += ["--isStranded", "-s", opts["stranded"]]
        if opts.get("cryptic"):
            argv += ["--beta2Cryptic"]
        assert cli.main(argv) == 0
        got = open(str(tmp_path / case) + ".SpliSER.tsv").read()
        assert got == helpers.expected(case, variant)[0]


@pytest.mark.parametrize("seed", range(0, 24))
@pytest.mark.parametrize("stranded", [0, 1, 2])
def test_gpu_query_tables_like_combine(seed, stranded, ctx, oracle_lib):
    """Tables as `combine` asks about them: rows without links between them, partial partner / competitor lists, partners
    that are no rows, rows without strand.  The range kernel's junction table is built from each row's own lists, so these
    take the same path as `process` tables; both modes, against the oracle (and the pair kernel)."""
    import randcase
    arr, rs = randcase.make_case(seed + 300, bool(stranded))
    q = randcase.query_table(arr, seed)
    if not len(q["pos"]) or rs.n == 0:
        pytest.skip("empty case")
    sq = native.SiteArrays(q["pos"], q["strand"], q["part_off"], q["part_pos"], q["comp_off"], q["comp_pos"],
                           part_site=np.full(len(q["part_pos"]), -1, np.int32))
    r = native.ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar)
    for combine in (0, 1):
        want = oracle_lib.check_bam(q["pos"], q["strand"], q["part_off"], q["part_pos"], q["comp_off"], q["comp_pos"], rs.pos, rs.flag,
                                    rs.cig_off, rs.cigar, stranded, combine)
        for flags in KERNELS.values():
            for w, g in zip(want, ctx.count(sq, r, stranded, combine, flags)):
                assert np.array_equal(w, g)
