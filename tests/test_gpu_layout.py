"""BAM-native arrays resident on the device -> counters, both ways the library has: the layout kernel
(spliser_amd/csrc/spl_devpack.hip: the chunked, class-partitioned records of spl_pack.h in memory, one launch per read set, every
read classified once) followed by the range kernel, and the FUSED range kernel (spl_kernels.hip: a pass over a read
set whose segments lie in one set of arrays makes a tile's records in LDS and counts them there; SPL_FUSED=0 turns it off) -- held
to the oracle (SpliSER_v0_1_8.py:408-559 is what all restate), on reads made to cross every boundary the kernels have: threads of
four reads, waves, tiles, chunks (cells of the grid over the arrays' indexes), segments that begin anywhere, CIGAR stretches longer
than the workgroup's stage in LDS."""
import numpy as np
import pytest

import helpers
import randcase
from spliser_amd import native, samio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    native.build()
    with native.Context(0) as c:
        yield c


def _repeat(rs, times, order=None):
    """Every read `times` times (runs of each class cross threads, waves and chunks), optionally re-ordered."""
    keep = np.repeat(np.arange(rs.n), times)
    if order is not None:
        keep = keep[order(len(keep))]
    n_ops = np.diff(rs.cig_off.astype(np.int64))
    op_idx = np.concatenate([np.arange(rs.cig_off[i], rs.cig_off[i + 1]) for i in keep]) if len(keep) else np.zeros(0, np.int64)
    return samio.ReadSet(rs.pos[keep], rs.flag[keep], np.concatenate(([0], np.cumsum(n_ops[keep]))).astype(np.uint32),
                         rs.cigar[op_idx.astype(np.int64)])


def _arrays(rs):
    return native.ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar)


def _count_soa(ctx, arr, segments, stranded, combine=0, expected=0):
    """segments: [(ReadSet, shift)] -> counters after upload_soa + layout + one counting pass."""
    with ctx.upload_soa([_arrays(rs) for rs, _ in segments]) as soa:
        with ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
            dr = ctx.begin_reads(expected)
            try:
                for k, (_, shift) in enumerate(segments):
                    dr.add_soa(soa, k, shift)
                dr.finish()
                ctx.count_launch(ds, dr, stranded, combine)
                first = ds.counters()
                dr.relayout()                      # the same records once more, then the same counters
                ctx.count_launch(ds, dr, stranded, combine)
                again = ds.counters()
            finally:
                dr.free()
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    return first


@pytest.mark.parametrize("seed", range(0, 12))
@pytest.mark.parametrize("stranded", [0, 1, 2])
@pytest.mark.parametrize("chunk", ["2048", "4096"])
@pytest.mark.parametrize("fused", ["1", "0"])
def test_layout_kernel_counts_like_the_oracle(seed, stranded, chunk, fused, ctx, oracle_lib, monkeypatch):
    """fused = 1 (the default): the fused kernel's pass (a stranded one with its smaller windows).  fused = 0: layout + range."""
    arr, rs = randcase.make_case(seed + 900, bool(stranded))
    if arr.n == 0 or rs.n == 0:
        pytest.skip("empty case")
    monkeypatch.setenv("SPL_FORCE_CHUNK", chunk)
    monkeypatch.setenv("SPL_FUSED", fused)
    big = _repeat(rs, 67)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, stranded, 0)
    got = _count_soa(ctx, arr, [(big, 0)], stranded)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)


@pytest.mark.parametrize("seed", range(0, 8))
@pytest.mark.parametrize("stranded", [1, 2])
@pytest.mark.parametrize("fused", ["1", "0"])
def test_stranded_pass_in_combine_mode(seed, stranded, fused, ctx, oracle_lib, monkeypatch):
    """What `combine` over the kept reads of a STRANDED library takes (only the builder's fuzz had it): the stranded
    instantiation of the fused kernel (windows of 508 distinct positions) with combine_mode = 1 -- a flanking read counts toward
    beta2Simple (SpliSER_v0_1_8.py:529-536) -- and layout + range beside it, both chunk sizes."""
    arr, rs = randcase.make_case(seed + 1200, True)
    if arr.n == 0 or rs.n == 0:
        pytest.skip("empty case")
    big = _repeat(rs, 53)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, stranded, 1)
    monkeypatch.setenv("SPL_FUSED", fused)
    for chunk in ("2048", "4096"):
        monkeypatch.setenv("SPL_FORCE_CHUNK", chunk)
        got = _count_soa(ctx, arr, [(big, 0)], stranded, 1)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)


def _with_clips(rs, seed, fraction=0.6):
    """Soft / hard clips in front of and behind the reads' CIGARs, as a local aligner writes them: one op each side mostly, now and
    then two (H then S), now and then an insertion inside -- POS and the aligned part stay as they are."""
    rng = np.random.default_rng(seed)
    off = rs.cig_off.astype(np.int64)
    ops, offs = [], [0]
    for i in range(rs.n):
        mine = [int(x) for x in rs.cigar[off[i]:off[i + 1]]]
        if rng.random() < fraction and mine:
            how = int(rng.integers(0, 8))
            lead = [(int(rng.integers(1, 30)) << 4) | int(rng.choice([4, 5]))] if how & 1 else []
            trail = [(int(rng.integers(1, 30)) << 4) | int(rng.choice([4, 5]))] if how & 2 else []
            if how == 7:
                lead = [(3 << 4) | 5] + lead             # H S ... : two ops in front
            if how == 4 and (mine[0] >> 4) > 4 and (mine[0] & 15) == 0:      # an insertion inside the first aligned block
                a = mine[0] >> 4
                mine = [((a // 2) << 4), (2 << 4) | 1, ((a - a // 2) << 4)] + mine[1:]
            mine = lead + mine + trail
        ops.extend(mine)
        offs.append(len(ops))
    return samio.ReadSet(rs.pos, rs.flag, np.array(offs, np.uint32), np.array(ops, np.uint32))


@pytest.mark.parametrize("seed", range(0, 10))
@pytest.mark.parametrize("stranded", [0, 1])
def test_clipped_reads_take_the_straight_line_path_and_count_alike(seed, stranded, ctx, oracle_lib, monkeypatch):
    """Reads with one clip in front and / or behind are classified by the fused kernel's second straight-line tier
    (spl_pack.h: classify_clipped); two ops in front, or an insertion inside, go on to the general classifier.  To checkBam such
    ops are nothing (SpliSER_v0_1_8.py:457-464): the counters are the oracle's on the clipped CIGARs, and the same as layout +
    range's (whose layout kernel has no such tier)."""
    arr, rs = randcase.make_case(seed + 1300, bool(stranded))
    if arr.n == 0 or rs.n == 0:
        pytest.skip("empty case")
    big = _with_clips(_repeat(rs, 37), seed)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, stranded, 0)
    for fused in ("1", "0"):
        monkeypatch.setenv("SPL_FUSED", fused)
        got = _count_soa(ctx, arr, [(big, 0)], stranded)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)


@pytest.mark.parametrize("seed", range(0, 6))
@pytest.mark.parametrize("combine", [0, 1])
def test_segments_that_begin_anywhere(seed, combine, ctx, oracle_lib, monkeypatch):
    """Several segments laid end to end in one set of arrays, their lengths anything (1, 3, 2047, 4097 ... reads), so that segments
    begin and end at every offset inside a thread's four reads and inside a chunk's cell, and two segments share a cell; an
    empty segment between them."""
    arr, rs = randcase.make_case(seed + 950, False)
    if arr.n == 0 or rs.n == 0:
        pytest.skip("empty case")
    rng = np.random.default_rng(seed)
    big = _repeat(rs, 41)
    cuts = sorted(set(int(x) for x in rng.integers(0, big.n, 5)) | {1, 3, min(big.n, 2047), min(big.n, 4097)})
    bounds = [0] + [c for c in cuts if 0 < c < big.n] + [big.n]
    segs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        o0, o1 = int(big.cig_off[a]), int(big.cig_off[b])
        segs.append(samio.ReadSet(big.pos[a:b], big.flag[a:b], (big.cig_off[a:b + 1] - o0).astype(np.uint32), big.cigar[o0:o1]))
    segs.insert(2, samio.ReadSet.empty())
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, 0, combine)
    for chunk, fused in (("2048", "1"), ("4096", "1"), ("2048", "0"), ("4096", "0")):
        monkeypatch.setenv("SPL_FORCE_CHUNK", chunk)
        monkeypatch.setenv("SPL_FUSED", fused)
        got = _count_soa(ctx, arr, [(s, 0) for s in segs], 0, combine)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_cigars_longer_than_the_stage(fused, ctx, oracle_lib, monkeypatch):
    """Long-read CIGARs: hundreds of ops a read, so that a chunk's ops are many times the 4 ops a read the workgroup stages in
    LDS -- ops beyond the stage come from memory, WIDE reads' ops are read by the counting kernels from the array they came in."""
    monkeypatch.setenv("SPL_FUSED", fused)
    arr, _ = randcase.make_case(977, False)
    rng = np.random.default_rng(5)
    lo, hi = int(arr.pos.min()) - 50, int(arr.pos.max()) + 50
    pos, flag, offs, ops = [], [], [0], []
    for i in range(9000):
        start = int(rng.integers(max(1, lo - 400), hi))
        n = int(rng.integers(1, 260)) if i % 3 else int(rng.integers(1, 4))
        for k in range(n):
            code = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 7, 8]))
            ops.append((int(rng.integers(1, 9)) << 4) | code)
        pos.append(start); flag.append(int(rng.choice([0, 16]))); offs.append(len(ops))
    order = np.argsort(np.array(pos), kind="stable")
    rs0 = samio.ReadSet(np.array(pos, np.int32), np.array(flag, np.uint16), np.array(offs, np.uint32), np.array(ops, np.uint32))
    rs = _repeat(rs0, 1, order=lambda n: order)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, rs, 0, 0)
    got = _count_soa(ctx, arr, [(rs, 0)], 0)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)


def test_device_and_host_segments_in_one_read_set(ctx, oracle_lib):
    """A read set may hold segments packed on the host (spl_reads_add) beside segments laid out on the device (spl_reads_add_soa):
    one flat chunk list, one chunk order."""
    arr, rs = randcase.make_case(931, False)
    big = _repeat(rs, 53)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    w1 = ocount(arr, big, 0, 0)
    with ctx.upload_soa([_arrays(big)]) as soa, ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
        with ctx.begin_reads() as dr:
            dr.add(_arrays(big), 0)
            dr.add_soa(soa, 0, 0)
            dr.add(_arrays(big), 0)
            dr.finish()
            ctx.count_launch(ds, dr, 0, 0)
            got = ds.counters()
    for w, g in zip(w1, got):
        assert np.array_equal(3 * w.astype(np.int64), g.astype(np.int64))


def test_the_arrays_outlive_their_handle(ctx, oracle_lib):
    """WIDE reads' ops are read from the uploaded cigar array: the read set keeps it alive after spl_soa_free."""
    arr, rs = randcase.make_case(933, False)
    big = _repeat(rs, 29)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, 0, 0)
    soa = ctx.upload_soa([_arrays(big)])
    dr = ctx.layout_read_segments(soa, [0])
    soa.free()
    junk = ctx.upload_soa([_arrays(big)])       # (something else takes freed memory, were it freed)
    with ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
        ctx.count_launch(ds, dr, 0, 0)
        got = ds.counters()
    dr.free()
    junk.free()
    for w, g in zip(want, got):
        assert np.array_equal(w, g)


def test_shifted_segments_and_the_coordinate_limit(ctx):
    """A segment moved beyond the coordinate space is refused at spl_reads_add_soa, as spl_reads_add refuses it."""
    rs = samio.ReadSet(np.array([100], np.int32), np.array([0], np.uint16), np.array([0, 1], np.uint32), np.array([(50 << 4) | 0], np.uint32))
    with ctx.upload_soa([_arrays(rs)]) as soa:
        with ctx.begin_reads() as dr:
            with pytest.raises(native.SpliserNativeError):
                dr.add_soa(soa, 0, 2147483500)
            with pytest.raises(native.SpliserNativeError):
                dr.add_soa(soa, 3, 0)


def test_two_read_sets_laid_out_again_and_counted_in_turn(ctx, oracle_lib):
    """bench.py's step: every shard's records made again from its arrays, then counted; several steps in a row, the counters of
    the first and the last step the oracle's."""
    ocount, _ = helpers.oracle_engine(oracle_lib)
    shards = []
    for seed in (941, 942):
        arr, rs = randcase.make_case(seed, False)
        big = _repeat(rs, 151)
        soa = ctx.upload_soa([_arrays(big)])
        shards.append((ocount(arr, big, 0, 0), ctx.upload_sites(native.SiteArrays.from_chrom(arr)), ctx.layout_read_segments(soa, [0]), soa))
    try:
        for step in range(4):
            for _, ds, dr, _ in shards:
                dr.relayout()
                ctx.count_launch(ds, dr, 0, 0)
            ctx.pass_barrier()
            if step in (0, 3):
                for want, ds, _, _ in shards[::-1]:
                    for w, g in zip(want, ds.counters()):
                        assert np.array_equal(w, g)
    finally:
        for _, ds, dr, soa in shards:
            dr.free(); soa.free(); ds.free()


def test_a_fused_read_set_is_laid_out_when_a_pass_needs_records(ctx, oracle_lib):
    """A read set finished fused has no records in memory (spl_reads_layout_bytes: 0 written).  The pair kernel and spl_junctions
    read records: the set is laid out then, once, and counts the same before and after."""
    arr, rs = randcase.make_case(936, False)
    big = _repeat(rs, 97)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, 0, 0)
    want_fr = ocount(arr, big, 1, 0)
    with ctx.upload_soa([_arrays(big)]) as soa, ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
        dr = ctx.layout_read_segments(soa, [0])
        try:
            read, written = dr.layout_bytes()
            assert read == 10 * big.n + 4 * len(big.cigar) and written == 0
            ctx.count_launch(ds, dr, 0, 0)
            for w, g in zip(want, ds.counters()):
                assert np.array_equal(w, g)
            ctx.count_launch(ds, dr, 1, 0)                        # a stranded pass: fused as well
            for w, g in zip(want_fr, ds.counters()):
                assert np.array_equal(w, g)
            assert dr.layout_bytes()[1] == 0                      # (still fused)
            ctx.count_launch(ds, dr, 0, 0, native.OPT_WAVE_AGGREGATION)   # the merging variant reads records
            for w, g in zip(want, ds.counters()):
                assert np.array_equal(w, g)
            assert dr.layout_bytes()[1] > 0                       # (records in memory now)
            ctx.count_launch(ds, dr, 0, 0)                        # ... and the plain pass over them
            for w, g in zip(want, ds.counters()):
                assert np.array_equal(w, g)
        finally:
            dr.free()
        dr = ctx.layout_read_segments(soa, [0])
        try:
            ctx.count_launch(ds, dr, 0, 0, native.OPT_PAIR_KERNEL)   # the literal (read, site) kernel reads records
            for w, g in zip(want, ds.counters()):
                assert np.array_equal(w, g)
            assert dr.layout_bytes()[1] > 0
        finally:
            dr.free()
        dr = ctx.layout_read_segments(soa, [0])
        fused_junctions = dr.junctions()
        dr.free()
    with ctx.upload_reads(_arrays(big)) as packed:
        host_junctions = packed.junctions()
    assert len(host_junctions["left"]) > 0
    for k in host_junctions:
        assert np.array_equal(fused_junctions[k], host_junctions[k]), k


def test_segments_from_two_sets_of_arrays(ctx, oracle_lib):
    """A read set whose segments lie in two handles' arrays is not fused (the fused pass reads ONE set of arrays): both are laid
    out, each by its own launch, and counted together."""
    arr, rs = randcase.make_case(938, False)
    big = _repeat(rs, 61)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, 0, 0)
    with ctx.upload_soa([_arrays(big)]) as a, ctx.upload_soa([_arrays(big)]) as b, ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
        with ctx.begin_reads() as dr:
            dr.add_soa(a, 0, 0)
            dr.add_soa(b, 0, 0)
            dr.finish()
            assert dr.layout_bytes()[1] > 0          # (records in memory: not fused)
            ctx.count_launch(ds, dr, 0, 0)
            got = ds.counters()
    for w, g in zip(want, got):
        assert np.array_equal(2 * w.astype(np.int64), g.astype(np.int64))
