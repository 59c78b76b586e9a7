"""BAM decode on the device (spl_bam_decode_device: BGZF inflate, CRC32, record scan and extraction as HIP kernels) against the
host decoder on the same files: identical arrays per reference, identical record counts -- and the files the device path does
not take (unsorted, CG-tag CIGARs, damaged) end up with the host decoder, whose results and errors are the contract."""
import numpy as np
import pytest

from spliser_amd import native, samio
from test_bam_decode import _random_sets, _same, edge_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    native.build()
    with native.Context(0) as c:
        yield c


def _both(path, ctx, names, truth=None):
    host = native.BamFile(path, threads=4)
    dev = native.BamFile(path, threads=4, defer=True)
    took = dev.decode_on_device(ctx)
    assert dev.n_records == host.n_records
    for c in names:
        a, b = dev.reads(c), host.reads(c)
        if truth is not None:
            _same(b, truth[c])      # (the host decoder against what was written ...)
            _same(a, truth[c])      # (... and the device decoder)
        _same(a, b)
        assert dev.wait_ref(c) == host.wait_ref(c)
    host.close()
    dev.close()
    return took


@pytest.mark.parametrize("seq_mode,level", [(0, 1), (1, 1), (1, 6), (1, 0), (2, 6), (2, 1), (2, 9)])
def test_device_decode_matches_host(ctx, tmp_path, seq_mode, level):
    names, sets = _random_sets(21 + seq_mode + level, 50_000, 4)
    path = str(tmp_path / "d.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
    assert _both(path, ctx, names, sets) is True


def test_python_writer_files_unplaced_and_empty_references(ctx, tmp_path):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, sets = samio.read_sam(os.path.join(root, "tests", "golden", "random_b", "reads.sam"))
    path = str(tmp_path / "x.bam")
    samio.write_bam(path, names + ["empty_ref"], [10 ** 8] * (len(names) + 1), [(c, sets[c]) for c in names if c in sets], with_seq=True, unplaced=5)
    assert _both(path, ctx, names + ["empty_ref"]) is True


def test_files_the_device_path_hands_to_the_host(ctx, tmp_path):
    names, sets = _random_sets(4, 20_000, 3)
    # records of an earlier reference after a later one
    path = str(tmp_path / "unsorted.bam")
    native.write_bam(path, names, [10 ** 8] * 3, [sets[names[0]], sets[names[1]], sets[names[2]]], level=1, threads=2, seq_mode=1)
    shuffled = str(tmp_path / "shuffled.bam")
    samio.write_bam(shuffled, names, [10 ** 8] * 3, [(names[1], sets[names[1]]), (names[0], sets[names[0]])], with_seq=True)
    dev = native.BamFile(shuffled, threads=2, defer=True)
    assert dev.decode_on_device(ctx) is False
    assert dev.wait_all() is False          # (the host decoder's verdict: not sorted by reference)
    dev.close()
    # CIGARs parked in CG tags
    tagged = str(tmp_path / "cg.bam")
    samio.write_bam(tagged, names, [10 ** 8] * 3, [(c, sets[c]) for c in names], with_seq=True, long_cigar_tag=True)
    assert _both(tagged, ctx, names) is False
    # a flipped byte in the middle: the device notices (inflate or CRC), the host decoder reports
    data = bytearray(open(path, "rb").read())
    data[len(data) // 2] ^= 0xFF
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(data))
    dev = native.BamFile(bad, threads=2, defer=True)
    assert dev.decode_on_device(ctx) is False
    with pytest.raises(native.SpliserNativeError):
        dev.wait_all()
    dev.close()


def test_records_larger_than_a_block_on_the_device(ctx, tmp_path):
    pos = np.array([10, 20, 30, 40, 50, 60], np.int32)
    ops = [[(50 << 4) | 0], [(300_000 << 4) | 0], [(20 << 4) | 0, (100 << 4) | 3, (30 << 4) | 0], [(700_000 << 4) | 0],
           [(700_000 << 4) | 0], [(75 << 4) | 0]]
    cig_off = np.concatenate(([0], np.cumsum([len(o) for o in ops]))).astype(np.uint32)
    cigar = np.array([x for o in ops for x in o], np.uint32)
    want = samio.ReadSet(pos, np.array([0, 16, 0, 0, 16, 0], np.uint16), cig_off, cigar)
    path = str(tmp_path / "big.bam")
    samio.write_bam(path, ["c0"], [10 ** 8], [("c0", want)], with_seq=True)
    dev = native.BamFile(path, threads=2, defer=True)
    dev.decode_on_device(ctx)               # (either path may take it: the result is what counts)
    _same(dev.reads("c0"), want)
    dev.close()


@pytest.mark.parametrize("stranded", [None, "fr"])
def test_process_with_gpu_decode_writes_the_same_file(tmp_path, stranded):
    from spliser_amd import synth
    from spliser_amd.process import process
    name = "mouse_stranded" if stranded else "arabidopsis"
    wl = synth.Workload(name, scale=0.01, seed=9)
    prefix = str(tmp_path / "s")
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=bool(stranded))
    synth.write_gff(prefix + ".gff", wl.genome)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    texts = []
    for gpu in (False, True):
        out = prefix + (".gpu" if gpu else ".host")
        process(prefix + ".bam", prefix + ".bed", out, annotationFile=prefix + ".gff", isStranded=bool(stranded), strandedType=stranded,
                isbeta2Cryptic=bool(stranded), log=lambda m: None, gpuDecode=gpu)
        texts.append(open(out + ".SpliSER.tsv").read())
    assert texts[0] == texts[1] and texts[0].count("\n") > 100


def test_many_small_files_all_block_kinds(ctx, tmp_path):
    """Small files through the Python writer at every zlib level: stored blocks (level 0), fixed-code blocks (a handful of
    records), dynamic blocks, blocks of a few bytes, files of one record -- device decode against what was written."""
    rng = np.random.default_rng(77)
    taken = 0
    for k in range(36):
        n = int(rng.choice([1, 2, 5, 40, 300, 3000]))
        pos = np.sort(rng.integers(1, 100000, n)).astype(np.int32)
        n_ops = rng.choice([1, 3, 5], size=n)
        cig_off = np.concatenate(([0], np.cumsum(n_ops))).astype(np.uint32)
        cigar = np.empty(int(cig_off[-1]), np.uint32)
        j = 0
        for i in range(n):
            for q in range(int(n_ops[i])):
                cigar[j] = (int(rng.integers(1, 200)) << 4) | (0 if q % 2 == 0 else 3)
                j += 1
        want = samio.ReadSet(pos, rng.choice([0, 16, 99, 147], size=n).astype(np.uint16), cig_off, cigar)
        path = str(tmp_path / ("f%d.bam" % k))
        samio.write_bam(path, ["a", "b"], [10 ** 6, 10 ** 6], [("b" if k % 3 == 0 else "a", want)], level=k % 10, with_seq=bool(k % 2),
                        unplaced=k % 4)
        dev = native.BamFile(path, threads=2, defer=True)
        taken += bool(dev.decode_on_device(ctx))
        ref = "b" if k % 3 == 0 else "a"
        _same(dev.reads(ref), want)
        assert dev.reads("a" if ref == "b" else "b").n == 0
        assert dev.n_records == n + k % 4
        dev.close()
    assert taken == 36


@pytest.mark.parametrize("strategy", ["filtered", "huffman_only", "rle", "fixed"])
def test_deflate_strategies(ctx, tmp_path, monkeypatch, strategy):
    """The same records deflated the ways zlib can be told to: Huffman codes only (no matches at all), run-length matches only
    (every distance is 1), fixed codes throughout, the 'filtered' heuristics -- block shapes an aligner's BAM never has and a
    DEFLATE decoder must read all the same."""
    import struct
    import zlib
    strat = {"filtered": zlib.Z_FILTERED, "huffman_only": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE, "fixed": zlib.Z_FIXED}[strategy]

    def block(payload, level):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15, 8, strat)
        data = comp.compress(payload) + comp.flush()
        if len(data) + 26 > 0x10000:      # (Huffman-only output of incompressible bytes can outgrow a block: store it instead)
            comp = zlib.compressobj(0, zlib.DEFLATED, -15)
            data = comp.compress(payload) + comp.flush()
        header = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 0x42, 0x43, 2, len(data) + 25)
        return header + data + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)
    monkeypatch.setattr(samio, "_bgzf_block", block)
    names, sets = _random_sets(31, 6_000, 2)
    path = str(tmp_path / "s.bam")
    samio.write_bam(path, names, [10 ** 8] * 2, [(c, sets[c]) for c in names], with_seq=True)
    assert _both(path, ctx, names, sets) is True


@pytest.mark.parametrize("seed", range(0, 10))
@pytest.mark.parametrize("stranded", [0, 1, 2])
def test_reads_laid_out_on_the_device_count_like_the_oracle(seed, stranded, ctx, oracle_lib, tmp_path, monkeypatch):
    """The device packer (spl_devpack.hip) on what the host packer's tests use: adversarial reads -- 0N ops, adjacent N ops,
    clipped / indel / = / X CIGARs, unmapped-but-placed records, reads of every class -- each repeated so that runs of every
    class cross thread, wave and chunk boundaries of the layout kernels.  BAM -> device decode -> device layout -> counting
    kernels, against the oracle on the arrays that were written; both chunk sizes."""
    import helpers
    import randcase
    arr, rs = randcase.make_case(seed + 700, bool(stranded))
    if arr.n == 0 or rs.n == 0:
        pytest.skip("empty case")
    times = 61
    n_ops = np.diff(rs.cig_off.astype(np.int64))
    keep = np.repeat(np.arange(rs.n), times)
    op_idx = np.concatenate([np.arange(rs.cig_off[i], rs.cig_off[i + 1]) for i in keep]) if rs.n else np.zeros(0, np.int64)
    big = samio.ReadSet(rs.pos[keep], rs.flag[keep], np.concatenate(([0], np.cumsum(n_ops[keep]))).astype(np.uint32),
                        rs.cigar[op_idx.astype(np.int64)])
    path = str(tmp_path / "adv.bam")
    samio.write_bam(path, ["c1"], [10 ** 8], [("c1", big)], level=6)
    ocount, _ = helpers.oracle_engine(oracle_lib)
    want = ocount(arr, big, stranded, 0)
    for chunk in ("2048", "4096"):
        monkeypatch.setenv("SPL_FORCE_CHUNK", chunk)
        bam = native.BamFile(path, threads=2, defer=True)
        assert bam.decode_on_device(ctx)
        with ctx.upload_sites(native.SiteArrays.from_chrom(arr)) as ds:
            with ctx.begin_reads() as dr:
                dr.add_bam(bam, "c1", 0)
                dr.finish()
                ctx.count_launch(ds, dr, stranded, 0)
                got = ds.counters()
        bam.close()
        for w, g in zip(want, got):
            assert np.array_equal(w, g)


@pytest.mark.parametrize("room", ["600", "4096", "20000"])
def test_a_block_that_needs_more_token_room_than_its_file_was_given(ctx, tmp_path, monkeypatch, room):
    """The decoding kernel's token room per block is sized from the file's mean compressed block (a third of the worst case for real
    files); a block that needs more says so (SPL_Z_TOKENS) and the share is decoded again with the worst case's room -- in whole
    and in shares, with windows of a few blocks, the result what the host decoder gives."""
    monkeypatch.setenv("SPL_Z_TOKEN_ROOM", room)
    for window, seq_mode, level, seed in ((None, 1, 1, 51), ("5", 2, 6, 52), (None, 1, 0, 53)):
        if window:
            monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", window)
        names, sets = _random_sets(seed, 30_000, 3)
        path = str(tmp_path / ("t%d.bam" % seed))
        native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
        assert _both(path, ctx, names, sets) is True


@pytest.mark.parametrize("window", ["2", "3", "7", "16", "40"])
def test_inflate_windows_of_a_few_blocks(ctx, tmp_path, monkeypatch, window):
    """The stream is inflated, scanned and emptied of its records a window at a time (49 152 blocks in production): windows of a few
    blocks on files of a few hundred -- every window ends in the middle of a record, blocks near a window's end are done
    again by the next, the extracted arrays grow as they go -- must give what the host decoder gives; records larger than a
    window are the host decoder's."""
    monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", window)
    for seq_mode, level, seed in ((1, 1, 31), (1, 6, 32), (0, 1, 33), (1, 0, 34), (2, 6, 35)):
        names, sets = _random_sets(seed, 30_000, 3)
        path = str(tmp_path / ("w%d.bam" % seed))
        native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
        assert _both(path, ctx, names, sets) is True
    # reads of 300 000 and 700 000 bases with SEQ and QUAL (records of 450 KB and a megabyte) between ordinary ones
    pos = np.array([10, 20, 30, 40, 50, 60], np.int32)
    ops = [[(50 << 4) | 0], [(300_000 << 4) | 0], [(20 << 4) | 0, (100 << 4) | 3, (30 << 4) | 0], [(700_000 << 4) | 0],
           [(700_000 << 4) | 0], [(75 << 4) | 0]]
    cig_off = np.concatenate(([0], np.cumsum([len(o) for o in ops]))).astype(np.uint32)
    want = samio.ReadSet(pos, np.array([0, 16, 0, 0, 16, 0], np.uint16), cig_off, np.array([x for o in ops for x in o], np.uint32))
    path = str(tmp_path / "big.bam")
    samio.write_bam(path, ["c0"], [10 ** 8], [("c0", want)], with_seq=True)
    dev = native.BamFile(path, threads=2, defer=True)
    dev.decode_on_device(ctx)      # (a megabyte is 17 blocks: the record's bytes travel from window to window in the room in
    _same(dev.reads("c0"), want)   #  front of each; what counts is the result)
    dev.close()


@pytest.mark.parametrize("dense", ["0", "1"])
def test_either_decoding_kernel_on_every_kind_of_file(ctx, tmp_path, monkeypatch, dense):
    """SPL_Z_DENSE: the decoding kernel for blocks of literals (0) and the denser one for blocks that deflate well (1), each forced
    onto files of both kinds -- sequence-like at level 1, htslib-shaped at level 6, constant bytes, stored blocks -- windows of a
    few blocks: what the host decoder gives, either way."""
    monkeypatch.setenv("SPL_Z_DENSE", dense)
    monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", "11")
    for seq_mode, level, seed in ((1, 1, 71), (2, 6, 72), (0, 1, 73), (1, 0, 74), (2, 9, 75), (1, 6, 76)):
        names, sets = _random_sets(seed, 30_000, 3)
        path = str(tmp_path / ("d%d.bam" % seed))
        native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
        assert _both(path, ctx, names, sets) is True


@pytest.mark.parametrize("levels", ["0", "1", "2", "3", "4"])
def test_stream_set_ups_of_the_decode(ctx, tmp_path, monkeypatch, levels):
    """SPL_STREAM_PRIORITIES: the decode's streams at priority levels of their own and kept by the process (1, the default), all at
    the normal level and made per call as in rounds 3-5 (0), and the variants the profile measured (2: the short kernels' stream at
    the low level; 3: made per call; 4: the file's pieces on the context's copy stream) -- windows of a few blocks, several calls
    in a row (the second finds the first's streams), the same reads every way."""
    monkeypatch.setenv("SPL_STREAM_PRIORITIES", levels)
    monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", "9")
    for seq_mode, level, seed in ((1, 1, 51), (2, 6, 52), (1, 1, 53)):
        names, sets = _random_sets(seed, 30_000, 3)
        path = str(tmp_path / ("p%d.bam" % seed))
        native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
        assert _both(path, ctx, names, sets) is True


@pytest.mark.parametrize("bufs", [("1", "1"), ("2", "1"), ("1", "3"), ("3", "2"), ("4", "4")])
def test_windows_in_flight(ctx, tmp_path, monkeypatch, bufs):
    """A window's inflated bytes (copying kernel -> CRC32, scan, extraction) and its tokens (decoding kernel -> copying kernel) have
    buffers of their own, n_buf and n_zw of them, and every kernel waits for the reader of ITS buffer (the decoding of window
    k + n_zw for the copying of window k, the copying of window k + n_buf for the extraction of window k): any numbers of the two,
    windows of 3 and 16 blocks on files of a few hundred, must give what the host decoder gives."""
    monkeypatch.setenv("SPL_INFLATE_BUFFERS", bufs[0])
    monkeypatch.setenv("SPL_INFLATE_TOKEN_BUFFERS", bufs[1])
    for window, seq_mode, level, seed in (("3", 1, 1, 41), ("16", 2, 6, 42), ("3", 0, 6, 43)):
        monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", window)
        names, sets = _random_sets(seed, 30_000, 3)
        path = str(tmp_path / ("f%d.bam" % seed))
        native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=3, seq_mode=seq_mode)
        assert _both(path, ctx, names, sets) is True


@pytest.mark.parametrize("late", [None, "1"])
def test_references_complete_before_the_decoder_has_cleared_up(tmp_path, monkeypatch, late):
    """`process` counts as soon as the device decoder has made the file's references complete (spl_bam_wait_device), while the
    decoder's thread still holds its streams, events and lists (and stands aside until the file is closed); SPL_PUBLISH_LATE=1 is
    the order until round 4.  Either way the .SpliSER.tsv is the host decoder's, and closing the file ends the decoder's thread."""
    import threading
    from spliser_amd import synth
    from spliser_amd.process import process, wait_deferred_close
    if late:
        monkeypatch.setenv("SPL_PUBLISH_LATE", late)
    wl = synth.Workload("arabidopsis", scale=0.02, seed=14)
    prefix = str(tmp_path / "p")
    wl.write_inputs(prefix, bam=False)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    n_before = threading.active_count()
    tm = process(prefix + ".bam", prefix + ".bed", prefix + ".dev", annotationFile=prefix + ".gff", log=lambda m: None)
    assert tm["bam_decode"] == "device"
    process(prefix + ".bam", prefix + ".bed", prefix + ".host", annotationFile=prefix + ".gff", gpuDecode=False, log=lambda m: None)
    assert open(prefix + ".dev.SpliSER.tsv").read() == open(prefix + ".host.SpliSER.tsv").read()
    wait_deferred_close()
    assert threading.active_count() <= n_before      # (no decoder thread left lingering behind a closed file)


@pytest.mark.parametrize("window", [None, "5"])
@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0, 0), (0,) * 7])
def test_process_decodes_in_shares(tmp_path, monkeypatch, devices, window):
    """Several contexts (here on one GPU): the file is cut into equal stretches at any BGZF block, every context inflates and
    extracts its own stretch and counts it -- chromosomes that lie across contexts by adding the contexts' counters -- and the
    .SpliSER.tsv is the one the host decoder's reads give."""
    from spliser_amd import synth
    from spliser_amd.process import process
    wl = synth.Workload("arabidopsis", scale=0.02, seed=12)
    prefix = str(tmp_path / "s")
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions)
    synth.write_gff(prefix + ".gff", wl.genome)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    process(prefix + ".bam", prefix + ".bed", prefix + ".host", annotationFile=prefix + ".gff", log=lambda m: None, gpuDecode=False)
    if window:
        monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", window)
    seen = {}
    real = native.BamFile.decode_on_devices_async
    monkeypatch.setattr(native.BamFile, "decode_on_devices_async", lambda self, devs: seen.setdefault("plan", real(self, devs)))
    tm = process(prefix + ".bam", prefix + ".bed", prefix + ".dev", annotationFile=prefix + ".gff", log=lambda m: None, devices=devices)
    assert tm["bam_decode"] == "device"
    plan = seen["plan"]
    assert len(plan) == len(devices) and sorted(set(c for _, names in plan for c in names)) == sorted(wl.genome.chrom_names)
    assert open(prefix + ".dev.SpliSER.tsv").read() == open(prefix + ".host.SpliSER.tsv").read()


def test_reads_of_every_share_come_back_to_the_host(tmp_path):
    """A decode in shares keeps every share's reads on its device; a reader on the host (``BamFile.reads``) gets copies from all
    of them, each reference from the share that holds it."""
    names, sets = _random_sets(41, 20_000, 4)
    path = str(tmp_path / "u.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    bam = native.BamFile(path, defer=True)
    plan = bam.decode_on_devices_async([0, 0, 0])
    assert bam.join_decoders() is True and len(plan) == 3
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()


@pytest.mark.parametrize("n_shares", [2, 3, 5, 8])
def test_shares_over_long_records_an_empty_reference_and_an_unplaced_tail(tmp_path, n_shares):
    """What a share's edge may fall on: records longer than a BGZF block (the first record of a share then begins several blocks
    past its first block), a reference without reads between two that have them, and records without a reference after the last
    one.  Whichever decoder ends up with the file, every read is there once and ``n_records`` counts the unplaced ones."""
    path = str(tmp_path / "edges.bam")
    names, sets = edge_file(path, 5 + n_shares)
    bam = native.BamFile(path, defer=True)
    try:
        bam.decode_on_devices_async([0] * n_shares)
    except native.SpliserNativeError:
        pass
    bam.join_decoders()
    assert bam.n_records == 3 * 900 + 700
    for c in names:
        got = bam.reads(c)
        if sets[c].n == 0:
            assert got is None or got.n == 0
        else:
            _same(got, sets[c])
    bam.close()


def _take(rs, lo, hi):
    o0, o1 = int(rs.cig_off[lo]), int(rs.cig_off[hi])
    return samio.ReadSet(rs.pos[lo:hi], rs.flag[lo:hi], rs.cig_off[lo:hi + 1] - rs.cig_off[lo], rs.cigar[o0:o1])


@pytest.mark.parametrize("stray_of,after", [("c0", 40), ("c3", 40), ("c0", 6000)])
def test_a_stray_record_at_a_shares_edge_is_not_lost(tmp_path, stray_of, after):
    """a decode in shares skips records of other references in the first and last two blocks of a share as the
    neighbour's.  In a file that is NOT sorted by reference such a record is nobody's: one record of c0 (or of c3) among the first
    records of c2, where the second of two shares begins.  The file has to end up with the host decoder, whose reads are all there."""
    names, sets = _random_sets(97, 6_000, 4)
    stray = _take(sets[stray_of], 0, 1)
    after = min(after, sets["c2"].n)
    order = [("c0", sets["c0"]), ("c1", sets["c1"]), ("c2", _take(sets["c2"], 0, after)), (stray_of, stray)]
    if after < sets["c2"].n:
        order.append(("c2", _take(sets["c2"], after, sets["c2"].n)))
    order.append(("c3", sets["c3"]))
    path = str(tmp_path / "stray.bam")
    samio.write_bam(path, names, [10 ** 8] * len(names), order, with_seq=True)
    bam = native.BamFile(path, defer=True)
    try:
        bam.decode_on_devices_async([0, 0])
    except native.SpliserNativeError:
        pass                                         # (a plan that cannot be made: the host decoder's file as well)
    on_device = bam.join_decoders()
    got = {c: bam.reads(c) for c in names}
    n = sum(r.n for r in got.values() if r is not None)
    assert n == sum(sets[c].n for c in names) + 1, "a record was extracted by nobody (device: %s)" % on_device
    assert got[stray_of].n == sets[stray_of].n + 1
    bam.close()


@pytest.mark.parametrize("ring,window", [("4", None), ("5", "24"), ("9", "40"), ("6", "7")])
def test_the_file_image_is_a_ring_on_the_device(tmp_path, monkeypatch, ring, window):
    """Only a few pieces of the file are on the device at a time (spl_capi.cpp decode_share: a piece's slot goes to the piece R
    further on when the last window that reads it is decoded).  Staging buffers of 2 MB and a ring of 4-9 of them for a file of
    a dozen pieces: every slot is reused several times, windows end where their bytes would not fit the ring."""
    monkeypatch.setenv("SPL_STAGE_MB", "2")
    monkeypatch.setenv("SPL_IMAGE_RING_PIECES", ring)
    if window:
        monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", window)
    names, sets = _random_sets(53, 90_000, 4)
    path = str(tmp_path / "ring.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=3, seq_mode=1)
    import os
    assert os.path.getsize(path) > 8 * (2 << 20)
    with native.Context(0) as ctx:      # (a context of its own: its staging buffers have the size asked for here)
        assert _both(path, ctx, names, sets) is True
    bam = native.BamFile(path, defer=True)
    plan = bam.decode_on_devices_async([0, 0, 0])
    assert bam.join_decoders() is True and len(plan) == 3
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()


def test_a_file_larger_than_the_free_device_memory(tmp_path, monkeypatch):
    """device memory for the decode is O(window) + what is extracted, not O(file).  A device that "has" 340 MB
    free decodes a file of 350 MB on the device (a ring of 2 MB slots, windows of 48 blocks; what it sets aside for the
    reads it extracts, a fifth of the inflated stream, is most of that); with 24 MB it declines, says why, and the host threads
    give the same reads."""
    import os
    monkeypatch.setenv("SPL_STAGE_MB", "2")
    monkeypatch.setenv("SPL_INFLATE_WINDOW_BLOCKS", "48")
    from spliser_amd import synth
    wl = synth.Workload("arabidopsis", scale=0.25, seed=14, workers=2)       # 5 M reads of 150 bp: 73 bytes a read in the file, 19 extracted
    names = wl.genome.chrom_names
    sets = dict(zip(names, wl.reads))
    path = str(tmp_path / "big.bam")
    native.write_bam(path, names, wl.genome.chrom_lengths, wl.reads, level=1, threads=3, seq_mode=1)
    size_mb = os.path.getsize(path) / 2 ** 20
    assert size_mb > 345
    native.lib().spl_trim(-1)        # (nothing held from earlier calls: the limit is all there is)
    monkeypatch.setenv("SPL_DEV_FREE_LIMIT_MB", "340")
    with native.Context(0) as ctx:
        dev = native.BamFile(path, threads=4, defer=True)
        took = dev.decode_on_device(ctx)
        assert took is True and dev.decline_reason() == "", dev.decline_reason()
        for c in names:
            _same(dev.reads(c), sets[c])
        dev.close()
        native.lib().spl_trim(-1)
        monkeypatch.setenv("SPL_DEV_FREE_LIMIT_MB", "24")
        dev = native.BamFile(path, threads=4, defer=True)
        assert dev.decode_on_device(ctx) is False and "device memory" in dev.decline_reason()
        for c in names:
            _same(dev.reads(c), sets[c])
        dev.close()


@pytest.mark.parametrize("stranded", [None, "rf"])
def test_the_command_line_as_a_child_process(tmp_path, stranded):
    """``python -m spliser_amd process`` the way a user runs it -- a fresh interpreter that leaves through ``os._exit`` (``__main__``) --
    writes the file the function writes in this process, twice (the second child finds the first's file in the page cache); a usage
    error still ends with argparse's exit code."""
    import os
    import subprocess
    import sys
    from spliser_amd import synth
    from spliser_amd.process import process
    wl = synth.Workload("mouse_stranded" if stranded else "arabidopsis", scale=0.01, seed=21)
    prefix = str(tmp_path / "s")
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=bool(stranded))
    synth.write_gff(prefix + ".gff", wl.genome)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    process(prefix + ".bam", prefix + ".bed", prefix + ".here", annotationFile=prefix + ".gff", isStranded=bool(stranded), strandedType=stranded,
            log=lambda m: None)
    want = open(prefix + ".here.SpliSER.tsv").read()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k, extra in enumerate(({}, {})):
        argv = [sys.executable, "-m", "spliser_amd", "process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-A", prefix + ".gff", "-o", prefix + ".cli%d" % k]
        if stranded:
            argv += ["--isStranded", "-s", stranded]
        r = subprocess.run(argv, cwd=root, env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:]
        assert open(prefix + ".cli%d.SpliSER.tsv" % k).read() == want
    r = subprocess.run([sys.executable, "-m", "spliser_amd", "process", "-B", prefix + ".bam"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=300)
    assert r.returncode == 2 and "required" in r.stdout
