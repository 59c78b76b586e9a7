"""The BAM decoder's batch scheme (bam_reader.cpp, decode_worker): records straddle blocks and batches, first record boundaries are
guessed per batch and accepted only when the sequential walk arrives there.  Every way through it must give back exactly the
records that were written: default batches, one block per batch (nearly every batch starts inside a record), the forced
sequential path, records larger than a batch, one thread and many."""
import numpy as np
import pytest

from spliser_amd import native, samio


@pytest.fixture(scope="module")
def built():
    native.build()
    return native.lib()


def _same(got, want):
    assert got.n == want.n
    assert np.array_equal(got.pos, want.pos) and np.array_equal(got.flag, want.flag)
    assert np.array_equal(got.cig_off, want.cig_off) and np.array_equal(got.cigar, want.cigar)
    assert got.max_end == want.max_end


def _random_sets(seed, n_per_ref, n_ref):
    rng = np.random.default_rng(seed)
    names = ["c%d" % k for k in range(n_ref)]
    sets = {}
    for c in names:
        n = int(n_per_ref * rng.uniform(0.5, 1.5))
        pos = np.sort(rng.integers(1, 5_000_000, n)).astype(np.int32)
        n_ops = rng.choice([1, 3, 5, 7, 2], size=n, p=[0.6, 0.25, 0.08, 0.02, 0.05])
        cig_off = np.concatenate(([0], np.cumsum(n_ops))).astype(np.uint32)
        cigar = np.empty(int(cig_off[-1]), np.uint32)
        k = 0
        for i in range(n):
            m = int(n_ops[i])
            for j in range(m):
                if m == 2:
                    code, length = (4, 5) if j == 0 else (0, 70)
                else:
                    code, length = (0, int(rng.integers(10, 80))) if j % 2 == 0 else (3, int(rng.integers(70, 4000)))
                cigar[k] = (length << 4) | code
                k += 1
        flag = rng.choice([0, 16, 99, 147, 83, 163, 256, 1024], size=n).astype(np.uint16)
        sets[c] = samio.ReadSet(pos, flag, cig_off, cigar)
    return names, sets


@pytest.mark.parametrize("env", [dict(), dict(SPL_BAM_BATCH_BLOCKS="1"), dict(SPL_BAM_BATCH_BLOCKS="3"), dict(SPL_BAM_FORCE_RESYNC="1"),
                                 dict(SPL_BAM_BATCH_BLOCKS="2", SPL_BAM_FORCE_RESYNC="1")])
@pytest.mark.parametrize("seq_mode", [0, 1, 2])
def test_many_batches(built, tmp_path, monkeypatch, env, seq_mode):
    names, sets = _random_sets(11 + seq_mode, 60_000, 3)
    path = str(tmp_path / "m.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=3, seq_mode=seq_mode)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for threads in (1, 3, 8):
        with_stream = threads == 3
        bam = native.BamFile(path, threads=threads, stream=with_stream)
        assert bam.n_records == sum(s.n for s in sets.values())
        for c in names:
            _same(bam.reads(c), sets[c])
        bam.close()


@pytest.mark.parametrize("blocks", ["1", "2", "8"])
def test_records_larger_than_a_batch(built, tmp_path, monkeypatch, blocks):
    # reads of 300 000 and 700 000 bases with SEQ and QUAL: records of 450 KB and over a megabyte between ordinary ones
    pos = np.array([10, 20, 30, 40, 50, 60], np.int32)
    ops = [[(50 << 4) | 0], [(300_000 << 4) | 0], [(20 << 4) | 0, (100 << 4) | 3, (30 << 4) | 0], [(700_000 << 4) | 0],
           [(700_000 << 4) | 0], [(75 << 4) | 0]]
    cig_off = np.concatenate(([0], np.cumsum([len(o) for o in ops]))).astype(np.uint32)
    cigar = np.array([x for o in ops for x in o], np.uint32)
    want = samio.ReadSet(pos, np.array([0, 16, 0, 0, 16, 0], np.uint16), cig_off, cigar)
    path = str(tmp_path / "big.bam")
    samio.write_bam(path, ["c0"], [10 ** 8], [("c0", want)], with_seq=True)
    monkeypatch.setenv("SPL_BAM_BATCH_BLOCKS", blocks)
    for threads in (1, 4):
        bam = native.BamFile(path, threads=threads)
        assert bam.n_records == 6
        _same(bam.reads("c0"), want)
        bam.close()


def test_broken_directory_behind_a_good_start_is_reported(built, tmp_path):
    """The blocks are walked beside the decode: a file whose first blocks and EOF marker are fine and whose middle is not must
    fail the decode (every way of waiting for it), not end early."""
    names, sets = _random_sets(5, 40_000, 2)
    path = str(tmp_path / "t.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    data = bytearray(open(path, "rb").read())
    # find a block header in the second half and break its magic
    at = data.find(b"\x1f\x8b\x08\x04", len(data) // 2)
    assert at > 0
    data[at + 1] = 0
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(data))
    with pytest.raises(native.SpliserNativeError) as e1:
        native.BamFile(bad, threads=2)
    assert "not BGZF" in str(e1.value)
    bam = native.BamFile(bad, threads=2, stream=True)      # the header is fine: the opening call succeeds
    with pytest.raises(native.SpliserNativeError):
        bam.wait_all()
    with pytest.raises(native.SpliserNativeError):
        bam.wait_ref(names[-1])
    bam.close()
    # the end cut off in the middle of a block: refused by the opening call itself
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(bytes(open(path, "rb").read()[:-5000]))
    with pytest.raises(native.SpliserNativeError) as e2:
        native.BamFile(cut, threads=2, stream=True)
    assert "truncated" in str(e2.value) or "corrupt" in str(e2.value)


def _directory(path, monkeypatch, one_thread, least=None, threads=4):
    """The whole block directory of a file, as the device decoder asks for it (internal entry points, by their C++ names)."""
    import ctypes
    lib = native.lib()
    walk = getattr(lib, "_Z16spl_bam_walk_allP7spl_bam")
    count = getattr(lib, "_Z19spl_bam_block_countPK7spl_bam")
    get = getattr(lib, "_Z17spl_bam_block_getPK7spl_bammP18spl_bam_block_info")
    count.restype = ctypes.c_size_t

    class Info(ctypes.Structure):
        _fields_ = [("data_off", ctypes.c_uint64), ("uoff", ctypes.c_uint64), ("data_len", ctypes.c_uint32), ("isize", ctypes.c_uint32),
                    ("crc", ctypes.c_uint32)]
    if one_thread:
        monkeypatch.setenv("SPL_WALK_ONE_THREAD", "1")
    else:
        monkeypatch.delenv("SPL_WALK_ONE_THREAD", raising=False)
        monkeypatch.setenv("SPL_WALK_PARALLEL_MIN", str(least))
    bam = native.BamFile(path, threads=threads, defer=True)
    try:
        rc = walk(bam._h)
        if rc:
            return rc, lib.spl_last_error().decode()
        out = []
        info = Info()
        for i in range(count(bam._h)):
            get(bam._h, ctypes.c_size_t(i), ctypes.byref(info))
            out.append((info.data_off, info.uoff, info.data_len, info.isize, info.crc))
        return 0, out
    finally:
        bam.close()


@pytest.mark.parametrize("least", [4096, 100_000, 1 << 20])
def test_directory_by_several_threads_is_the_one_thread_directory(built, tmp_path, monkeypatch, least):
    names, sets = _random_sets(11, 60_000, 3)
    path = str(tmp_path / "t.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    rc, want = _directory(path, monkeypatch, True)
    assert rc == 0 and len(want) > 50
    for threads in (2, 8):
        rc, got = _directory(path, monkeypatch, False, least, threads)
        assert rc == 0 and got == want
    # a file that is not what it says in its second half: the same refusal, in the same words, either way
    data = bytearray(open(path, "rb").read())
    at = data.find(b"\x1f\x8b\x08\x04", len(data) // 2)
    data[at + 1] = 0
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(data))
    rc1, text1 = _directory(bad, monkeypatch, True)
    rc2, text2 = _directory(bad, monkeypatch, False, least)
    assert rc1 != 0 and (rc1, text1) == (rc2, text2) and "not BGZF" in text1


def test_a_reservation_nobody_takes_up_ends_with_the_host_decoder(built, tmp_path):
    """decode_on_device_async reserves the file for the device decoder before its thread has a context; when that thread fails
    on the way (no device, no memory) the file must not wait for ever for a decoder that will not come."""
    names, sets = _random_sets(3, 5_000, 2)
    path = str(tmp_path / "r.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    bam = native.BamFile(path, threads=2, defer=True)
    assert native.lib().spl_bam_reserve_device(bam._h) == 0
    bam.start_host_decode()
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()
    # ... and through the thread that decode_on_device_async starts, with a device that does not exist
    bam = native.BamFile(path, threads=2, defer=True)
    t = bam.decode_on_device_async(device=4096)
    t.join()
    assert bam.device_error is not None and not bam.on_device
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()
    # ... and whoever waits for the OUTCOME of the device decoders (process Step 3: join_decoders -> spl_bam_wait_device) is told
    # "the host threads have it" as soon as that is so -- not after their decode, and not after the decoder thread's exit
    bam = native.BamFile(path, threads=2, defer=True)
    bam.decode_on_device_async(device=4096)
    assert bam.join_decoders() is False
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()
    bam = native.BamFile(path, threads=2, defer=True)
    bam.decode_on_devices_async([4096, 4097])
    assert bam.join_decoders() is False and bam.wait_all() is True
    bam.close()
    # a file closed while somebody waits for that outcome: the waiter returns (the reservation never taken up, the file cancelled)
    import threading
    bam = native.BamFile(path, threads=2, defer=True)
    assert native.lib().spl_bam_reserve_device(bam._h) == 0
    got = []
    bam._device_thread = threading.Thread(target=lambda: None)
    bam._device_thread.start()
    w = threading.Thread(target=lambda: got.append(bam.join_decoders()))
    w.start()
    w.join(0.2)
    assert w.is_alive()                        # (nobody has decided yet)
    native.lib().spl_bam_cancel(bam._h)
    native.lib().spl_bam_start(bam._h)
    w.join(10)
    assert not w.is_alive() and got == [False]
    bam.close()


def test_where_process_decodes(built, tmp_path, monkeypatch):
    """open_and_decode: on the GPU -- one decoder with one device, a share of the file per device with several -- unless told
    to use the host's threads."""
    from spliser_amd import process
    names, sets = _random_sets(8, 3_000, 2)
    path = str(tmp_path / "p.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    calls = []
    monkeypatch.setattr(native.BamFile, "decode_on_device_async", lambda self, device=0: calls.append(("device", device)))
    monkeypatch.setattr(native.BamFile, "decode_on_devices_async", lambda self, devices: calls.append(("shares", tuple(devices))))
    for devices, asked, want in (((0,), None, ("device", 0)), ((3,), None, ("device", 3)), ((0, 1), None, ("shares", (0, 1))),
                                 ((0, 0), True, ("shares", (0, 0))), ((0,), False, None), ((0, 1), False, None)):
        del calls[:]
        source = process.open_and_decode(path, devices, asked, 2)
        assert calls == ([want] if want else []), (devices, asked, calls)
        if want is None:       # (whoever was not handed to the fake device decoders decodes on the host)
            _same(source.reads(names[0]), sets[names[0]])
        source.close()


def test_share_plan_cuts_anywhere_and_covers_every_reference(built, tmp_path):
    """spl_bam_share_plan: as many stretches of the file as were asked for, in file order; the references a share can hold overlap
    with its neighbours' by the reference a cut falls into (and at most one more), and together they cover every reference."""
    import ctypes
    names, sets = _random_sets(9, 30_000, 5)
    path = str(tmp_path / "s.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    for want in (1, 2, 3, 8):
        bam = native.BamFile(path, defer=True)
        n = ctypes.c_int(0)
        assert native.lib().spl_bam_share_plan(bam._h, ctypes.c_int(want), ctypes.byref(n)) == 0
        assert n.value == want
        ranges = []
        for k in range(n.value):
            lo, hi = ctypes.c_int(0), ctypes.c_int(0)
            assert native.lib().spl_bam_share_range(bam._h, ctypes.c_int(k), ctypes.byref(lo), ctypes.byref(hi)) == 0
            assert lo.value < hi.value
            ranges.append((lo.value, hi.value))
        assert ranges[0][0] == 0 and ranges[-1][1] == len(names) + 1     # (the last share also takes the records without a reference)
        assert all(a[1] - 1 <= b[0] <= a[1] for a, b in zip(ranges, ranges[1:])) and all(a[0] <= b[0] for a, b in zip(ranges, ranges[1:]))
        assert sum(sum(bam.share_count_host(k)) for k in range(n.value)) == sum(sets[c].n for c in names)
        bam.close()


def test_close_does_not_wait_for_the_rest_of_the_file(built, tmp_path):
    """Closing a file whose decode has only begun (the caller failed elsewhere) stops the decoders at their next batch
    (spl_bam_cancel) instead of waiting for the whole file; a file nobody began to decode never is."""
    import time
    names, sets = _random_sets(3, 120_000, 2)
    path = str(tmp_path / "c.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=2, seq_mode=1)
    t = time.perf_counter()
    whole = native.BamFile(path, threads=1)
    whole.close()
    t_whole = time.perf_counter() - t
    t = time.perf_counter()
    bam = native.BamFile(path, threads=1, stream=True)
    bam.close()
    assert time.perf_counter() - t < max(0.05, 0.6 * t_whole)
    late = native.BamFile(path, defer=True)
    late.close()


def _plain_sets(sizes, seed=5):
    """Reference k with sizes[k] reads of 100M (vectorised: the share planner only cares where references begin in the file)."""
    rng = np.random.default_rng(seed)
    names = ["chr%d" % (k + 1) for k in range(len(sizes))]
    sets = {}
    for c, n in zip(names, sizes):
        pos = np.sort(rng.integers(1, 50_000_000, n)).astype(np.int32)
        sets[c] = samio.ReadSet(pos, rng.choice([0, 16], size=n).astype(np.uint16), np.arange(n + 1, dtype=np.uint32), np.full(n, (100 << 4), np.uint32))
    return names, sets


def _plan(bam, want):
    """-> [(tid_lo, tid_hi, file bytes, records per reference by the host's walk of the share)]"""
    import ctypes
    n = ctypes.c_int(0)
    assert native.lib().spl_bam_share_plan(bam._h, ctypes.c_int(want), ctypes.byref(n)) == 0
    shares = []
    for k in range(n.value):
        lo, hi = ctypes.c_int(0), ctypes.c_int(0)
        assert native.lib().spl_bam_share_range(bam._h, ctypes.c_int(k), ctypes.byref(lo), ctypes.byref(hi)) == 0
        shares.append((lo.value, hi.value, bam.share_info(k), bam.share_count_host(k)))
    return shares


HG38_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]


def _check_plan(shares, sizes, want, balance, unplaced=0):
    """Every record is in exactly one share (the host's walk of every share arrives at the next share's first record, and the
    shares' records per reference add up to the file's), the references a share holds are among those it says it can hold, the
    shares are in file order and equal in file bytes."""
    assert len(shares) == want
    assert shares[0][0] == 0 and shares[-1][1] == len(sizes) + 1
    assert all(a[2]["u_hi"] == b[2]["u_lo"] for a, b in zip(shares, shares[1:]))
    total = np.zeros(len(sizes) + 1, np.int64)
    for lo, hi, info, per in shares:
        per = np.array(per)
        total += per
        held = np.nonzero(per)[0]
        assert len(held) and held.min() >= lo and held.max() < hi
        assert hi - lo <= (held.max() - held.min() + 1) + 1      # (a superset by one reference at most)
    assert total[:-1].tolist() == list(sizes) and total[-1] == unplaced
    size = [info["file_bytes"] for _, _, info, _ in shares]
    assert max(size) <= balance * sum(size) / len(size), size


def test_share_plan_for_eight_devices_on_a_human_shaped_file(built, tmp_path):
    """The target machine has eight GPUs: a file of 24 references sized like hg38's chromosomes is cut into EIGHT stretches of
    equal size in file bytes -- anywhere, not at reference boundaries (whole-reference shares were 6-18 % of the file,
    max / mean 1.40): max / mean <= 1.05, and two, five and sixteen as well."""
    sizes = [m * 120 for m in HG38_MB]
    names, sets = _plain_sets(sizes)
    path = str(tmp_path / "h.bam")
    native.write_bam(path, names, [3 * 10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=3, seq_mode=1)
    for want in (8, 2, 5, 16):
        bam = native.BamFile(path, defer=True)
        _check_plan(_plan(bam, want), sizes, want, 1.05)
        bam.close()


def test_share_plan_on_a_five_chromosome_genome_and_odd_files(built, tmp_path):
    """An A. thaliana-shaped file (five references) gives EIGHT shares (whole-reference shares: five at most); so does a file with
    one reference that holds 40 % of the reads and one without any; a file of a single reference; and a file too small to be cut
    stays whole."""
    cases = {"five": [30400, 19700, 23500, 18600, 27000], "one": [90000], "skew": [3000] * 24}
    cases["skew"][2], cases["skew"][5] = 48000, 0
    for seed, (name, sizes) in enumerate(sorted(cases.items())):
        names, sets = _plain_sets(sizes, seed=6 + seed)
        path = str(tmp_path / (name + ".bam"))
        native.write_bam(path, names, [3 * 10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=3, seq_mode=1)
        bam = native.BamFile(path, defer=True)
        _check_plan(_plan(bam, 8), sizes, 8, 1.1)
        bam.close()
    names, sets = _plain_sets([40, 30], seed=9)
    path = str(tmp_path / "tiny.bam")
    native.write_bam(path, names, [3 * 10 ** 8] * len(names), [sets[c] for c in names], level=1, threads=1, seq_mode=1)
    bam = native.BamFile(path, defer=True)
    shares = _plan(bam, 8)
    assert len(shares) == 1 and shares[0][0] == 0 and shares[0][1] == 3 and shares[0][3][:2] == [40, 30]
    bam.close()
    sizes = cases["skew"]
    names, sets = _plain_sets(sizes, seed=8)
    path = str(tmp_path / "skew.bam")
    # the host decoder agrees with what was written (the file itself is sound)
    whole = native.BamFile(path, threads=2)
    for c in names:
        got = whole.reads(c)
        assert (got.n if got is not None else 0) == sets[c].n
    whole.close()


def edge_file(path, seed, n=900, unplaced=700):
    """A file for a share's edges to fall on: records longer than a BGZF block (2 % of them, 70-260 kb of SEQ and QUAL), a
    reference without reads between two that have them, records without a reference after the last one."""
    rng = np.random.default_rng(seed)
    names = ["c0", "gap", "c1", "c2"]
    sets = {}
    for c in ("c0", "c1", "c2"):
        pos = np.sort(rng.integers(1, 10 ** 7, n)).astype(np.int32)
        long_one = rng.random(n) < 0.02
        ops, off = [], [0]
        for i in range(n):
            if long_one[i]:
                ops += [(int(rng.integers(70_000, 260_000)) << 4) | 0]
            else:
                ops += [(int(rng.integers(20, 90)) << 4) | 0, (int(rng.integers(60, 5000)) << 4) | 3, (int(rng.integers(20, 90)) << 4) | 0]
            off.append(len(ops))
        sets[c] = samio.ReadSet(pos, rng.choice([0, 16, 99, 147], size=n).astype(np.uint16), np.array(off, np.uint32), np.array(ops, np.uint32))
    sets["gap"] = samio.ReadSet(np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
    samio.write_bam(path, names, [10 ** 8] * 4, [(c, sets[c]) for c in ("c0", "c1", "c2")], with_seq=True, unplaced=unplaced, level=1)
    return names, sets


@pytest.mark.parametrize("want", [2, 3, 5, 8])
def test_share_plan_over_long_records_and_an_unplaced_tail(built, tmp_path, want):
    """Cuts that fall inside records of several blocks, next to a reference without reads and among the records without a
    reference: the plan still gives every record to exactly one share."""
    path = str(tmp_path / "edges.bam")
    names, sets = edge_file(path, 5 + want)
    bam = native.BamFile(path, defer=True)
    _check_plan(_plan(bam, want), [sets[c].n for c in names], want, 1.6, unplaced=700)
    bam.close()


@pytest.mark.parametrize("seq_mode,level", [(1, 1), (2, 6), (0, 1)])
def test_the_hosts_sample_of_a_file_says_what_its_records_take(built, tmp_path, seq_mode, level):
    """``spl_bam_sample`` (what the device decoder sizes its arrays by before it has scanned a window): records and CIGAR ops per
    inflated byte from three places of the file, within a fifth of the file's own -- and the long-record file's, whose sample
    falls between records of 200 kb."""
    names, sets = _random_sets(33, 60_000, 3)
    path = str(tmp_path / "s.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=level, threads=2, seq_mode=seq_mode)
    bam = native.BamFile(path, defer=True)
    rec, ops, nbytes = bam.sample()
    bam.close()
    whole = native.BamFile(path, threads=2)
    n_all = sum(whole.reads(c).n for c in names)
    ops_all = sum(int(whole.reads(c).cig_off[-1]) for c in names)
    whole.close()
    inflated = 0
    import gzip
    with gzip.open(path, "rb") as fh:
        while True:
            chunk = fh.read(1 << 24)
            if not chunk:
                break
            inflated += len(chunk)
    assert rec > 100 and abs(rec / nbytes - n_all / inflated) < 0.2 * n_all / inflated
    assert abs(ops / nbytes - ops_all / inflated) < 0.25 * ops_all / inflated
    path2 = str(tmp_path / "edges.bam")
    names2, sets2 = edge_file(path2, 9)
    bam = native.BamFile(path2, defer=True)
    rec, ops, nbytes = bam.sample()
    bam.close()
    assert rec > 0 and nbytes > 0 and ops >= rec


def test_htslib_shaped_writer(built, tmp_path):
    """seq_mode 2 (what the bench's `htslib` legs decode): whole records per BGZF block, as htslib's bam_write1 cuts them; every
    record well-formed -- Illumina-style name, reg2bin as the SAM specification computes it, mate fields only for paired flags,
    NH / HI / AS / nM tags (XS:A on spliced reads) -- and the reads that went in come out of the host decoder."""
    import struct
    import zlib
    names, sets = _random_sets(23, 4000, 2)
    path = str(tmp_path / "h.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=6, threads=2, seq_mode=2)
    raw = open(path, "rb").read()
    at, blocks = 0, []
    while at < len(raw):
        bsize = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        blocks.append(zlib.decompress(raw[at + 18:at + bsize - 8], -15))
        at += bsize
    assert blocks[-1] == b""                                   # the EOF marker
    head = blocks[0]
    l_text = struct.unpack_from("<i", head, 4)[0]
    assert head[:4] == b"BAM\x01" and struct.unpack_from("<i", head, 8 + l_text)[0] == len(names)
    n_rec = n_spliced = 0
    for blk in blocks[1:-1]:                                   # (the header has a block of its own, as after htslib's flush)
        p = 0
        while p < len(blk):
            bs = struct.unpack_from("<i", blk, p)[0]
            assert p + 4 + bs <= len(blk), "a record is cut by a block's end"
            tid, pos0, l_name, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", blk, p + 4)
            name = blk[p + 36:p + 36 + l_name]
            assert name.endswith(b"\x00") and name.count(b":") == 6
            cig = struct.unpack_from("<%dI" % n_cig, blk, p + 36 + l_name)
            rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
            qlen = sum(c >> 4 for c in cig if (c & 15) in (0, 1, 4, 7, 8))
            assert l_seq == qlen and bin_ == samio._reg2bin(pos0, pos0 + max(rlen, 1))
            assert (mtid, mpos, tlen) == (-1, -1, 0) if not flag & 1 else (mtid == tid and mpos >= 0)
            tags = blk[p + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq:p + 4 + bs]
            assert tags[:3] == b"NHC" and b"HIC" in tags and b"nMC" in tags and (b"ASC" in tags or b"ASS" in tags)
            spliced = any((c & 15) == 3 for c in cig)
            assert (b"XSA" in tags) == spliced
            n_spliced += spliced
            n_rec += 1
            p += 4 + bs
    assert n_rec == sum(s.n for s in sets.values()) and n_spliced > 0
    bam = native.BamFile(path, threads=2)
    for c in names:
        _same(bam.reads(c), sets[c])
    bam.close()
