"""The host packer (spliser_amd/csrc/spl_pack.cpp through ``spl_pack_host``) against a plain-Python statement of the layout
rules of spl_pack.h: every read must come back from its record with POS, flag and the ops checkBam's walk depends on
(SpliSER_v0_1_8.py:457-464: ops that do not consume the reference change nothing), in its chunk, in its run, in file order.
No GPU involved."""
import numpy as np
import pytest

from spliser_amd import native, samio, synth
from tests import randcase

COORD_MAX = 2147483581
CHUNK = 2048
REC = (8, 16, 24, 24)


def kind(op):
    return {0: 1, 7: 1, 8: 1, 3: 2, 2: 3}.get(int(op) & 15, 0)


def as_m(ops):
    """M, = and X are the same thing to checkBam ("mappedRegion", :458-460): the length-only records give back M."""
    return [(o >> 4) << 4 if kind(o) == 1 else o for o in ops]


def classify(pos, flag, ops):
    """-> (run, ops the record must give back)"""
    ops = [int(o) for o in ops]
    n_all = len(ops)
    cons = [o for o in ops if kind(o)] if n_all <= 8 else None
    placed = not (flag & 4) and pos >= 0
    room = COORD_MAX - pos
    if cons is not None and len(cons) <= 3:
        if placed and len(cons) == 1 and kind(cons[0]) == 1 and (cons[0] >> 4) < 65536 and (cons[0] >> 4) <= room:
            return 0, as_m(cons)
        if (placed and len(cons) == 3 and [kind(o) for o in cons] == [1, 2, 1] and (cons[0] >> 4) < 65536
                and sum(o >> 4 for o in cons) <= room):
            return 1, as_m(cons)
        return 3, cons
    if (placed and cons is not None and len(cons) == 5 and [kind(o) for o in cons] == [1, 2, 1, 2, 1]
            and all((cons[k] >> 4) < 65536 for k in (0, 2, 4)) and sum(o >> 4 for o in cons) <= room):
        return 2, as_m(cons)
    return 3, ops


def unpack(desc, rec, wide):
    """-> per chunk, per run: list of (pos, flag, ops)"""
    words = rec.view(np.uint8)
    out = []
    for d in desc:
        n = [int(v) for v in d["n"]]
        off = [0, (n[0] * 8 + 15) & ~15]
        off.append(off[1] + n[1] * 16)
        off.append(off[2] + n[2] * 24)
        base = int(d["rec_off"])
        runs = []
        for r in range(4):
            lst = []
            for i in range(n[r]):
                a = base + off[r] + i * REC[r]
                w = words[a:a + REC[r]].view(np.uint32)
                pos, flag = int(w[0].view(np.int32) if hasattr(w[0], "view") else w[0]), int(w[1]) & 0xffff
                pos = int(np.int32(np.uint32(w[0])))
                if r == 0:
                    ops = [(int(w[1]) >> 16) << 4]
                elif r == 1:
                    ops = [(int(w[1]) >> 16) << 4, (int(w[2]) << 4) | 3, int(w[3]) << 4]
                elif r == 2:
                    ops = [(int(w[1]) >> 16) << 4, (int(w[2]) << 4) | 3, (int(w[3]) & 0xffff) << 4, (int(w[4]) << 4) | 3,
                           (int(w[3]) >> 16) << 4]
                else:
                    sub, n_ops = int(w[1]) >> 29, int(w[5])
                    assert (int(w[1]) >> 16) & 0x1fff == min(n_ops, 0x1fff)
                    if sub == 4:
                        ops = [int(v) for v in wide[int(w[4]):int(w[4]) + n_ops]]
                        assert ops[:2] == [int(w[2]), int(w[3])]
                    else:
                        assert sub == 3 and n_ops <= 3
                        ops = [int(v) for v in w[2:2 + n_ops]]
                        assert all(int(v) == 0xf for v in w[2 + n_ops:5])
                lst.append((pos, flag, ops))
            runs.append(lst)
        out.append(runs)
    return out


def check(reads, threads=3):
    ra = native.ReadArrays(reads.pos, reads.flag, reads.cig_off, reads.cigar)
    desc, rec, wide = native.pack_host(ra, threads=threads)
    assert len(desc) == (reads.n + CHUNK - 1) // CHUNK
    got = unpack(desc, rec, wide)
    weight = (2, 5, 9)
    for c, runs in enumerate(got):
        want = [[], [], [], []]
        cost = 0
        lo, hi = c * CHUNK, min(reads.n, (c + 1) * CHUNK)
        for i in range(lo, hi):
            ops = reads.cigar[int(reads.cig_off[i]):int(reads.cig_off[i + 1])]
            run, keep = classify(int(reads.pos[i]), int(reads.flag[i]), ops)
            want[run].append((int(reads.pos[i]), int(reads.flag[i]), keep))
            cost += weight[run] if run < 3 else (14 if len(keep) > 3 else 6)
        assert runs == want, "chunk %d" % c
        assert int(desc[c]["first_pos"]) == int(reads.pos[lo])
        assert int(desc[c]["cost"]) == cost
    # the same bytes whatever the number of threads
    d1, r1, w1 = native.pack_host(ra, threads=1)
    assert np.array_equal(d1, desc) and np.array_equal(r1, rec) and np.array_equal(w1, wide)
    return desc


def test_pack_corner_cigars():
    recs = [(0, 100, "50M"), (16, 100, "50="), (0, 100, "20M100N30M"), (0, 100, "5S20M100N30M3S"), (0, 100, "20M100N30M50N40M"),
            (0, 100, "2H20M2I100N30M50N40M1P"), (4, 100, "50M"), (0, 100, "*"), (0, 100, "10S"), (0, 100, "10M5D10M"),
            (0, 100, "10M0N10M"), (0, 100, "10M5N5N10M"), (0, 100, "10N20M"), (0, 100, "20M10N"), (0, 100, "70000M"),
            (0, 100, "70000M10N5M"), (0, 100, "5M10N70000M"), (0, 100, "5M10N5M10N70000M"), (0, 100, "1M1N1M1N1M1N1M"),
            (0, 100, "1M1I1M1I1M1I1M1I1M"), (0, 100, "1M1D" * 10), (1024, 2147483000, "1000M"), (0, 2147483000, "100M")]
    reads = samio.ReadSet.from_records(recs)
    desc = check(reads, threads=1)
    assert desc[0]["n"].tolist() == [3, 4, 2, len(recs) - 9]


def test_pack_negative_and_empty():
    empty = samio.ReadSet.empty()
    d, r, w = native.pack_host(native.ReadArrays(empty.pos, empty.flag, empty.cig_off, empty.cigar))
    assert len(d) == 0 and len(r) == 0 and len(w) == 0
    neg = samio.ReadSet(np.array([-5, 7], np.int32), [0, 0], [0, 1, 2], [50 << 4, 50 << 4])
    desc = check(neg)
    assert desc[0]["n"].tolist() == [1, 0, 0, 1]


@pytest.mark.parametrize("seed", range(6))
def test_pack_random_cases(seed):
    _, reads = randcase.make_case(seed, stranded=bool(seed & 1))
    check(reads)


def test_pack_many_chunks():
    wl = synth.Workload("single_gene", n_reads=9000, seed=3, workers=1)
    reads = wl.reads[0]
    assert reads.n > 4 * CHUNK
    long_read = samio.ReadSet.from_records([(0, int(reads.pos[-1]), "1M1D" * 5000)])
    both = samio.ReadSet(np.concatenate((reads.pos, long_read.pos)), np.concatenate((reads.flag, long_read.flag)),
                         np.concatenate((reads.cig_off, long_read.cig_off[1:] + reads.cig_off[-1])),
                         np.concatenate((reads.cigar, long_read.cigar)))
    desc = check(both, threads=4)
    assert int(desc["n"][:, 1].sum()) > 0 and int(desc["n"][:, 0].sum()) > 0


def test_the_layout_kernels_classifier_is_the_host_packers():
    """spl_pack.h has the read classifier twice: classify_ops (the host packer's, checked above against the Python statement of the
    layout rules) and classify_lean + classify_fast5 (the layout kernel's, spl_devpack.hip: the same decision from a bit mask of the
    consuming ops, and straight-line from a signature of the first five ops' kinds for CIGARs without clips or insertions).
    Built for the host (tests/hostsim/pack_classify_host.cpp) and run side by side on CIGARs of every shape: 0 ... 12 ops of every
    code including the undefined ones, lengths around 65535 / 65536 and 2^28 - 1, POS negative and at the end of the coordinate
    space, flag 0x4 -- every field the layout uses must agree."""
    import ctypes
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
    so, src = os.path.join(here, "libpack_classify_host.so"), os.path.join(here, "pack_classify_host.cpp")
    hdr = os.path.join(here, "..", "..", "spliser_amd", "csrc", "spl_pack.h")
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in (src, hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.classify_compare.restype = ctypes.c_int64
    rng = np.random.default_rng(77)
    n = 400_000
    n_ops = rng.choice([0, 1, 1, 1, 2, 3, 3, 3, 4, 5, 5, 5, 6, 7, 8, 8, 9, 12], n)
    cig_off = np.concatenate(([0], np.cumsum(n_ops))).astype(np.uint32)
    total = int(cig_off[-1])
    # op codes: mostly M / N with clips and indels between, every code 0..15 now and then
    codes = rng.choice([0, 0, 0, 0, 3, 3, 3, 1, 2, 4, 5, 7, 8, 6, 9, 15], total).astype(np.uint32)
    regular = rng.random(n) < 0.5           # half of the reads: M (N M)* with optional clips, the shapes the runs exist for
    for i in np.nonzero(regular)[0][:150_000]:
        a, b = int(cig_off[i]), int(cig_off[i + 1])
        k = b - a
        if k == 0:
            continue
        pat = [0, 3] * k
        seq = pat[:k] if k % 2 else [4] + pat[:k - 1]
        if k >= 3 and rng.random() < 0.3:
            seq[-1] = 4 if seq[-1] == 3 else seq[-1]
        codes[a:b] = seq
    lens = rng.choice([1, 2, 50, 100, 150, 65535, 65536, 70000, (1 << 28) - 1], total,
                      p=[0.1, 0.1, 0.2, 0.2, 0.25, 0.05, 0.05, 0.03, 0.02]).astype(np.uint32)
    cigar = np.concatenate((((lens << 4) | codes).astype(np.uint32), rng.integers(0, 2 ** 32, 8, dtype=np.uint64).astype(np.uint32)))   # (+ 8 words of anything behind the last read)
    pos = rng.choice([1, 100, 5_000_000, COORD_MAX - 100, COORD_MAX, -1, -5, 2147483647], n,
                     p=[0.1, 0.3, 0.4, 0.05, 0.05, 0.04, 0.03, 0.03]).astype(np.int32)
    flag = rng.choice([0, 16, 99, 147, 4, 20, 256, 2048, 65535], n).astype(np.uint16)
    first = ctypes.c_int64(-1)

    def p(a):
        return a.ctypes.data_as(ctypes.c_void_p)
    bad = lib.classify_compare(ctypes.c_int64(n), p(pos), p(flag), p(cig_off), p(cigar), ctypes.byref(first))
    i = first.value
    assert bad == 0, "classify_lean differs from classify_ops on %d reads, first: pos %d flag %d ops %s" % (
        bad, pos[i], flag[i], [(int(o) >> 4, int(o) & 15) for o in cigar[cig_off[i]:cig_off[i + 1]]])
