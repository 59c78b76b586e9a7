"""The body of the wave-per-block inflate kernel (spliser_amd/csrc/spl_inflate_wave.h) on the HOST: the same source, compiled
against tests/hostsim/wave_emul.h (a wave = 64 fibers, every cross-lane primitive a checked rendezvous), on the DEFLATE streams
tests/test_gpu_inflate_kernel.py runs on the GPU -- fixed, stored and dynamic blocks, 15-bit codes, matches at every small
distance and at the largest, several blocks per stream -- each against the bytes that went in, and on damaged streams: an error
code, nothing written past the block.  There is no GPU where this is built; this is where the kernel's logic is checked first."""
import ctypes
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostsim", "inflate_wave_host.cpp")
LIB = os.path.join(ROOT, "tests", "hostsim", "libinflate_wave_host.so")
DEPS = [SRC, os.path.join(ROOT, "tests", "hostsim", "wave_emul.h"), os.path.join(ROOT, "spliser_amd", "csrc", "spl_inflate_wave.h"),
        os.path.join(ROOT, "spliser_amd", "csrc", "spl_inflate.h")]


@pytest.fixture(scope="module")
def emul():
    extra = os.environ.get("SPL_EMUL_DEFINES", "").split()      # (kernel experiments: e.g. "-DSPLZ_RING=64 -DSPLZ_FIFO=128", the copying kernel's LDS per lane)
    lib_path = LIB if not extra else LIB.replace(".so", "_exp.so")
    if extra or not os.path.exists(lib_path) or any(os.path.getmtime(d) > os.path.getmtime(lib_path) for d in DEPS):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-Wno-unknown-pragmas", "-shared", "-fPIC", "-DEMUL_NO_MAIN"] + extra + ["-o", lib_path, SRC, "-lz"])
    lib = ctypes.CDLL(lib_path)
    lib.emul_inflate_blocks.restype = ctypes.c_int
    return lib


def _run(lib, streams):
    """streams: [(name, data, comp)] -> (status[], output bytes with 128 bytes of 0xA5 behind the last block)"""
    image = bytearray()
    blocks = np.zeros((len(streams), 4), np.uint64)  # spl_zblock: in, out, (in_len | out_len << 32), (crc | pad << 32)
    out_at = 0
    for k, (name, data, comp) in enumerate(streams):
        blocks[k] = (len(image), out_at, len(comp) | (len(data) << 32), zlib.crc32(data) & 0xffffffff)
        image += comp
        out_at += len(data)
    image += bytes(64)
    img = np.frombuffer(bytes(image), np.uint8).copy()
    out = np.full(out_at + 128, 0xA5, np.uint8)
    status = np.full(len(streams), 0xffffffff, np.uint32)
    rc = lib.emul_inflate_blocks(img.ctypes.data_as(ctypes.c_void_p), blocks.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(streams)),
                                 out.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, "the emulated wave broke a rule of spl_wave.h at block %d (see stderr)" % (-1 - rc)
    return status, out.tobytes(), blocks, image


def _streams(every):
    from tests.test_gpu_inflate_kernel import _streams as all_streams
    made = all_streams()
    for k, (name, data, comp) in enumerate(made):
        if name == "level0":
            made[k] = (name, zlib.decompress(comp, -15), comp)
    return [s for k, s in enumerate(made) if k % every == 0 or "flushes" in s[0] or s[0] in ("one", "level0")]


def test_streams_against_zlib(emul):
    streams = _streams(int(os.environ.get("SPL_EMUL_EVERY", "4")))
    status, got, _, _ = _run(emul, streams)
    at = 0
    for k, (name, data, comp) in enumerate(streams):
        assert status[k] == 0, (name, k, int(status[k]))
        assert got[at:at + len(data)] == data, (name, k)
        at += len(data)
    assert got[at:at + 128] == b"\xa5" * 128


def test_damaged_streams_end_with_an_error_code(emul):
    streams = _streams(9)[:14]
    _, _, blocks, image = _run(emul, streams[:1])
    rng = np.random.default_rng(3)
    bad_streams = []
    for name, data, comp in streams:
        c = bytearray(comp)
        c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        bad_streams.append((name, data, bytes(c)))
        bad_streams.append((name + "/cut", data, comp[:max(1, len(comp) // 2)]))     # the block's data ends early
    status, got, _, _ = _run(emul, bad_streams)
    at = 0
    for k, (name, data, comp) in enumerate(bad_streams):
        ok = status[k] == 0
        if ok:  # (a flipped bit can leave a valid stream of the same length: then the bytes differ, which the CRC is there for)
            assert len(got[at:at + len(data)]) == len(data)
        else:
            assert 1 <= status[k] <= 8, (name, int(status[k]))
        at += len(data)
    assert got[at:at + 128] == b"\xa5" * 128
    assert sum(1 for s in status if s != 0) >= len(bad_streams) // 2


def test_bgzf_blocks_of_a_bam(emul, tmp_path):
    from spliser_amd import native
    from tests.test_bam_decode import _random_sets
    native.build()
    names, sets = _random_sets(77, 3000, 2)
    path = str(tmp_path / "x.bam")
    native.write_bam(path, names, [10 ** 8] * len(names), [sets[c] for c in names], level=6, threads=2, seq_mode=1)
    raw = open(path, "rb").read()
    streams, at = [], 0
    while at < len(raw):
        bsize = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        comp = raw[at + 18:at + bsize - 8]
        streams.append(("bgzf", zlib.decompress(comp, -15), comp))
        at += bsize
    status, got, _, _ = _run(emul, streams)
    assert not status.any()
    assert got[:-128] == b"".join(s[1] for s in streams)


class _Bits:
    """DEFLATE's bit order: fields LSB first, Huffman codes MSB first (RFC 1951 3.1.1)."""

    def __init__(self):
        self.buf, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, bits):
        self.acc |= value << self.n
        self.n += bits
        while self.n >= 8:
            self.buf.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def code(self, value, bits):
        self.put(int(format(value, "0%db" % bits)[::-1], 2), bits)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def _mixed_block(n_pairs, n_stored, seed=5):
    """One DEFLATE stream: a fixed-code section of `n_pairs` x (lone literal, match of length 3 at distance 1) -- 5 bytes of
    tokens for 4 of output, the most a Huffman section can ask for -- then `n_stored` stored sections of ONE byte each (2 bytes of
    tokens per byte of output).  -> (data, comp)"""
    rng = np.random.default_rng(seed)
    w, data = _Bits(), bytearray()
    w.put(0, 1), w.put(1, 2)                        # not the last section, fixed code
    for lit in rng.integers(0, 144, n_pairs):
        w.code(0x30 + int(lit), 8)                  # literal 0..143: 8 bits, 00110000 + value
        w.code(1, 7)                                # 257 = length 3: 7 bits, 0000001
        w.code(0, 5)                                # distance code 0 = 1
        data += bytes([int(lit)]) * 4
    w.code(0, 7)                                    # 256, the section's end
    for k, b in enumerate(rng.integers(0, 256, n_stored)):
        w.put(1 if k == n_stored - 1 else 0, 1), w.put(0, 2)
        w.align()
        w.put(1, 16), w.put(0xfffe, 16), w.put(int(b), 8)
        data.append(int(b))
    w.align()
    comp = bytes(w.buf)
    assert zlib.decompress(comp, -15) == bytes(data)
    return bytes(data), comp


def test_stored_sections_cannot_overrun_a_blocks_token_room(emul):
    """the stored path added its tokens without looking at the room.  15 250 pairs are 76 250 bytes of tokens;
    4 536 one-byte stored sections behind them want 9 072 more, and a block's room is 82 048: the block has to be REFUSED (the
    host decoder takes it then), not written over its neighbour -- the emulator's harness checks the bytes behind the room."""
    over = _mixed_block(15250, 4536)
    assert len(over[0]) == 65536 and len(over[1]) <= 65536
    fits = _mixed_block(15250, 2800)
    status, got, _, _ = _run(emul, [("fits", *fits), ("over", *over), ("fits again", *fits)])
    assert list(status) == [0, 9, 0]                # SPL_Z_TOKENS (more tokens than any block's room holds) for the middle one, its neighbours untouched
    assert got[:len(fits[0])] == fits[0]
    at = len(fits[0]) + len(over[0])
    assert got[at:at + len(fits[0])] == fits[0]


def test_literal_runs_longer_than_128_in_one_lane(emul):
    """Huffman-only streams of bytes that are nearly all the same: the common byte's code is ONE bit, a lane's 256 bits hold two
    hundred literals and more, so a run's length byte (127 = 128 literals at most) has to be counted and written more than once per
    lane -- count_from tallies them where a run ends, the writing pass where one begins."""
    rng = np.random.default_rng(11)
    streams = []
    for k, p in enumerate((0.01, 0.03, 0.10)):
        data = np.where(rng.random(60000) < p, rng.integers(1, 256, 60000), 0).astype(np.uint8).tobytes()
        comp = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        streams.append(("skewed%d" % k, data, comp.compress(data) + comp.flush()))
    status, got, _, _ = _run(emul, streams)
    at = 0
    for k, (name, data, comp) in enumerate(streams):
        assert status[k] == 0, (name, int(status[k]))
        assert got[at:at + len(data)] == data, name
        at += len(data)
