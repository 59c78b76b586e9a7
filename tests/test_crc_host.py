"""The body of the CRC32 kernel (spliser_amd/csrc/spl_crc.h: a block by one lane as S streams whose registers are put together with
GF(2) arithmetic) built for the host and held against zlib.  BGZF trailer: RFC 1952 §8 / SAM specification §4.1; the reference reads
its BAM through `samtools view` (SpliSER_v0_1_8.py:422), which verifies this checksum for every block."""
import ctypes
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(HERE, "hostsim", "libcrc_host.so")
    srcs = [os.path.join(HERE, "hostsim", "crc_host.cpp"), os.path.join(HERE, "..", "spliser_amd", "csrc", "spl_crc.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    L = ctypes.CDLL(so)
    L.crc_block.restype = ctypes.c_uint32
    L.crc_x2n.restype = ctypes.c_uint32
    L.crc_mulmod.restype = ctypes.c_uint32
    return L


def test_powers_of_x_are_what_zlib_combines_with(lib):
    # crc32_combine(crc(A), crc(B), len(B)) is crc(A + B): with B = m zero bytes that is the register times x^(8m)
    assert lib.crc_x2n(0) == 0x40000000 and lib.crc_x2n(1) == 0x20000000 and lib.crc_x2n(3) == 0x00800000   # x, x^2, x^8
    assert lib.crc_mulmod(0x80000000, 0x12345678) == 0x12345678                                              # x^0 is the one
    for k in range(5, 24):
        assert lib.crc_x2n(k) == lib.crc_mulmod(lib.crc_x2n(k - 1), lib.crc_x2n(k - 1))


@pytest.mark.parametrize("streams", [1, 2, 4, 8])
def test_every_length_around_the_cuts(lib, streams):
    rng = np.random.default_rng(7 + streams)
    data = rng.integers(0, 256, 70000, dtype=np.uint8)
    raw = data.tobytes()
    lengths = list(range(0, 600)) + [1023, 1024, 1025, 4095, 4096, 4111, 0xff00 - 1, 0xff00, 0xff00 + 1, 65535, 65536] \
        + [int(x) for x in rng.integers(600, 65536, 200)]
    for n in lengths:
        for start in (0, 1, 3):                      # (payloads start wherever the block before ended: no alignment)
            if start + n > len(raw):
                continue
            got = lib.crc_block(data[start:].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n), streams)
            assert got == zlib.crc32(raw[start:start + n]) & 0xffffffff, (streams, n, start)


@pytest.mark.parametrize("streams", [4, 8])
def test_constant_and_sparse_payloads(lib, streams):
    for fill in (0x00, 0xff, 0x41):
        for n in (0, 15, 64, 128, 129, 65280, 65536):
            buf = np.full(max(n, 1), fill, np.uint8)
            assert lib.crc_block(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n), streams) == zlib.crc32(buf[:n].tobytes()) & 0xffffffff
    buf = np.zeros(65280, np.uint8)
    for at in (0, 16319, 16320, 32640, 65279):       # one byte set: in the first part, at the cuts, at the very end
        buf[:] = 0
        buf[at] = 0x80
        assert lib.crc_block(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(buf)), streams) == zlib.crc32(buf.tobytes()) & 0xffffffff


@pytest.fixture(scope="module")
def wave_lib():
    so = os.path.join(HERE, "hostsim", "libcrc_wave_host.so")
    csrc = os.path.join(HERE, "..", "spliser_amd", "csrc")
    srcs = [os.path.join(HERE, "hostsim", "crc_wave_host.cpp"), os.path.join(HERE, "hostsim", "wave_emul.h"), os.path.join(csrc, "spl_crc_wave.h"),
            os.path.join(csrc, "spl_crc.h"), os.path.join(csrc, "spl_wave.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    return ctypes.CDLL(so)


def _wave_crcs(wave_lib, data, spans):
    off = np.array([a for a, _ in spans], np.uint64)
    ln = np.array([n for _, n in spans], np.uint32)
    out = np.zeros(len(spans), np.uint32)
    rc = wave_lib.crc_wave_blocks(data.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p), ln.ctypes.data_as(ctypes.c_void_p),
                                  ctypes.c_uint32(len(spans)), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, "the wave broke a rule of the emulator (%d)" % rc
    return out


def test_a_block_by_one_wave_against_zlib(wave_lib):
    """spl_crc_wave.h (the body of spl_crc32_wave_kernel, round 6): a wave walks the block in rows of 1024 bytes laid against the
    block's END, lane i the bytes [16 i, 16 i + 16) of every row, twenty independent table look-ups a row, the lanes' registers put
    together with a constant factor each.  Under the wave emulator against zlib: every length around a row, around the first row's
    masks (what lies in front of the block counts as zeros, the first four bytes are complemented), any alignment."""
    rng = np.random.default_rng(17)
    data = rng.integers(0, 256, 140000, dtype=np.uint8)
    raw = data.tobytes()
    lengths = list(range(0, 80)) + list(range(1000, 1060)) + list(range(2030, 2070)) + [4095, 4096, 4111, 0xff00 - 1, 0xff00, 0xff00 + 1, 65535, 65536] \
        + [int(x) for x in rng.integers(80, 65536, 60)]
    spans = [(start, n) for n in lengths for start in (0, 1, 3, 16, 4097) if start + n <= len(raw)]
    got = _wave_crcs(wave_lib, data, spans)
    for (start, n), g in zip(spans, got.tolist()):
        assert g == zlib.crc32(raw[start:start + n]) & 0xffffffff, (start, n)
    # blocks one behind the other, as in the stream (a wave must not depend on what its neighbours hold), and sparse payloads
    at, spans = 0, []
    for n in [65280, 1, 0, 17, 1024, 1023, 1025, 30000]:
        spans.append((at, n))
        at += n
    got = _wave_crcs(wave_lib, data, spans)
    assert got.tolist() == [zlib.crc32(raw[a:a + n]) & 0xffffffff for a, n in spans]
    for fill in (0x00, 0xff):
        buf = np.full(70000, fill, np.uint8)
        spans = [(5, n) for n in (0, 3, 4, 15, 16, 64, 1024, 65280, 65536)]
        assert _wave_crcs(wave_lib, buf, spans).tolist() == [zlib.crc32(buf[5:5 + n].tobytes()) & 0xffffffff for _, n in spans]
    buf = np.zeros(65280, np.uint8)
    for at in (0, 3, 4, 15, 16, 1023, 1024, 32640, 65279):
        buf[:] = 0
        buf[at] = 0x80
        assert _wave_crcs(wave_lib, buf, [(0, len(buf))]).tolist() == [zlib.crc32(buf.tobytes()) & 0xffffffff]
