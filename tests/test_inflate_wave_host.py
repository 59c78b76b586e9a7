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
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-Wno-unknown-pragmas", "-shared", "-fPIC", "-DEMUL_NO_MAIN", "-o", LIB, SRC, "-lz"])
    lib = ctypes.CDLL(LIB)
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
