"""The inflate and CRC32 kernels (spl_inflate.hip) on raw DEFLATE streams made by zlib, without a BAM around them: payloads
chosen for what BAM records never force -- 15-bit codes, matches at the largest distance and length, every strategy and level,
streams of several blocks with empty stored blocks between them, payloads of one byte -- each against the bytes that went in."""
import ctypes
import zlib

import numpy as np
import pytest

from spliser_amd import native

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _payloads():
    rng = np.random.default_rng(5)
    out = {}
    # literal frequencies like Fibonacci numbers: the Huffman tree wants 20 levels, zlib cuts it to 15-bit codes
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    skew = np.concatenate([np.full(n, k, np.uint8) for k, n in enumerate(fib)])
    rng.shuffle(skew)
    out["skewed"] = skew.tobytes()[:65000]
    out["run"] = bytes([7]) * 65280                                       # distance 1, length 258, over and over
    chunk = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    out["far"] = (chunk + chunk)[:65280]                                   # matches at distance 32768
    out["random"] = rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()  # incompressible: stored blocks
    out["bytes"] = bytes(range(256)) * 200
    out["one"] = b"x"
    out["two_values"] = rng.choice(np.array([65, 67], np.uint8), 65280).tobytes()
    out["text"] = (b"read_%d\tchr1\t%d\t60\t76M\t=\t%d\tACGTTGCA\tFFFFFFFF\n" * 700 % tuple(range(2100)))[:65280]
    periods = b"".join(bytes(rng.integers(0, 256, p, dtype=np.uint8)) * (3000 // p) for p in (1, 2, 3, 4, 5, 6, 7, 8, 9, 31, 32, 33))
    out["periods"] = periods[:65280]                                       # copies at every small distance
    return out


def _streams():
    made = []
    for name, data in _payloads().items():
        for level in (1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                made.append((name, data, c.compress(data) + c.flush()))
        # several DEFLATE blocks in one stream, empty stored blocks (what a sync flush leaves) between them
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        parts, step = [], max(1, len(data) // 5)
        for k in range(0, len(data), step):
            parts.append(c.compress(data[k:k + step]))
            parts.append(c.flush(zlib.Z_SYNC_FLUSH if (k // step) % 2 else zlib.Z_FULL_FLUSH))
        parts.append(c.flush())
        made.append((name + "+flushes", data, b"".join(parts)))
    made.append(("level0", b"abc" * 20000, zlib.compressobj(0, zlib.DEFLATED, -15).compress(b"abc" * 20000) + b"\x03\x00"))
    return made


@pytest.mark.parametrize("dense", ["0", "1"])
def test_inflate_and_crc_kernels_against_zlib(monkeypatch, dense):
    """... with either decoding kernel: the one with 5 KB of token room a tile and four waves a SIMD, and the denser one (3 KB,
    five waves) that files of well-deflating blocks get (SPL_Z_DENSE forces one)."""
    monkeypatch.setenv("SPL_Z_DENSE", dense)
    tokens = True
    native.build()
    lib = native.lib()
    streams = _streams()
    for name, data, comp in streams:                      # (the streams are what zlib says they are)
        if name != "level0":
            assert zlib.decompress(comp, -15) == data
    image = bytearray()
    blocks = np.zeros((len(streams), 4), np.uint64)       # spl_zblock: in, out, (in_len | out_len << 32), (crc | pad << 32)
    out_at = 0
    for k, (name, data, comp) in enumerate(streams):
        if name == "level0":
            data = zlib.decompress(comp, -15)
            streams[k] = (name, data, comp)
        blocks[k] = (len(image), out_at, len(comp) | (len(data) << 32), zlib.crc32(data) & 0xffffffff)
        image += comp
        out_at += len(data)
    d_image = torch.zeros(len(image) + 64, dtype=torch.uint8, device="cuda:0")
    d_image[:len(image)] = torch.from_numpy(np.frombuffer(bytes(image), np.uint8).copy()).to("cuda:0")
    d_blocks = torch.from_numpy(blocks.view(np.int64)).to("cuda:0")
    d_out = torch.full((out_at + 128,), 0xA5, dtype=torch.uint8, device="cuda:0")
    d_status = torch.full((len(streams),), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    args = (ctypes.c_void_p(d_blocks.data_ptr()), ctypes.c_uint32(len(streams)))
    lib.spl_dev_inflate_work_bytes.restype = ctypes.c_size_t
    d_work = torch.zeros(lib.spl_dev_inflate_work_bytes(ctypes.c_uint32(len(streams))), dtype=torch.uint8, device="cuda:0")
    rc = lib.spl_dev_launch_inflate(ctypes.c_void_p(d_image.data_ptr()), *args, ctypes.c_void_p(d_out.data_ptr()),
                                    ctypes.c_void_p(d_status.data_ptr()), ctypes.c_void_p(d_work.data_ptr() if tokens else 0), ctypes.c_void_p(0))
    assert rc == 0
    rc = lib.spl_dev_launch_crc32(ctypes.c_void_p(d_out.data_ptr()), *args, ctypes.c_void_p(d_status.data_ptr()), ctypes.c_void_p(0))
    assert rc == 0
    torch.cuda.synchronize()
    status = d_status.cpu().numpy()
    got = d_out.cpu().numpy().tobytes()
    at = 0
    for k, (name, data, comp) in enumerate(streams):
        assert status[k] == 0, (name, k, int(status[k]))
        assert got[at:at + len(data)] == data, (name, k)
        at += len(data)
    assert got[at:at + 128] == b"\xa5" * 128              # nothing written behind the last block
    # damage: a flipped bit in the compressed bytes or a wrong CRC is an error code, never a hang or a write elsewhere
    bad = bytearray(image)
    for k in range(0, len(streams), 7):
        off = int(blocks[k][0]) + (int(blocks[k][2]) & 0xffffffff) // 2
        bad[off] ^= 0x10
    d_image[:len(bad)] = torch.from_numpy(np.frombuffer(bytes(bad), np.uint8).copy()).to("cuda:0")
    d_status.fill_(-1)
    torch.cuda.synchronize()
    assert lib.spl_dev_launch_inflate(ctypes.c_void_p(d_image.data_ptr()), *args, ctypes.c_void_p(d_out.data_ptr()),
                                      ctypes.c_void_p(d_status.data_ptr()), ctypes.c_void_p(d_work.data_ptr() if tokens else 0), ctypes.c_void_p(0)) == 0
    assert lib.spl_dev_launch_crc32(ctypes.c_void_p(d_out.data_ptr()), *args, ctypes.c_void_p(d_status.data_ptr()), ctypes.c_void_p(0)) == 0
    torch.cuda.synchronize()
    status = d_status.cpu().numpy()
    for k in range(len(streams)):
        assert (status[k] != 0) == (k % 7 == 0), (streams[k][0], k, int(status[k]))   # (the neighbours of a damaged block are untouched)
    assert d_out.cpu().numpy().tobytes()[at:at + 128] == b"\xa5" * 128
