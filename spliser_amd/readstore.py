"""``<out>.SpliSER.reads``: what ``process`` read from a sample's BAM, kept for ``combine`` (this build only; off by default).

The reference's ``combine`` goes back to a sample's BAM for every site the sample does not list (SpliSER_v0_1_8.py:869-904,
``checkBam`` at :903); here that is one more decode of the whole file per sample -- most of ``combine``'s time.  What
``checkBam`` reads from an alignment is flag, POS and CIGAR (:434-437): ``process --keepReads`` leaves exactly those, per
reference, BAM-native (pos int32, flag uint16, cig_off uint32, cigar uint32: 18.8 bytes a 150 bp read, a fifth of the BAM),
next to its ``.SpliSER.tsv``; ``combine`` takes them instead of the BAM WHEN THEY ARE THE BAM'S: the file is keyed by the BAM's
size, its modification time and the CRC32 of its first and last 64 KiB, and anything else -- another BAM, a newer one, a
truncated or foreign file -- is ignored and the BAM decoded as always.
"""
import json
import os
import struct
import zlib

import numpy as np

from . import samio

MAGIC = b"SPLREADS"
VERSION = 2       # (2: the payload's checksum in the header)
SUFFIX = ".SpliSER.reads"
_EDGE = 65536
_PIECE = 16 << 20


def _digest(view):
    """64 bits over a piece of the payload: xxh3 (10 GB/s, and it leaves the interpreter's lock) where the module is there, CRC32 otherwise."""
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(view)
    except ImportError:
        return zlib.crc32(view) & 0xFFFFFFFF


def _payload_sum(pieces, threads=8):
    """The payload's checksum: the pieces' digests (16 MB each, taken side by side on a few threads) hashed in file order."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        digests = list(pool.map(_digest, pieces))
    return "%016x" % _digest(struct.pack("<%dQ" % len(digests), *digests))


def path_for_tsv(tsv_path):
    """``X.SpliSER.tsv`` -> ``X.SpliSER.reads`` (None for a file that is not named like ``process``'s output)."""
    tail = ".SpliSER.tsv"
    return tsv_path[:-len(tail)] + SUFFIX if tsv_path.endswith(tail) else None


def bam_key(bam_path):
    """(size, mtime in ns, CRC32 over the first and the last 64 KiB) of the alignment file."""
    st = os.stat(bam_path)
    crc = 0
    with open(bam_path, "rb") as fh:
        crc = zlib.crc32(fh.read(_EDGE), crc)
        if st.st_size > _EDGE:
            fh.seek(max(_EDGE, st.st_size - _EDGE))
            crc = zlib.crc32(fh.read(_EDGE), crc)
    return int(st.st_size), int(st.st_mtime_ns), crc & 0xFFFFFFFF


def save(path, bam_path, reads_by_ref, threads=8):
    """reads_by_ref: [(reference name, ReadSet)] in file order.  Written beside its final name and moved there whole; the arrays
    go out in pieces of 16 MB on a few threads (``os.pwrite`` leaves the interpreter's lock: one thread copies 4 GB/s into the
    page cache, and a 20 M-read sample is 0.4 GB)."""
    from concurrent.futures import ThreadPoolExecutor
    import sys
    import time
    t0 = time.perf_counter()
    size, mtime_ns, crc = bam_key(bam_path)
    t_key = time.perf_counter()
    refs = [{"name": name, "n": int(rs.n), "ops": int(rs.cig_off[rs.n]) - int(rs.cig_off[0]) if rs.n else 0, "max_end": int(rs.max_end)}
            for name, rs in reads_by_ref]
    head_of = lambda digest: json.dumps({"version": VERSION, "bam_size": size, "bam_mtime_ns": mtime_ns, "bam_crc32": crc, "payload_sum": digest,   # noqa: E731
                                         "refs": refs}).encode("utf-8")
    fixed = MAGIC + struct.pack("<II", VERSION, len(head_of("0" * 16))) + head_of("0" * 16)     # (its length does not depend on the digest)
    at = len(fixed) + (-len(fixed) % 64)
    jobs = []      # (file offset, contiguous array)
    for kind, dt in (("pos", np.int32), ("flag", np.uint16), ("cig_off", np.uint32), ("cigar", np.uint32)):
        for _, rs in reads_by_ref:
            a = getattr(rs, kind)
            base = int(rs.cig_off[0]) if rs.n else 0     # (offsets count from the reference's first op)
            if kind == "cig_off":
                a = a[:rs.n + 1] if rs.n else np.zeros(1, np.uint32)
                if base:
                    a = (a.astype(np.int64) - base).astype(np.uint32)
            elif kind == "cigar":
                a = a[base:int(rs.cig_off[rs.n])] if rs.n else a[:0]
            else:
                a = a[:rs.n]
            a = np.ascontiguousarray(a, dtype=dt)
            if a.nbytes:
                jobs.append((at, a))
            at += a.nbytes
        at += -at % 64
    tmp = path + ".tmp%d" % os.getpid()
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, at)
        piece = _PIECE
        parts = [(off + lo, memoryview(a).cast("B")[lo:lo + piece]) for off, a in jobs for lo in range(0, a.nbytes, piece)]
        # the checksum of what goes out, array by array in pieces of 16 MB (what open_if_fresh takes again, the same way)
        t_jobs = time.perf_counter()
        head = head_of(_payload_sum([v for _, v in parts], threads))
        t_sum = time.perf_counter()
        os.pwrite(fd, MAGIC + struct.pack("<II", VERSION, len(head)) + head, 0)

        def put(part):
            off, view = part
            while len(view):
                n = os.pwrite(fd, view, off)
                off, view = off + n, view[n:]
        with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
            list(pool.map(put, parts))
        if os.environ.get("SPL_PROCESS_TIMING"):
            sys.stderr.write("[readstore] key of the BAM %.4f s, arrays made contiguous %.4f, checksum %.4f, %d pieces written %.4f\n"
                             % (t_key - t0, t_jobs - t_key, t_sum - t_jobs, len(parts), time.perf_counter() - t_sum))
    finally:
        os.close(fd)
    os.replace(tmp, path)


class ReadStore(object):
    """The reads of a ``.SpliSER.reads`` file as a source for ``process_sites``: ``reads(chrom)`` -> ReadSet (views of the file's
    bytes in memory) or None for a reference the BAM does not have."""

    def __init__(self, path, head, mm):
        self.path, self._mm = path, mm
        self._sets = {}
        refs = head["refs"]
        n_all, ops_all = sum(r["n"] for r in refs), sum(r["ops"] for r in refs)
        at = 16 + head["_head_len"]
        at += -at % 64

        def take(count, dt):
            nonlocal at
            a = np.frombuffer(mm, dtype=dt, count=count, offset=at)
            at += count * np.dtype(dt).itemsize
            at += -at % 64
            return a
        pos, flag = take(n_all, np.int32), take(n_all, np.uint16)
        cig_off, cigar = take(n_all + len(refs), np.uint32), take(ops_all, np.uint32)
        # The payload is what the header's checksum says, and every reference's CIGAR offsets begin at 0, never step back and end at
        # its number of ops: whoever walks cigar[cig_off[i] .. cig_off[i + 1]) (the native packer does, without asking) stays inside
        # the file.  ValueError: the caller decodes the BAM instead.
        pieces, b = [], {"pos": 0, "flag": 0, "cig_off": 0, "cigar": 0}     # (the pieces `save` took: array by array, reference by reference)
        for kind, arr, per in (("pos", pos, lambda r: r["n"]), ("flag", flag, lambda r: r["n"]), ("cig_off", cig_off, lambda r: r["n"] + 1), ("cigar", cigar, lambda r: r["ops"])):
            for r in refs:
                a = arr[b[kind]:b[kind] + per(r)]
                b[kind] += per(r)
                pieces.extend(memoryview(a).cast("B")[lo:lo + _PIECE] for lo in range(0, a.nbytes, _PIECE))
        if _payload_sum(pieces) != head.get("payload_sum"):
            raise ValueError("%s: the payload is not what its checksum says" % path)
        r0 = o0 = c0 = 0
        for r in refs:
            n, ops = r["n"], r["ops"]
            off = cig_off[c0:c0 + n + 1]
            if int(off[0]) != 0 or int(off[n]) != ops or (n and not bool(np.all(off[1:] >= off[:-1]))):
                raise ValueError("%s: CIGAR offsets of %s are not those of %d ops" % (path, r["name"], ops))
            self._sets[r["name"]] = samio.ReadSet(pos[r0:r0 + n], flag[r0:r0 + n], cig_off[c0:c0 + n + 1], cigar[o0:o0 + ops], max_end=r["max_end"])
            r0, o0, c0 = r0 + n, o0 + ops, c0 + n + 1
        self.n_reads = n_all

    def reads(self, chrom):
        return self._sets.get(chrom)

    def close(self):
        self._sets = {}
        self._mm = None


def open_if_fresh(path, bam_path):
    """-> ReadStore when ``path`` holds the reads of exactly this alignment file, None otherwise (missing, another version,
    damaged, or the BAM has changed since)."""
    if not path or not os.path.exists(path):
        return None
    try:
        with open(path, "rb") as fh:
            fixed = fh.read(16)
            if len(fixed) < 16 or fixed[:8] != MAGIC:
                return None
            version, head_len = struct.unpack("<II", fixed[8:16])
            if version != VERSION or head_len > (1 << 26):
                return None
            head = json.loads(fh.read(head_len).decode("utf-8"))
            if (head.get("bam_size"), head.get("bam_mtime_ns"), head.get("bam_crc32")) != bam_key(bam_path):
                return None
            refs = head["refs"]
            if any(not isinstance(r.get("n"), int) or not isinstance(r.get("ops"), int) or r["n"] < 0 or r["ops"] < 0 for r in refs):
                return None
            n_all, ops_all = sum(r["n"] for r in refs), sum(r["ops"] for r in refs)
            need = 16 + head_len
            need += -need % 64
            for count, size in ((n_all, 4), (n_all, 2), (n_all + len(refs), 4), (ops_all, 4)):
                need += count * size
                need += -need % 64
            if os.fstat(fh.fileno()).st_size < need:
                return None
            head["_head_len"] = head_len
            # mapped, with the pages looked up at once (MAP_POPULATE: one call instead of 90 000 faults for a 20 M-read sample);
            # read into memory of this process's own -- pieces on threads -- was measured and is slower: a copy more
            import mmap
            flags = mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0)
            buf = mmap.mmap(fh.fileno(), 0, flags=flags, prot=mmap.PROT_READ)
        return ReadStore(path, head, buf)
    except (OSError, ValueError, KeyError, TypeError):
        return None
