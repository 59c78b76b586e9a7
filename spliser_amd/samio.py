"""Alignment I/O on the host: SoA read sets, a SAM-text reader and a BAM/BGZF writer.

The GPU path consumes, per chromosome, exactly the three SAM columns ``checkBam`` reads from each
line ``samtools view`` prints (flag, POS, CIGAR -- SpliSER_v0_1_8.py:434-437) as structure-of-arrays
(``include/spliser.h``: ``spl_reads``).  Production input is BAM, decoded by the native reader in
``csrc/bam_reader.cpp`` (see ``native.BamFile``).  This module adds

  * ``read_sam``   -- a plain SAM-text reader (small inputs, fixtures; ``samtools view`` accepts SAM too),
  * ``write_bam``  -- a spec-conformant BGZF/BAM writer used to make test and synthetic BAM files
                      (the image has no samtools/htslib), and ``write_sam`` for their text twins.
"""
import re
import struct
import zlib

import numpy as np

__all__ = ["ReadSet", "read_sam", "write_bam", "write_sam", "cigar_ops", "cigar_string", "OP_CODES"]

OP_CODES = "MIDNSHP=XB"
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
_OP_OF = {c: i for i, c in enumerate(OP_CODES)}


def cigar_ops(cigar):
    """'50M100N50M' -> [50<<4|0, 100<<4|3, 50<<4|0]; '*' -> []."""
    if cigar == "*":
        return []
    return [(int(n) << 4) | _OP_OF[c] for n, c in _CIGAR_RE.findall(cigar)]


def cigar_string(ops):
    return "".join("%d%s" % (int(o) >> 4, OP_CODES[int(o) & 15]) for o in ops) or "*"


class ReadSet(object):
    """Reads of one chromosome in file order (SoA).  ``pos`` is 1-based (SAM POS)."""
    __slots__ = ("pos", "flag", "cig_off", "cigar", "max_end")

    def __init__(self, pos, flag, cig_off, cigar, max_end=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.flag = np.ascontiguousarray(flag, dtype=np.uint16)
        self.cig_off = np.ascontiguousarray(cig_off, dtype=np.uint32)
        self.cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        if max_end is None:
            max_end = self._max_end()
        self.max_end = int(max_end)

    @property
    def n(self):
        return int(self.pos.shape[0])

    def _max_end(self):
        if self.n == 0:
            return 0
        code = self.cigar & 15
        length = (self.cigar >> 4).astype(np.int64)
        consumes = (code == 0) | (code == 2) | (code == 3) | (code == 7) | (code == 8)
        csum = np.concatenate(([0], np.cumsum(np.where(consumes, length, 0))))
        ref_len = csum[self.cig_off[1:].astype(np.int64)] - csum[self.cig_off[:-1].astype(np.int64)]
        return int((self.pos.astype(np.int64) + np.maximum(ref_len, 1) - 1).max())

    def take(self, lo, hi):
        """Reads [lo, hi) as a ReadSet of their own (views where the arrays allow; the CIGAR offsets begin at 0 again; the bound on
        their ends is the whole set's, which it cannot exceed)."""
        lo, hi = max(0, int(lo)), min(self.n, int(hi))
        if hi <= lo:
            return ReadSet.empty()
        o0, o1 = int(self.cig_off[lo]), int(self.cig_off[hi])
        off = self.cig_off[lo:hi + 1] if o0 == 0 else (self.cig_off[lo:hi + 1].astype(np.int64) - o0).astype(np.uint32)
        return ReadSet(self.pos[lo:hi], self.flag[lo:hi], off, self.cigar[o0:o1], max_end=self.max_end)

    @classmethod
    def empty(cls):
        return cls(np.zeros(0, np.int32), np.zeros(0, np.uint16), np.zeros(1, np.uint32), np.zeros(0, np.uint32), 0)

    @classmethod
    def from_records(cls, records):
        """records: iterable of (flag, pos, cigar_string)."""
        pos, flag, off, ops = [], [], [0], []
        for f, p, c in records:
            pos.append(p)
            flag.append(f)
            ops.extend(cigar_ops(c))
            off.append(len(ops))
        return cls(np.asarray(pos, np.int64), np.asarray(flag, np.int64), np.asarray(off, np.int64),
                   np.asarray(ops, np.int64))


def read_sam(path):
    """Parse SAM text -> (ref_names, {chrom: ReadSet}).  Keeps every record that has an RNAME, no flag
    or MAPQ filtering, file order -- what ``samtools view`` (no -F/-q) would print."""
    names, per = [], {}
    with open(path, "r") as handle:
        for line in handle:
            if line.startswith("@"):
                if line.startswith("@SQ"):
                    for field in line.rstrip("\n").split("\t")[1:]:
                        if field.startswith("SN:"):
                            names.append(field[3:])
                continue
            cols = line.split("\t")
            if len(cols) < 6 or cols[2] == "*":
                continue
            rec = per.get(cols[2])
            if rec is None:
                rec = per[cols[2]] = ([], [], [0], [])
                if cols[2] not in names:
                    names.append(cols[2])
            rec[0].append(int(cols[3]))
            rec[1].append(int(cols[1]))
            rec[3].extend(cigar_ops(cols[5].strip()))
            rec[2].append(len(rec[3]))
    sets = {c: ReadSet(np.asarray(p, np.int64), np.asarray(f, np.int64), np.asarray(o, np.int64),
                       np.asarray(g, np.int64)) for c, (p, f, o, g) in per.items()}
    return names, sets


def write_sam(path, ref_names, ref_lengths, chrom_reads):
    """chrom_reads: list of (chrom, ReadSet) in file order."""
    with open(path, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for n, ln in zip(ref_names, ref_lengths):
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (n, ln))
        i = 0
        for chrom, rs in chrom_reads:
            for k in range(rs.n):
                ops = rs.cigar[rs.cig_off[k]:rs.cig_off[k + 1]]
                fh.write("r%d\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\n" % (i, rs.flag[k], chrom, rs.pos[k], cigar_string(ops)))
                i += 1


# ---------------------------------------------------------------------------------------------------
# BGZF / BAM writer (SAM spec section 4): enough of the format for htslib-compatible files.

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_BGZF_BLOCK = 0xFF00  # uncompressed payload per block (< 64 KiB)


def _bgzf_block(payload, level):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    data = comp.compress(payload) + comp.flush()
    bsize = len(data) + 25  # total block size - 1
    if bsize > 0xFFFF:
        raise ValueError("BGZF block too large")
    header = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 0x42, 0x43, 2, bsize)
    return header + data + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bam(path, ref_names, ref_lengths, chrom_reads, level=1, with_seq=False, unplaced=0, long_cigar_tag=False):
    """Write a BAM file.  chrom_reads: list of (chrom, ReadSet) in file order.

    with_seq       -- emit a dummy SEQ/QUAL of the query length (realistic record size) instead of '*'
    unplaced       -- append this many records without a reference (tid -1) at the end
    long_cigar_tag -- store every CIGAR with more than 3 ops the way htslib stores >65535-op CIGARs:
                      a placeholder ``<qlen>S<rlen>N`` in the record and the real ops in a CG:B,I tag
    """
    tid_of = {n: i for i, n in enumerate(ref_names)}
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, ln) for n, ln in zip(ref_names, ref_lengths))
    head = [b"BAM\x01", struct.pack("<i", len(text)), text.encode("ascii"), struct.pack("<i", len(ref_names))]
    for n, ln in zip(ref_names, ref_lengths):
        nb = n.encode("ascii") + b"\x00"
        head += [struct.pack("<i", len(nb)), nb, struct.pack("<i", ln)]
    buf = bytearray(b"".join(head))
    with open(path, "wb") as fh:
        def flush(final=False):
            nonlocal buf
            while len(buf) >= _BGZF_BLOCK or (final and len(buf)):
                fh.write(_bgzf_block(bytes(buf[:_BGZF_BLOCK]), level))
                del buf[:_BGZF_BLOCK]

        serial = 0
        for chrom, rs in chrom_reads:
            tid = tid_of[chrom]
            for k in range(rs.n):
                ops = [int(o) for o in rs.cigar[rs.cig_off[k]:rs.cig_off[k + 1]]]
                qlen = sum(o >> 4 for o in ops if (o & 15) in (0, 1, 4, 7, 8))
                rlen = sum(o >> 4 for o in ops if (o & 15) in (0, 2, 3, 7, 8))
                name = ("r%d" % serial).encode("ascii") + b"\x00"
                serial += 1
                pos0 = int(rs.pos[k]) - 1
                tags = b""
                rec_ops = ops
                l_seq = qlen if with_seq else 0
                if long_cigar_tag and len(ops) > 3:
                    rec_ops = [(qlen << 4) | 4, (rlen << 4) | 3]   # htslib: <l_seq>S<rlen>N placeholder
                    tags = b"NMC\x00" + b"CGBI" + struct.pack("<i", len(ops)) + struct.pack("<%dI" % len(ops), *ops)
                    l_seq = qlen
                seq = bytes([0x11]) * ((l_seq + 1) // 2)
                qual = bytes([30]) * l_seq
                flag = int(rs.flag[k])
                end0 = pos0 + (rlen if (rlen and not flag & 4) else 1)
                body = struct.pack("<iiBBHHHiiii", tid, pos0, len(name), 60, _reg2bin(pos0, end0), len(rec_ops), flag,
                                   l_seq, -1, -1, 0) + name + struct.pack("<%dI" % len(rec_ops), *rec_ops) + seq + qual + tags
                buf += struct.pack("<i", len(body)) + body
                if len(buf) >= _BGZF_BLOCK:
                    flush()
        for k in range(unplaced):
            name = ("u%d" % k).encode("ascii") + b"\x00"
            body = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, 0, -1, -1, 0) + name
            buf += struct.pack("<i", len(body)) + body
        flush(final=True)
        fh.write(_BGZF_EOF)
