"""ctypes front-end of oracle/spliser_oracle.c -- TEST INFRASTRUCTURE (see that file's header).

Callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "spliser_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_check_bam.restype = ctypes.c_int
        _LIB.orc_beta2_sse.restype = ctypes.c_int
        _LIB.orc_check_strand.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    """Threads for the site loop of ``check_bam`` (default 1, like the reference).  Only bench.py's all-cores CPU baseline
    asks for more."""
    lib().orc_set_threads(ctypes.c_int(int(n)))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def check_bam(site_pos, site_strand, part_off, part_pos, comp_off, comp_pos, r_pos, r_flag, cig_off, cigar,
              stranded=0, combine_mode=0):
    """checkBam over every site of one chromosome -> (beta1, beta2s_reads, dbl) uint32 arrays."""
    site_pos, site_strand = _c(site_pos, np.int32), _c(site_strand, np.uint8)
    part_off, part_pos = _c(part_off, np.uint32), _c(part_pos, np.int32)
    comp_off, comp_pos = _c(comp_off, np.uint32), _c(comp_pos, np.int32)
    r_pos, r_flag = _c(r_pos, np.int32), _c(r_flag, np.uint16)
    cig_off, cigar = _c(cig_off, np.uint32), _c(cigar, np.uint32)
    n = site_pos.shape[0]
    beta1 = np.zeros(n, np.uint32)
    b2s = np.zeros(n, np.uint32)
    dbl = np.zeros(max(int(part_off[-1]) if n else 0, 1), np.uint32)
    rc = lib().orc_check_bam(ctypes.c_int64(n), _p(site_pos), _p(site_strand), _p(part_off), _p(part_pos),
                             _p(comp_off), _p(comp_pos), ctypes.c_int64(r_pos.shape[0]), _p(r_pos), _p(r_flag),
                             _p(cig_off), _p(cigar), ctypes.c_int(stranded), ctypes.c_int(combine_mode),
                             _p(beta1), _p(b2s), _p(dbl))
    if rc != 0:
        raise MemoryError("orc_check_bam failed")
    return beta1, b2s, dbl[: int(part_off[-1]) if n else 0]


def beta2_sse(site_pos, part_off, part_pos, part_site, alpha, edge_cnt, beta1, b2s_reads, dbl, cryptic):
    """findBeta2Counts + calculateSSE -> (beta2_simple i64, beta2_cryptic i64, beta2_weighted f64, sse f64)."""
    site_pos = _c(site_pos, np.int32)
    part_off, part_pos, part_site = _c(part_off, np.uint32), _c(part_pos, np.int32), _c(part_site, np.int32)
    alpha, edge_cnt = _c(alpha, np.int64), _c(edge_cnt, np.int64)
    beta1, b2s_reads = _c(beta1, np.uint32), _c(b2s_reads, np.uint32)
    dbl = _c(dbl if len(dbl) else np.zeros(1), np.uint32)
    n = site_pos.shape[0]
    b2s = np.zeros(n, np.int64)
    b2c = np.zeros(n, np.int64)
    b2w = np.zeros(n, np.float64)
    sse = np.zeros(n, np.float64)
    lib().orc_beta2_sse(ctypes.c_int64(n), _p(site_pos), _p(part_off), _p(part_pos), _p(part_site), _p(alpha),
                        _p(edge_cnt), _p(beta1), _p(b2s_reads), _p(dbl), ctypes.c_int(1 if cryptic else 0),
                        _p(b2s), _p(b2c), _p(b2w), _p(sse))
    return b2s, b2c, b2w, sse


def junction_table(r_pos, r_flag, cig_off, cigar, stranded=0, min_anchor=0, min_intron=0, max_intron=0):
    """Plain-Python restatement of the junction table (checker for spl_junctions; small inputs only).

    Walks every read like checkBam walks its CIGAR (SpliSER_v0_1_8.py:457-483): M,=,X,D,N advance the cursor, an N op of
    length d ending at ``cur`` is the junction (cur-d-1, cur-1).  Records flagged unmapped (0x4) carry none.  Read strand
    by check_strand's rule (:374-406) when ``stranded`` is 1 (fr) or 2 (rf).  -> sorted list of
    (left, right, strand byte, count, anchor_left, anchor_right); anchors = reference bases of the read between the
    junction and the neighbouring N op / read end, maximum over the reads.  ``min_anchor`` / ``min_intron`` / ``max_intron`` (0 =
    no limit): a read counts for a junction only if both its anchors and the N op's length pass (regtools' -a / -m / -M).
    """
    table = {}
    for i in range(len(r_pos)):
        flag = int(r_flag[i])
        if flag & 4 or int(r_pos[i]) < 0:
            continue
        if stranded:
            first = bool(flag & 64) or not (flag & 1)
            rev = bool(flag & 16)
            plus = (first != rev) if stranded == 1 else (first == rev)
            strand = ord("+") if plus else ord("-")
        else:
            strand = ord("?")
        ops = [(int(o) & 15, int(o) >> 4) for o in cigar[int(cig_off[i]):int(cig_off[i + 1])]]
        cur = int(r_pos[i])
        before = 0
        for k, (code, d) in enumerate(ops):
            if code not in (0, 2, 3, 7, 8):
                continue
            cur += d
            if code != 3:
                before += d
                continue
            after = 0
            for code2, d2 in ops[k + 1:]:
                if code2 == 3:
                    break
                if code2 in (0, 2, 7, 8):
                    after += d2
            if before >= min_anchor and after >= min_anchor and d >= min_intron and (max_intron == 0 or d <= max_intron):
                key = (cur - d - 1, cur - 1, strand)
                c, a, b = table.get(key, (0, 0, 0))
                table[key] = (c + 1, max(a, before), max(b, after))
            before = 0
    return [k + table[k] for k in sorted(table)]
