"""Builds/loads tests/hostsim/classify_host.cpp (test-only host build of spl_classify.h)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libclassify_host.so")
        srcs = [os.path.join(HERE, "classify_host.cpp"), os.path.join(HERE, "..", "..", "spliser_amd", "csrc", "spl_classify.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, srcs[0]])
        _lib = ctypes.CDLL(so)
    return _lib


def count(arr, reads, stranded, combine_mode):
    def c(a, dt):
        return np.ascontiguousarray(a, dtype=dt)

    def p(a):
        return a.ctypes.data_as(ctypes.c_void_p)
    sp, ss = c(arr.pos, np.int32), c(arr.strand, np.uint8)
    po, pp, co, cp = c(arr.part_off, np.uint32), c(arr.part_pos, np.int32), c(arr.comp_off, np.uint32), c(arr.comp_pos, np.int32)
    n = sp.shape[0]
    beta1, b2s = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    dbl = np.zeros(max(int(po[-1]) if n else 0, 1), np.uint32)
    lib().sim_count(ctypes.c_int64(n), p(sp), p(ss), p(po), p(pp), p(co), p(cp), ctypes.c_int64(reads.n), p(reads.pos),
                    p(reads.flag), p(reads.cig_off), p(reads.cigar), ctypes.c_int(stranded), ctypes.c_int(combine_mode),
                    p(beta1), p(b2s), p(dbl))
    return beta1[:n], b2s[:n], dbl[: int(po[-1]) if n else 0]
