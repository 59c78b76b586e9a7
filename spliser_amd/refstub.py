"""The binding a SpliSER maintainer would put into SpliSER_v0_1_8.py: ``processSites`` (:681-692) on libspliser_hip.so.

INTEGRATION.md quotes this file.  It touches the reference's objects only through the accessors the reference itself uses
(``Site.getPos / getStrand / getPartners / getPartnerCounts / getCompetitorPos / getAlphaCount`` and the adders,
Gene_Site_Iter_Graph_v0_1_8.py:98-339), so ``marshal_sites`` can be -- and is, in the build container -- run on the reference's
LIVE ``site2D_array`` after its own Steps 0-2 and held against the arrays this build's site table gives for the same inputs
(tests/test_refstub_reference.py, ``-m reference``).  ``install(module_globals)`` replaces the reference's ``processSites``.
"""
import ctypes

import numpy as np


class _Sites(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64)] + [(k, ctypes.c_void_p) for k in
                ("pos", "strand", "part_off", "part_pos", "part_site", "comp_off", "comp_pos", "alpha", "edge_cnt")]


class _Reads(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64)] + [(k, ctypes.c_void_p) for k in ("pos", "flag", "cig_off", "cigar")]


class _Opts(ctypes.Structure):
    _fields_ = [("stranded", ctypes.c_int32), ("combine_mode", ctypes.c_int32), ("flags", ctypes.c_int32)]


def marshal_sites(sites, sample=0):
    """One chromosome's Site objects (already in ``Site.__lt__`` order, as ``site2D_array[k]`` is) -> the arrays of ``spl_sites``:
    rows in that order; a row's partners as its ``getPartners()`` list has them (the order of the Partners column), one edge per
    partner Site with that Site's row and the read count ``PartnerCounts`` holds for its position (two partner Sites at one
    position -- a BED strand that is none in a stranded analysis -- are two edges); competitors as ``getCompetitorPos()`` has
    them."""
    row = {id(s): i for i, s in enumerate(sites)}
    pos = np.array([s.getPos() for s in sites], np.int32)
    strand = np.array([ord(s.getStrand()[:1] or "\0") for s in sites], np.uint8)
    part_off = np.zeros(len(sites) + 1, np.uint32)
    comp_off = part_off.copy()
    part_pos, part_site, edge_cnt, comp_pos = [], [], [], []
    for i, s in enumerate(sites):
        counts = s.getPartnerCounts()
        for p in s.getPartners():               # the Site objects findBeta2Counts walks (:590); the Partners column's order
            part_pos.append(p.getPos())
            part_site.append(row.get(id(p), -1))
            edge_cnt.append(counts[p.getPos()][sample])
        comp_pos += s.getCompetitorPos()
        part_off[i + 1], comp_off[i + 1] = len(part_pos), len(comp_pos)
    return dict(pos=pos, strand=strand, part_off=part_off, part_pos=np.array(part_pos, np.int32),
                part_site=np.array(part_site, np.int32), comp_off=comp_off, comp_pos=np.array(comp_pos, np.int32),
                alpha=np.array([s.getAlphaCount(sample) for s in sites], np.int64), edge_cnt=np.array(edge_cnt, np.int64))


def make_process_sites(g, lib_path="libspliser_hip.so"):
    """-> a ``processSites`` for the reference module whose globals are ``g`` (``chrom_index``, ``site2D_array``)."""
    spl = ctypes.CDLL(lib_path)
    spl.spl_last_error.restype = ctypes.c_char_p
    spl.spl_bam_ref_name.restype = ctypes.c_char_p

    def processSites(inBAM, qChrom, isStranded, strandedType, isbeta2Cryptic, sample=0, numsamples=1):
        ctx, bam = ctypes.c_void_p(), ctypes.c_void_p()
        assert spl.spl_create(0, ctypes.byref(ctx)) == 0, spl.spl_last_error()
        assert spl.spl_bam_open(inBAM.encode(), 0, ctypes.byref(bam)) == 0, spl.spl_last_error()
        names = [spl.spl_bam_ref_name(bam, i).decode() for i in range(spl.spl_bam_n_ref(bam))]
        for c in g["chrom_index"]:
            if not (qChrom == c or qChrom == "All") or c not in names:
                continue
            sites = g["site2D_array"][g["chrom_index"].index(c)]
            if not sites:
                continue
            a = marshal_sites(sites, sample)
            order = ("pos", "strand", "part_off", "part_pos", "part_site", "comp_off", "comp_pos", "alpha", "edge_cnt")
            S = _Sites(len(sites), *[a[k].ctypes.data for k in order])
            R = _Reads()
            assert spl.spl_bam_reads(bam, names.index(c), ctypes.byref(R), None) == 0, spl.spl_last_error()
            O = _Opts({None: 0, "fr": 1, "rf": 2}[strandedType if isStranded else None], 0, 0)
            b1, b2r = np.zeros(len(sites), np.uint32), np.zeros(len(sites), np.uint32)
            dbl = np.zeros(max(len(a["part_pos"]), 1), np.uint32)
            assert spl.spl_count(ctx, ctypes.byref(S), ctypes.byref(R), ctypes.byref(O), b1.ctypes.data_as(ctypes.c_void_p),
                                 b2r.ctypes.data_as(ctypes.c_void_p), dbl.ctypes.data_as(ctypes.c_void_p)) == 0, spl.spl_last_error()
            b2s, b2c = np.zeros(len(sites), np.int64), np.zeros(len(sites), np.int64)
            b2w, sse = np.zeros(len(sites)), np.zeros(len(sites))
            assert spl.spl_sse(ctx, ctypes.byref(S), b1.ctypes.data_as(ctypes.c_void_p), b2r.ctypes.data_as(ctypes.c_void_p),
                               dbl.ctypes.data_as(ctypes.c_void_p), int(isbeta2Cryptic), b2s.ctypes.data_as(ctypes.c_void_p),
                               b2c.ctypes.data_as(ctypes.c_void_p), b2w.ctypes.data_as(ctypes.c_void_p), sse.ctypes.data_as(ctypes.c_void_p)) == 0, spl.spl_last_error()
            for i, s in enumerate(sites):                        # write back through the Site adders
                s.addBeta1Count(int(b1[i]), sample)
                s.addBeta2SimpleCount(int(b2s[i]), sample)
                s.addBeta2CrypticCount(int(b2c[i]), sample)
                vals = list(s.getBeta2WeightedCounts())          # (findBeta2Counts replaces the whole list, :623: only
                if len(vals) != numsamples:                      #  this sample's entry is this call's to set)
                    vals = [0.00] * numsamples
                vals[sample] = float(b2w[i])
                s.updateBeta2Weighted(vals)
                s.setSSE(float(sse[i]), sample)
        spl.spl_bam_close(bam)
        spl.spl_destroy(ctx)
    return processSites


def install(g, lib_path="libspliser_hip.so"):
    g["processSites"] = make_process_sites(g, lib_path)
