"""ctypes binding of libspliser_hip.so (C ABI: include/spliser.h).

This is the only door to the compute path.  There is no Python/numpy fallback: if the shared library
is missing or no MI355X is visible, every entry point raises -- loudly -- instead of computing on the
host.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SPLISER_HIP_LIB: load another build of the same library (A/B runs of two builds on one box)
LIB_PATH = os.environ.get("SPLISER_HIP_LIB") or os.path.join(HERE, "libspliser_hip.so")
CSRC = os.path.join(HERE, "csrc")

STRANDED_CODE = {None: 0, False: 0, "": 0, "fr": 1, "rf": 2}


class SpliserNativeError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libspliser_hip error %d: %s" % (code, message))
        self.code = code


class spl_sites(ctypes.Structure):
    _fields_ = [("n_sites", ctypes.c_int64), ("pos", ctypes.c_void_p), ("strand", ctypes.c_void_p),
                ("part_off", ctypes.c_void_p), ("part_pos", ctypes.c_void_p), ("part_site", ctypes.c_void_p),
                ("comp_off", ctypes.c_void_p), ("comp_pos", ctypes.c_void_p), ("alpha", ctypes.c_void_p),
                ("edge_cnt", ctypes.c_void_p)]


class spl_reads(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_int64), ("pos", ctypes.c_void_p), ("flag", ctypes.c_void_p),
                ("cig_off", ctypes.c_void_p), ("cigar", ctypes.c_void_p)]


class spl_opts(ctypes.Structure):
    _fields_ = [("stranded", ctypes.c_int32), ("combine_mode", ctypes.c_int32), ("flags", ctypes.c_int32)]


OPT_PAIR_KERNEL = 1
OPT_WAVE_AGGREGATION = 2


EXPORTS = [
    "spl_abi_version", "spl_last_error", "spl_device_count", "spl_trim", "spl_create", "spl_create_on_stream", "spl_destroy",
    "spl_sync", "spl_pass_barrier", "spl_timer_begin", "spl_timer_end", "spl_kernel_timing_begin", "spl_kernel_timing_collect", "spl_prof_enable", "spl_prof_report", "spl_count", "spl_sse", "spl_sites_upload", "spl_sites_free",
    "spl_reads_upload", "spl_reads_upload_segments", "spl_reads_begin", "spl_reads_begin_sized", "spl_reads_add", "spl_reads_add2", "spl_reads_add_bam", "spl_reads_add_bam_share", "spl_reads_finish",
    "spl_soa_upload", "spl_soa_upload2", "spl_soa_free", "spl_reads_add_soa", "spl_reads_relayout", "spl_layout_timing_collect", "spl_reads_layout_bytes",
    "spl_pack_host", "spl_reads_free", "spl_count_launch", "spl_sse_launch", "spl_counters_download",
    "spl_sse_download", "spl_count_algorithmic_bytes", "spl_literal_queue_size", "spl_last_launch_info", "spl_bam_open", "spl_bam_open_stream", "spl_bam_open_deferred", "spl_bam_decode_device", "spl_bam_reserve_device", "spl_bam_share_plan", "spl_bam_share_range", "spl_bam_share_info", "spl_bam_share_count_host", "spl_bam_share_ref", "spl_bam_decode_device_share", "spl_bam_decoded_on_device", "spl_bam_wait_device", "spl_bam_start", "spl_bam_compression_ratio", "spl_bam_sample", "spl_bam_wait_ref", "spl_bam_wait_all", "spl_bam_cancel", "spl_bam_decline_reason", "spl_bam_close",
    "spl_bam_n_ref", "spl_bam_ref_name", "spl_bam_ref_length", "spl_bam_n_records", "spl_bam_reads", "spl_bam_write", "spl_bam_write2",
    "spl_gene_search", "spl_junctions", "spl_junctions_get", "spl_tsv_append", "spl_tsv_append_many", "spl_fmt_fixed",
    "spl_bed_open", "spl_gff_open", "spl_text_close", "spl_text_rows", "spl_text_n_chrom", "spl_text_chrom_name", "spl_text_chrom",
    "spl_text_i64", "spl_text_strand", "spl_text_names",
    "spl_combine_open", "spl_combine_close", "spl_combine_rows", "spl_combine_n_texts", "spl_combine_text", "spl_combine_region_runs",
    "spl_combine_keep_gene", "spl_combine_merge", "spl_combine_n_sites", "spl_combine_n_gap_sites", "spl_combine_skipped",
    "spl_combine_n_tables", "spl_combine_table", "spl_combine_answers", "spl_combine_write", "spl_fmt_repr",
]

_lib = None


def build(force=False):
    """Compile libspliser_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpliserNativeError(-2, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                         "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.spl_last_error.restype = ctypes.c_char_p
        L.spl_bam_ref_name.restype = ctypes.c_char_p
        L.spl_bam_decline_reason.restype = ctypes.c_char_p
        L.spl_bam_ref_length.restype = ctypes.c_int64
        L.spl_bam_n_records.restype = ctypes.c_int64
        for name in ("spl_destroy", "spl_sites_free", "spl_reads_free", "spl_soa_free", "spl_bam_close", "spl_text_close", "spl_combine_close"):
            getattr(L, name).restype = None
        for name in ("spl_combine_rows", "spl_combine_region_runs", "spl_combine_n_sites", "spl_combine_n_gap_sites", "spl_combine_skipped"):
            getattr(L, name).restype = ctypes.c_int64
        L.spl_combine_text.restype = ctypes.c_char_p
        L.spl_text_rows.restype = ctypes.c_int64
        L.spl_text_chrom_name.restype = ctypes.c_char_p
        for name in ("spl_text_chrom", "spl_text_i64", "spl_text_strand", "spl_text_names"):
            getattr(L, name).restype = ctypes.c_void_p
        if L.spl_abi_version() != 1:
            raise SpliserNativeError(-1, "ABI version mismatch")
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SpliserNativeError(rc, lib().spl_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _to_i32(a, what):
    a = np.asarray(a)
    if a.size and (a.max() > 2147483581 or a.min() < -2147483648):
        raise SpliserNativeError(-6, "%s exceeds the int32 shard coordinate space; split the shard" % what)
    return _arr(a, np.int32)


def device_count():
    n = ctypes.c_int(0)
    _check(lib().spl_device_count(ctypes.byref(n)))
    return n.value


class SiteArrays(object):
    """Keeps the numpy arrays behind a ``spl_sites`` alive."""

    def __init__(self, pos, strand, part_off, part_pos, comp_off, comp_pos, part_site=None, alpha=None, edge_cnt=None):
        self.pos = _to_i32(pos, "site position")
        self.strand = _arr(strand, np.uint8)
        self.part_off = _arr(part_off, np.uint32)
        self.part_pos = _to_i32(part_pos, "partner position")
        self.comp_off = _arr(comp_off, np.uint32)
        self.comp_pos = _to_i32(comp_pos, "competitor position")
        self.part_site = None if part_site is None else _arr(part_site, np.int32)
        self.alpha = None if alpha is None else _arr(alpha, np.int64)
        self.edge_cnt = None if edge_cnt is None else _arr(edge_cnt, np.int64)
        self.n = int(self.pos.shape[0])
        self.n_part = int(self.part_off[-1]) if self.n else 0
        self.c = spl_sites(self.n, _ptr(self.pos), _ptr(self.strand), _ptr(self.part_off), _ptr(self.part_pos),
                           _ptr(self.part_site), _ptr(self.comp_off), _ptr(self.comp_pos), _ptr(self.alpha),
                           _ptr(self.edge_cnt))

    @classmethod
    def from_chrom(cls, arr, offset=0):
        """From sites.ChromArrays, optionally shifted into a shard coordinate space."""
        return cls(arr.pos + offset, arr.strand, arr.part_off, arr.part_pos + offset, arr.comp_off,
                   arr.comp_pos + offset, arr.part_site, arr.alpha, arr.edge_cnt)


class ReadArrays(object):
    def __init__(self, pos, flag, cig_off, cigar):
        self.pos = _to_i32(pos, "read position")
        self.flag = _arr(flag, np.uint16)
        self.cig_off = _arr(cig_off, np.uint32)
        self.cigar = _arr(cigar, np.uint32)
        self.n = int(self.pos.shape[0])
        if self.cig_off.shape[0] != self.n + 1:
            raise ValueError("cig_off must have n_reads + 1 entries")
        self.c = spl_reads(self.n, _ptr(self.pos), _ptr(self.flag), _ptr(self.cig_off), _ptr(self.cigar))


class Context(object):
    """One GPU + one stream (``spl_ctx``)."""

    def __init__(self, device=0, stream=None):
        self._h = ctypes.c_void_p()
        if stream is None:
            _check(lib().spl_create(ctypes.c_int(device), ctypes.byref(self._h)))
        else:
            _check(lib().spl_create_on_stream(ctypes.c_int(device), ctypes.c_void_p(stream), ctypes.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().spl_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- one-shot host-buffer calls ----------------------------------------------------------------
    def count(self, sites, reads, stranded=0, combine_mode=0, flags=0):
        """checkBam for all sites of a shard -> (beta1, beta2s_reads, dbl) uint32."""
        beta1 = np.zeros(max(sites.n, 1), np.uint32)
        b2s = np.zeros(max(sites.n, 1), np.uint32)
        dbl = np.zeros(max(sites.n_part, 1), np.uint32)
        opts = spl_opts(int(stranded), int(combine_mode), int(flags))
        _check(lib().spl_count(self._h, ctypes.byref(sites.c), ctypes.byref(reads.c), ctypes.byref(opts),
                               _ptr(beta1), _ptr(b2s), _ptr(dbl)))
        return beta1[:sites.n], b2s[:sites.n], dbl[:sites.n_part]

    def sse(self, sites, beta1, b2s_reads, dbl, cryptic):
        """findBeta2Counts + calculateSSE -> (beta2_simple, beta2_cryptic, beta2_weighted, sse)."""
        n = sites.n
        beta1, b2s_reads = _arr(beta1, np.uint32), _arr(b2s_reads, np.uint32)
        dbl = _arr(dbl if len(dbl) else np.zeros(1), np.uint32)
        o = [np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.float64),
             np.zeros(max(n, 1), np.float64)]
        _check(lib().spl_sse(self._h, ctypes.byref(sites.c), _ptr(beta1), _ptr(b2s_reads), _ptr(dbl),
                             ctypes.c_int(1 if cryptic else 0), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3])))
        return tuple(a[:n] for a in o)

    # -- device-resident pipeline --------------------------------------------------------------------
    def upload_sites(self, sites):
        h = ctypes.c_void_p()
        _check(lib().spl_sites_upload(self._h, ctypes.byref(sites.c), ctypes.byref(h)))
        return DeviceSites(self, h, sites.n, sites.n_part)

    def upload_reads(self, reads):
        h = ctypes.c_void_p()
        _check(lib().spl_reads_upload(self._h, ctypes.byref(reads.c), ctypes.byref(h)))
        return DeviceReads(self, h, reads.n)

    def begin_reads(self, expected_reads=0):
        """A read set to which segments are added one by one (``DeviceReads.add`` / ``add_bam``), then ``finish()``.
        ``expected_reads``: how many reads are going to be added, when known (sets of 64 M and more get larger chunks)."""
        h = ctypes.c_void_p()
        _check(lib().spl_reads_begin_sized(self._h, ctypes.c_int64(int(expected_reads)), ctypes.byref(h)))
        return DeviceReads(self, h, 0)

    def upload_read_segments(self, segments):
        """segments: [(ReadArrays-like with .c, position shift)] laid end to end as ONE device read set, copied straight
        from where they are (``spl_reads_upload_segments``)."""
        n = len(segments)
        segs = (spl_reads * max(n, 1))()
        shifts = (ctypes.c_int32 * max(n, 1))()
        total = 0
        for k, (reads, shift) in enumerate(segments):
            segs[k] = reads.c
            shifts[k] = int(shift)
            total += reads.n
        h = ctypes.c_void_p()
        _check(lib().spl_reads_upload_segments(self._h, ctypes.c_int(n), segs, shifts, ctypes.byref(h)))
        return DeviceReads(self, h, total)

    def upload_soa(self, segments, max_ends=None):
        """segments: [ReadArrays-like with .c] -> the BAM-native arrays as they are, laid end to end in device memory
        (``spl_soa_upload``): what a decode on the device leaves.  Read sets are laid out from them by the layout kernel
        (``DeviceReads.add_soa`` + ``finish``; ``relayout``).  ``max_ends``: per segment the last base its reads cover, if known."""
        n = len(segments)
        segs = (spl_reads * max(n, 1))()
        for k, reads in enumerate(segments):
            segs[k] = reads.c
        h = ctypes.c_void_p()
        ends = None if max_ends is None else (ctypes.c_int64 * max(n, 1))(*[int(v) if v is not None else -1 for v in max_ends])
        _check(lib().spl_soa_upload2(self._h, ctypes.c_int(n), segs, ends, ctypes.byref(h)))
        return DeviceSoA(self, h, [r.n for r in segments])

    def layout_read_segments(self, soa, shifts):
        """One read set from all segments of a ``DeviceSoA``, segment k moved by shifts[k]: the layout kernel's launch is
        queued, nothing is waited for."""
        dr = self.begin_reads(sum(soa.n))
        try:
            for k, shift in enumerate(shifts):
                dr.add_soa(soa, k, shift)
            return dr.finish()
        except Exception:
            dr.free()
            raise

    def layout_timing_collect(self, capacity=4096):
        """Durations (ms) of the layout kernel's launches since ``kernel_timing_begin`` (before ``kernel_timing_collect``)."""
        ms = (ctypes.c_float * capacity)()
        n = ctypes.c_int(0)
        _check(lib().spl_layout_timing_collect(self._h, ms, ctypes.c_int(capacity), ctypes.byref(n)))
        return [ms[i] for i in range(n.value)]

    def count_launch(self, dsites, dreads, stranded=0, combine_mode=0, flags=0):
        opts = spl_opts(int(stranded), int(combine_mode), int(flags))
        _check(lib().spl_count_launch(self._h, dsites._h, dreads._h, ctypes.byref(opts)))

    def sse_launch(self, dsites, cryptic):
        _check(lib().spl_sse_launch(self._h, dsites._h, ctypes.c_int(1 if cryptic else 0)))

    def sync(self):
        _check(lib().spl_sync(self._h))

    def pass_barrier(self):
        """Later launches start after all earlier ones (tails of counting passes included) are done; the host does not wait."""
        _check(lib().spl_pass_barrier(self._h))

    def timer_begin(self):
        _check(lib().spl_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float(0)
        _check(lib().spl_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    def kernel_timing_begin(self, max_records):
        _check(lib().spl_kernel_timing_begin(self._h, ctypes.c_int(max_records)))

    def kernel_timing_collect(self, capacity=4096):
        ms = (ctypes.c_float * capacity)()
        n = ctypes.c_int(0)
        _check(lib().spl_kernel_timing_collect(self._h, ms, ctypes.c_int(capacity), ctypes.byref(n)))
        return [ms[i] for i in range(n.value)]

    def launch_info(self):
        g, b, l = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(lib().spl_last_launch_info(self._h, ctypes.byref(g), ctypes.byref(b), ctypes.byref(l)))
        return dict(grid=g.value, block=b.value, lds_bytes=l.value)


class DeviceSites(object):
    def __init__(self, ctx, h, n, n_part):
        self.ctx, self._h, self.n, self.n_part = ctx, h, n, n_part

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def counters(self):
        beta1 = np.zeros(max(self.n, 1), np.uint32)
        b2s = np.zeros(max(self.n, 1), np.uint32)
        dbl = np.zeros(max(self.n_part, 1), np.uint32)
        _check(lib().spl_counters_download(self.ctx._h, self._h, _ptr(beta1), _ptr(b2s), _ptr(dbl)))
        return beta1[:self.n], b2s[:self.n], dbl[:self.n_part]

    def sse_results(self):
        n = self.n
        o = [np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.float64),
             np.zeros(max(n, 1), np.float64)]
        _check(lib().spl_sse_download(self.ctx._h, self._h, _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3])))
        return tuple(a[:n] for a in o)

    def free(self):
        if self._h:
            lib().spl_sites_free(self.ctx._h, self._h)
            self._h = ctypes.c_void_p()


class DeviceReads(object):
    def __init__(self, ctx, h, n):
        self.ctx, self._h, self.n = ctx, h, n

    def add(self, reads, shift=0, max_end=None):
        """One more segment from host arrays (``ReadArrays``), moved by ``shift`` into the shard's coordinate space.  ``max_end``:
        the last base the reads cover, if known (a decoder's arrays, a kept file's)."""
        _check(lib().spl_reads_add2(self.ctx._h, self._h, ctypes.byref(reads.c), ctypes.c_int32(int(shift)),
                                    ctypes.c_int64(-1 if max_end is None else int(max_end))))
        self.n += reads.n

    def add_bam(self, bam, chrom, shift=0):
        """One more segment: the reads of reference ``chrom`` straight from the decoder's buffers (waits for that reference
        to be complete; the rest of the file may still be decoding).  -> number of reads added."""
        tid = bam._tid[chrom]
        n_reads, _ = bam.wait_ref(chrom)
        _check(lib().spl_reads_add_bam(self.ctx._h, self._h, bam._h, ctypes.c_int(tid), ctypes.c_int32(int(shift))))
        self.n += n_reads
        return n_reads

    def add_bam_share(self, bam, share, chrom, shift=0):
        """One more segment: what share ``share`` of a decode in shares holds of reference ``chrom``, from that share's arrays on
        this device (``BamFile.decode_on_devices_async``; the decoders must be done: ``join_decoders``).  -> number of reads added."""
        n_reads, _ = bam.share_ref(share, chrom)
        if n_reads:
            _check(lib().spl_reads_add_bam_share(self.ctx._h, self._h, bam._h, ctypes.c_int(int(share)), ctypes.c_int(bam._tid[chrom]), ctypes.c_int32(int(shift))))
            self.n += n_reads
        return n_reads

    def add_soa(self, soa, seg, shift=0):
        """One more segment: segment ``seg`` of BAM-native arrays resident on the device (``Context.upload_soa``)."""
        _check(lib().spl_reads_add_soa(self.ctx._h, self._h, soa._h, ctypes.c_int(int(seg)), ctypes.c_int32(int(shift))))
        self.n += soa.n[seg]

    def finish(self):
        _check(lib().spl_reads_finish(self.ctx._h, self._h))
        return self

    def layout_bytes(self):
        """-> (bytes of BAM-native arrays the layout kernel reads, bytes of records it writes) for this read set."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().spl_reads_layout_bytes(self.ctx._h, self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def relayout(self):
        """The layout kernel once more, into the same records (segments that were laid out on the device)."""
        _check(lib().spl_reads_relayout(self.ctx._h, self._h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def literal_queue_size(self):
        out = ctypes.c_int64(0)
        _check(lib().spl_literal_queue_size(self.ctx._h, self._h, ctypes.byref(out)))
        return out.value

    def junctions(self, stranded=0, min_anchor=0, min_intron=0, max_intron=0):
        """Junction table of this read set, computed on the device (``spl_junctions``): dict of arrays left, right,
        strand (bytes '+', '-' or '?'), count, anchor_left, anchor_right, sorted by (left, right, strand)."""
        n = ctypes.c_int64(0)
        _check(lib().spl_junctions(self.ctx._h, self._h, ctypes.c_int(int(stranded)), ctypes.c_int32(int(min_anchor)),
                                   ctypes.c_int32(int(min_intron)), ctypes.c_int32(int(max_intron)), ctypes.byref(n)))
        n = n.value
        out = dict(left=np.empty(n, np.int32), right=np.empty(n, np.int32), strand=np.empty(n, np.uint8),
                   count=np.empty(n, np.uint32), anchor_left=np.empty(n, np.uint32), anchor_right=np.empty(n, np.uint32))
        _check(lib().spl_junctions_get(self.ctx._h, _ptr(out["left"]), _ptr(out["right"]), _ptr(out["strand"]), _ptr(out["count"]),
                                       _ptr(out["anchor_left"]), _ptr(out["anchor_right"])))
        return out

    def free(self):
        if self._h:
            lib().spl_reads_free(self.ctx._h, self._h)
            self._h = ctypes.c_void_p()


class DeviceSoA(object):
    """BAM-native reads (pos, flag, cig_off, cigar) resident in HBM; ``n`` = reads per segment."""
    def __init__(self, ctx, h, n):
        self.ctx, self._h, self.n = ctx, h, list(n)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def free(self):
        if self._h:
            lib().spl_soa_free(self.ctx._h, self._h)
            self._h = ctypes.c_void_p()


def algorithmic_bytes(dsites, dreads):
    out = ctypes.c_int64(0)
    _check(lib().spl_count_algorithmic_bytes(dsites._h, dreads._h, ctypes.byref(out)))
    return out.value


def gene_search(g_left, g_right, g_strand, q_pos, q_strand, is_stranded):
    """binary_gene_search for a batch (``spl_gene_search``): -> int32 gene index per query, -1 = none."""
    g_left = np.ascontiguousarray(g_left, np.int64)
    g_right = np.ascontiguousarray(g_right, np.int64)
    g_strand = np.ascontiguousarray(g_strand, np.uint8)
    q_pos = np.ascontiguousarray(q_pos, np.int64)
    q_strand = np.ascontiguousarray(q_strand, np.uint8)
    out = np.empty(q_pos.shape[0], np.int32)
    _check(lib().spl_gene_search(_ptr(g_left), _ptr(g_right), _ptr(g_strand), ctypes.c_int64(g_left.shape[0]), _ptr(q_pos),
                                 _ptr(q_strand), ctypes.c_int64(q_pos.shape[0]), ctypes.c_int(1 if is_stranded else 0), _ptr(out)))
    return out


def _blob(texts):
    """list of str -> (bytes, uint32 offsets[n + 1])."""
    enc = [t.encode("utf-8") for t in texts]
    off = np.zeros(len(enc) + 1, np.uint32)
    if enc:
        np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.int64, count=len(enc)), out=off[1:])
    return b"".join(enc), off


def tsv_prepare(arr):
    """What ``spl_tsv_append`` wants of a chromosome's rows that no count changes -- the strand and gene texts as blobs, the
    table's columns as the integers it reads -- made once and kept on ``arr``: ``process`` has this done while the alignment
    file is still being decoded, so that only the formatting itself is left when the counts arrive."""
    st = getattr(arr, "_tsv_static", None)
    if st is not None:
        return st
    if getattr(arr, "_strand_text", 0) is None:      # (a table built from arrays: the texts follow from the arrays, no list of str)
        codes = np.ascontiguousarray(arr.strand, np.uint8)
        there = codes != 0
        strand_blob = codes[there].tobytes()
        strand_off = np.zeros(arr.n + 1, np.uint32)
        np.cumsum(there, out=strand_off[1:])
    else:
        strand_blob, strand_off = _blob(arr.strand_text)
    if getattr(arr, "_genes", 0) is None and arr.gene_idx is not None:
        names_blob, names_off = _blob(list(arr.gene_names) + ["NA"])
        gi = np.where(arr.gene_idx >= 0, arr.gene_idx, len(arr.gene_names)).astype(np.int64)
        first = names_off[:-1].astype(np.int64)[gi]
        length = np.diff(names_off.astype(np.int64))[gi]
        gene_off = np.zeros(arr.n + 1, np.uint32)
        np.cumsum(length, out=gene_off[1:])
        take = np.repeat(first - gene_off[:-1].astype(np.int64), length) + np.arange(int(gene_off[-1]), dtype=np.int64)
        gene_blob = np.frombuffer(names_blob, np.uint8)[take].tobytes() if arr.n else b""
    else:
        gene_blob, gene_off = _blob(arr.genes)
    c64 = lambda a: np.ascontiguousarray(a, np.int64)   # noqa: E731
    st = dict(strand_blob=strand_blob, strand_off=strand_off, gene_blob=gene_blob, gene_off=gene_off, pos=c64(arr.pos), alpha=c64(arr.alpha),
              part_off=np.ascontiguousarray(arr.part_off, np.uint32), comp_off=np.ascontiguousarray(arr.comp_off, np.uint32),
              part_pos=c64(arr.part_pos), edge_cnt=c64(arr.edge_cnt), comp_pos=c64(arr.comp_pos), chrom=arr.chrom.encode("utf-8"))
    arr._tsv_static = st
    return st


class spl_tsv_rows(ctypes.Structure):
    _fields_ = [("chrom", ctypes.c_char_p), ("n_sites", ctypes.c_int64), ("pos", ctypes.c_void_p), ("strand_blob", ctypes.c_char_p),
                ("strand_off", ctypes.c_void_p), ("gene_blob", ctypes.c_char_p), ("gene_off", ctypes.c_void_p), ("sse", ctypes.c_void_p),
                ("alpha", ctypes.c_void_p), ("beta1", ctypes.c_void_p), ("beta2_simple", ctypes.c_void_p), ("beta2_cryptic", ctypes.c_void_p),
                ("beta2_weighted", ctypes.c_void_p), ("part_off", ctypes.c_void_p), ("part_pos", ctypes.c_void_p), ("edge_cnt", ctypes.c_void_p),
                ("comp_off", ctypes.c_void_p), ("comp_pos", ctypes.c_void_p)]


def tsv_append_many(path, items, cryptic):
    """Rows of several chromosomes, [(ChromArrays, results)], appended to ``path`` in that order by ``spl_tsv_append_many`` (same
    bytes as tsv.format_chrom for each)."""
    n = len(items)
    if n == 0:
        return
    rows = (spl_tsv_rows * n)()
    keep = []
    addr = lambda a: None if a is None else a.ctypes.data   # noqa: E731
    c64 = lambda a: np.ascontiguousarray(a, np.int64)       # noqa: E731
    for k, (arr, res) in enumerate(items):
        st = tsv_prepare(arr)
        beta1 = np.ascontiguousarray(res["beta1"], np.uint32)
        b2s = c64(res["beta2_simple"])
        sse = np.ascontiguousarray(res["sse"], np.float64)
        b2c = c64(res["beta2_cryptic"]) if cryptic else None
        b2w = np.ascontiguousarray(res["beta2_weighted"], np.float64) if cryptic else None
        keep.append((st, beta1, b2s, sse, b2c, b2w))
        r = rows[k]
        r.chrom, r.n_sites = st["chrom"], arr.n
        r.pos, r.strand_blob, r.strand_off = addr(st["pos"]), st["strand_blob"], addr(st["strand_off"])
        r.gene_blob, r.gene_off = st["gene_blob"], addr(st["gene_off"])
        r.sse, r.alpha, r.beta1, r.beta2_simple = addr(sse), addr(st["alpha"]), addr(beta1), addr(b2s)
        r.beta2_cryptic, r.beta2_weighted = addr(b2c), addr(b2w)
        r.part_off, r.part_pos, r.edge_cnt = addr(st["part_off"]), addr(st["part_pos"]), addr(st["edge_cnt"])
        r.comp_off, r.comp_pos = addr(st["comp_off"]), addr(st["comp_pos"])
    _check(lib().spl_tsv_append_many(os.fsencode(path), ctypes.c_int32(n), rows, ctypes.c_int(1 if cryptic else 0)))
    del keep


def tsv_append(path, arr, res, cryptic):
    """Rows of one chromosome appended to ``path`` (``spl_tsv_append_many`` with one; same bytes as tsv.format_chrom)."""
    tsv_append_many(path, [(arr, res)], cryptic)


def write_bam(path, ref_names, ref_lengths, read_sets, level=1, threads=0, seq_mode=0):
    """Fast native BAM writer for synthetic workloads: read_sets[i] = samio.ReadSet of reference i.  ``seq_mode`` 0: constant
    SEQ / QUAL bytes (deflates to a few bytes per record); 1: pseudo-random bases, binned qualities (deflates ~4x)."""
    n = len(ref_names)
    names = (ctypes.c_char_p * n)(*[s.encode("ascii") for s in ref_names])
    lens = (ctypes.c_int64 * n)(*[int(v) for v in ref_lengths])
    arr = (spl_reads * n)()
    keep = []
    for i, rs in enumerate(read_sets):
        ra = ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar)
        keep.append(ra)
        arr[i] = ra.c
    _check(lib().spl_bam_write2(os.fsencode(path), ctypes.c_int(n), names, lens, arr, ctypes.c_int(level), ctypes.c_int(threads),
                                ctypes.c_int(seq_mode)))


def trim(device=-1):
    """Device memory the library keeps for its next call goes back to the driver (``spl_trim``)."""
    _check(lib().spl_trim(ctypes.c_int(device)))


def prof_enable(on=True):
    """The library's own stopwatch over all its kernels, process-wide (``spl_prof_enable``): clears what was recorded."""
    _check(lib().spl_prof_enable(ctypes.c_int(1 if on else 0)))


def prof_report():
    """-> [{"kernel", "calls", "ms", "bytes"}, ...] of the launches since ``prof_enable`` (waits for the devices)."""
    import json
    need = lib().spl_prof_report(None, ctypes.c_int(0))
    buf = ctypes.create_string_buffer(need + 16)
    lib().spl_prof_report(buf, ctypes.c_int(need + 16))
    return json.loads(buf.value.decode("ascii"))


class BamFile(object):
    """BAM decode on host threads (``spl_bam_*``); replaces ``samtools view`` per site.

    ``stream=True``: the constructor returns once the header is read and the decode goes on in the background; ``wait_ref``
    blocks until one reference is complete, ``DeviceReads.add_bam`` sends its reads to the GPU from the decoder's own buffers,
    ``wait_all`` ends the decode and tells whether the file was sorted by reference (if not, references taken early were
    incomplete)."""

    def __init__(self, path, threads=0, stream=False, defer=False):
        self._h = ctypes.c_void_p()
        opener = lib().spl_bam_open_deferred if defer else (lib().spl_bam_open_stream if stream else lib().spl_bam_open)
        _check(opener(os.fsencode(path), ctypes.c_int(threads), ctypes.byref(self._h)))
        self.on_device = None    # (defer=True: set by decode_on_device)
        self.ref_names = [lib().spl_bam_ref_name(self._h, i).decode("ascii") for i in range(lib().spl_bam_n_ref(self._h))]
        self.ref_lengths = [lib().spl_bam_ref_length(self._h, i) for i in range(len(self.ref_names))]
        self._tid = {n: i for i, n in enumerate(self.ref_names)}
        self._views = {}

    def start_host_decode(self):
        """A file opened with ``defer=True``: decode on the host's threads, from now on in the background."""
        _check(lib().spl_bam_start(self._h))
        self.on_device = False

    def compression_ratio(self):
        """Inflated bytes per file byte over the first record blocks (0.0 = cannot tell); ``defer=True`` files only."""
        r = ctypes.c_double(0.0)
        _check(lib().spl_bam_compression_ratio(self._h, ctypes.byref(r)))
        return r.value

    def decode_on_device_async(self, device=0):
        """``decode_on_device`` on a thread and a context of its own, started now: the caller goes on (Steps 0-2), anybody who
        waits for a reference meanwhile waits for this decode.  -> the thread (join it before closing the file)."""
        import threading
        _check(lib().spl_bam_reserve_device(self._h))
        self.on_device = False

        def run():
            try:
                with Context(device) as ctx:
                    self.decode_on_device(ctx)
            except BaseException as exc:   # (the C side never leaves the file without a decoder; keep the reason)
                self.device_error = exc
                try:                       # (... and if it never got as far as the C side -- no context -- the reservation must
                    lib().spl_bam_start(self._h)   #  not outlive this thread: the host threads take the file)
                except Exception:
                    pass
        t = threading.Thread(target=run)
        t.start()
        self._device_thread = t
        return t

    def decode_on_devices_async(self, devices):
        """The decode in SHARES, one per entry of ``devices`` (a device may appear more than once: a context each): the file is
        cut into stretches of EQUAL size in file bytes, at any BGZF block (``spl_bam_share_plan``), and every device inflates and
        extracts its own stretch over its own PCIe link.  A reference may lie in several shares: each device counts its stretch
        against the reference's whole site table and the partial counters are added (``process.process_sites``).  -> [(device,
        [names of the references the share can hold records of])], the plan (``shares``; ``share_bytes``: the file bytes of each;
        ``share_ref`` says what a share really holds once the decoders are done).  Threads in ``_device_threads``."""
        import threading
        n = ctypes.c_int(0)
        _check(lib().spl_bam_share_plan(self._h, ctypes.c_int(len(devices)), ctypes.byref(n)))
        _check(lib().spl_bam_reserve_device(self._h))
        self.on_device = False
        plan, sizes = [], []
        for k in range(n.value):
            lo, hi = ctypes.c_int(0), ctypes.c_int(0)
            _check(lib().spl_bam_share_range(self._h, ctypes.c_int(k), ctypes.byref(lo), ctypes.byref(hi)))
            plan.append((devices[k], [self.ref_names[t] for t in range(lo.value, min(hi.value, len(self.ref_names)))]))
            sizes.append(self.share_info(k)["file_bytes"])
        self.shares = plan
        self.share_bytes = sizes
        self._share_errors = []

        def run(k, device):
            flag = ctypes.c_int(0)
            try:
                with Context(device) as ctx:
                    _check(lib().spl_bam_decode_device_share(ctx._h, self._h, ctypes.c_int(k), ctypes.byref(flag)))
            except BaseException as exc:   # (no context, or the call itself failed: the share must still be reported, or the
                self._share_errors.append(exc)   # file waits for ever -- the host threads take all of it then)
                try:
                    lib().spl_bam_start(self._h)
                except Exception:
                    pass
        self._device_threads = [threading.Thread(target=run, args=(k, dev)) for k, (dev, _) in enumerate(plan)]
        for t in self._device_threads:
            t.start()
        self._device_thread = self._device_threads[0]
        return plan

    def sample(self):
        """-> (records, CIGAR ops, inflated bytes) of what the host sampled at three places of the file (``spl_bam_sample``)."""
        out = (ctypes.c_int64 * 3)()
        _check(lib().spl_bam_sample(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def share_info(self, k):
        """-> {file_bytes, u_lo, u_hi, tail_blocks} of share ``k`` of the plan (``spl_bam_share_info``)."""
        v = [ctypes.c_int64(0) for _ in range(4)]
        _check(lib().spl_bam_share_info(self._h, ctypes.c_int(int(k)), *[ctypes.byref(x) for x in v]))
        return dict(file_bytes=v[0].value, u_lo=v[1].value, u_hi=v[2].value, tail_blocks=v[3].value)

    def share_ref(self, k, chrom):
        """-> (reads, largest end coordinate) of what share ``k`` holds of reference ``chrom`` (after ``join_decoders`` said True)."""
        n, me = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().spl_bam_share_ref(self._h, ctypes.c_int(int(k)), ctypes.c_int(self._tid[chrom]), ctypes.byref(n), ctypes.byref(me)))
        return n.value, me.value

    def share_count_host(self, k):
        """-> records of share ``k`` per reference by the host's inflate and a plain walk (the last entry: records without a
        reference); raises when the walk from the share's first record does not arrive at the next share's.  No GPU."""
        out = (ctypes.c_int64 * (len(self.ref_names) + 1))()
        _check(lib().spl_bam_share_count_host(self._h, ctypes.c_int(int(k)), out))
        return list(out)

    def decline_reason(self):
        """Why the device decoder left the file to the host threads ('' if it did not)."""
        return lib().spl_bam_decline_reason(self._h).decode("utf-8", "replace")

    def join_decoders(self):
        """Waits for the OUTCOME of the device decoders started by ``decode_on_device_async`` / ``decode_on_devices_async``; ->
        True when the reads are on the device(s) and every reference is complete, False when the host threads have the file (they
        may still be decoding it).  The decoders' threads themselves are joined by ``close``: what they give back on their way
        out (buffers, streams, events, their context: 10-13 ms for a large file) nobody has to wait for."""
        threads = getattr(self, "_device_threads", None) or ([self._device_thread] if getattr(self, "_device_thread", None) is not None else [])
        if not threads:
            return bool(getattr(self, "on_device", False))
        flag = ctypes.c_int(0)
        _check(lib().spl_bam_wait_device(self._h, ctypes.byref(flag)))
        self.on_device = bool(flag.value)
        return self.on_device

    def decode_on_device(self, ctx):
        """A file opened with ``defer=True``: inflate it and extract its records on the GPU of ``ctx`` (every reference is
        complete on return).  -> True; False when the file is not one for the device path (unsorted, CG-tag CIGARs, malformed)
        and the host threads have been started on it instead.  Without this call the first wait starts the host decode."""
        flag = ctypes.c_int(0)
        _check(lib().spl_bam_decode_device(ctx._h, self._h, ctypes.byref(flag)))
        self.on_device = bool(flag.value)
        return self.on_device

    @property
    def n_records(self):
        return lib().spl_bam_n_records(self._h)     # (waits for the end of the decode)

    def wait_ref(self, chrom):
        """-> (reads on that reference, largest end coordinate) once the reference is complete."""
        n, me = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().spl_bam_wait_ref(self._h, ctypes.c_int(self._tid[chrom]), ctypes.byref(n), ctypes.byref(me)))
        return n.value, me.value

    def wait_all(self):
        """Waits for the end of the decode (raises its error, if any).  -> True when the file was sorted by reference."""
        ok = ctypes.c_int(0)
        _check(lib().spl_bam_wait_all(self._h, ctypes.byref(ok)))
        return bool(ok.value)

    def reads(self, chrom):
        """-> samio.ReadSet-compatible views of the BAM-native arrays (borrowed from the native object, which every view
        keeps alive) or None when the file has no such reference (the reference's samtools call would print an error and
        yield nothing).  Waits for the whole file."""
        from .samio import ReadSet
        tid = self._tid.get(chrom)
        if tid is None:
            return None
        if chrom in self._views:
            return self._views[chrom]
        r = spl_reads()
        me = ctypes.c_int64(0)
        _check(lib().spl_bam_reads(self._h, ctypes.c_int(tid), ctypes.byref(r), ctypes.byref(me)))
        n = r.n_reads
        if n == 0:
            return ReadSet.empty()
        owner = _BamHandle(self)

        def view(ptr, count, dt):
            buf = (ctypes.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr)
            buf._owner = owner      # the array's base object keeps the native decoder alive
            return np.frombuffer(buf, dtype=dt, count=count)
        cig_off = view(r.cig_off, n + 1, np.uint32)
        n_cig = int(cig_off[-1])
        rs = ReadSet.__new__(ReadSet)
        rs.pos = view(r.pos, n, np.int32)
        rs.flag = view(r.flag, n, np.uint16)
        rs.cig_off = cig_off
        rs.cigar = view(r.cigar, n_cig, np.uint32) if n_cig else np.zeros(0, np.uint32)
        rs.max_end = me.value
        self._views[chrom] = rs
        return rs

    def close(self):
        """Closes the native decoder -- unless views of its arrays are still alive: then their owner closes it.  A decode that is
        still running is told to stop first (``spl_bam_cancel``): nobody who closes the file wants the rest of it."""
        if self._h:
            lib().spl_bam_cancel.restype = None
            lib().spl_bam_cancel(self._h)
        for t in (getattr(self, "_device_threads", None) or []) + ([self._device_thread] if getattr(self, "_device_thread", None) is not None else []):
            t.join()                 # (the device decoders work on the native object: not under their feet)
        self._device_thread = None
        self._device_threads = None
        h, self._h = self._h, ctypes.c_void_p()
        views, self._views = self._views, {}
        if h and not views:
            lib().spl_bam_close(h)
        elif h:
            _BamHandle.release(h, views)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BamHandle(object):
    """Shared ownership of a native decoder between a BamFile and the numpy views handed out from it."""
    _parked = {}

    def __init__(self, bam):
        self._bam = bam          # a view keeps its BamFile (and so the native object) alive

    @classmethod
    def release(cls, handle, views):
        # BamFile.close() with live views: the native object is closed when the last view is gone
        import weakref
        key = handle.value
        cls._parked[key] = handle
        left = [len(views)]

        def gone(_):
            left[0] -= 1
            if left[0] == 0:
                lib().spl_bam_close(cls._parked.pop(key))
        for rs in views.values():
            weakref.finalize(rs.pos, gone, None)


def pack_host(reads, threads=1):
    """The host packer alone (``spl_pack_host``): -> (chunk descriptors as a structured array, record blob uint8, wide ops)."""
    nc, rb, nw = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _check(lib().spl_pack_host(ctypes.byref(reads.c), ctypes.c_int(threads), ctypes.byref(nc), ctypes.byref(rb), ctypes.byref(nw),
                               None, None, None))
    desc = np.zeros(max(nc.value, 1), np.dtype([("rec_off", "<u8"), ("wide_off", "<u8"), ("first_pos", "<i4"), ("cost", "<u4"),
                                                 ("n", "<u2", (4,))]))
    rec = np.zeros(max(rb.value, 1), np.uint8)
    wide = np.zeros(max(nw.value, 1), np.uint32)
    _check(lib().spl_pack_host(ctypes.byref(reads.c), ctypes.c_int(threads), ctypes.byref(nc), ctypes.byref(rb), ctypes.byref(nw),
                               _ptr(desc), _ptr(rec), _ptr(wide)))
    return desc[:nc.value], rec[:rb.value], wide[:nw.value]


class TextColumns(object):
    """A BED12 junction file or the gene lines of an annotation as columns (``spl_bed_open`` / ``spl_gff_open``)."""
    __slots__ = ("chrom_names", "chrom", "left", "right", "alpha", "strand", "names")


def _text_columns(opener, path, with_alpha, with_names):
    h = ctypes.c_void_p()
    rc = opener(os.fsencode(path), ctypes.byref(h))
    if rc == -5:
        return None           # something the native reader will not vouch for: the caller reads line by line
    _check(rc)
    try:
        L = lib()
        rows = L.spl_text_rows(h)

        def col(ptr, dt):
            if not rows:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(rows,)).copy()
        out = TextColumns()
        out.chrom_names = [L.spl_text_chrom_name(h, k).decode("utf-8") for k in range(L.spl_text_n_chrom(h))]
        out.chrom = col(L.spl_text_chrom(h), np.int32)
        out.left = col(L.spl_text_i64(h, 0), np.int64)
        out.right = col(L.spl_text_i64(h, 1), np.int64)
        out.alpha = col(L.spl_text_i64(h, 2), np.int64) if with_alpha else None
        out.strand = col(L.spl_text_strand(h), np.uint8)
        out.names = None
        if with_names:
            off_p = ctypes.c_void_p()
            blob_p = L.spl_text_names(h, ctypes.byref(off_p))
            off = np.ctypeslib.as_array(ctypes.cast(off_p, ctypes.POINTER(ctypes.c_uint32)), shape=(rows + 1,)).tolist()
            blob = ctypes.string_at(blob_p, off[-1]) if rows and off[-1] else b""
            out.names = [blob[off[i]:off[i + 1]].decode("ascii") for i in range(rows)]
        return out
    except UnicodeDecodeError:
        return None
    finally:
        lib().spl_text_close(h)


def read_bed_columns(path):
    """-> TextColumns of the 12-column lines of a BED file, or None when the file must be read line by line."""
    return _text_columns(lib().spl_bed_open, path, True, False)


def read_gff_genes(path):
    """-> TextColumns of the ``gene`` lines of a GFF / GTF file, or None when the file must be read line by line."""
    return _text_columns(lib().spl_gff_open, path, False, True)


class spl_query_table(ctypes.Structure):
    _fields_ = [("chrom", ctypes.c_char_p), ("n", ctypes.c_int64), ("pos", ctypes.c_void_p), ("site", ctypes.c_void_p),
                ("strand", ctypes.c_void_p), ("part_off", ctypes.c_void_p), ("part_pos", ctypes.c_void_p),
                ("comp_off", ctypes.c_void_p), ("comp_pos", ctypes.c_void_p)]


def _view(addr, n, dt):
    """n items of dtype dt at address addr, copied out (the library owns the memory)."""
    if n == 0 or not addr:
        return np.zeros(0, dt)
    return np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()


def _ascii_gene(gene):
    """The native walk compares gene names as bytes of files it has checked to be ASCII: a name outside ASCII can match nothing there,
    and must not be turned into '?' (a gene may be called that) -- the Python walk takes it."""
    try:
        return str(gene).encode("ascii")
    except UnicodeEncodeError:
        raise SpliserNativeError(-5, "gene name %r is not ASCII: the Python walk of combine takes it" % (gene,))


class Combine(object):
    """The host walk of ``combine`` / ``combineShallow`` on columns (``spl_combine_*``, csrc/spl_combine.cpp): the per-sample
    .SpliSER.tsv files parsed and merged natively, the gap-fill queries as tables, the answers handed back, the .combined.tsv
    written.  Raises SpliserNativeError with code -5 for files the native parser does not take (spliser_amd/combine.py then
    walks them in Python)."""

    def __init__(self, tsv_paths):
        self._h = None
        self.n = len(tsv_paths)
        paths = (ctypes.c_char_p * self.n)(*[os.fsencode(p) for p in tsv_paths])
        h = ctypes.c_void_p()
        _check(lib().spl_combine_open(paths, ctypes.c_int32(self.n), ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().spl_combine_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def rows(self, idx):
        return int(lib().spl_combine_rows(self._h, ctypes.c_int32(idx)))

    def region_runs(self):
        """-> per sample the regions of its file in order, one entry per run of lines."""
        L = lib()
        out = []
        for idx in range(self.n):
            n = int(L.spl_combine_region_runs(self._h, ctypes.c_int32(idx), None, ctypes.c_int64(0)))
            ids = np.zeros(max(n, 1), np.int32)
            L.spl_combine_region_runs(self._h, ctypes.c_int32(idx), _ptr(ids), ctypes.c_int64(n))
            out.append([L.spl_combine_text(self._h, ctypes.c_int32(int(i))).decode("ascii") for i in ids[:n]])
        return out

    def keep_gene(self, gene):
        _check(lib().spl_combine_keep_gene(self._h, _ascii_gene(gene)))

    def merge(self, chroms, is_stranded, q_gene, shallow=None):
        """The lock-step walk.  -> [(position, samples with evidence)] of the sites combineShallow dropped."""
        names = (ctypes.c_char_p * max(len(chroms), 1))(*[c.encode("ascii", "replace") for c in chroms])
        ms, mr, me = (0, 0, 0.0) if shallow is None else shallow
        _check(lib().spl_combine_merge(self._h, names, ctypes.c_int32(len(chroms)), ctypes.c_int(1 if is_stranded else 0),
                                       _ascii_gene(q_gene), ctypes.c_int(0 if shallow is None else 1),
                                       ctypes.c_int64(int(ms)), ctypes.c_int64(int(mr)), ctypes.c_double(float(me))))
        p = ctypes.c_void_p()
        n = int(lib().spl_combine_skipped(self._h, ctypes.byref(p)))
        sk = _view(p.value, 2 * n, np.int64)
        return [(int(sk[2 * k]), int(sk[2 * k + 1])) for k in range(n)]

    @property
    def n_sites(self):
        return int(lib().spl_combine_n_sites(self._h))

    @property
    def n_gap_sites(self):
        return int(lib().spl_combine_n_gap_sites(self._h))

    def tables(self, idx):
        """The gap-fill queries of sample idx: [(region, dict of arrays: pos, site, strand, part_off, part_pos, comp_off, comp_pos)]
        in the order the walk met the regions, rows by position."""
        L = lib()
        out = []
        for k in range(int(L.spl_combine_n_tables(self._h, ctypes.c_int32(idx)))):
            t = spl_query_table()
            _check(L.spl_combine_table(self._h, ctypes.c_int32(idx), ctypes.c_int32(k), ctypes.byref(t)))
            n = int(t.n)
            part_off = _view(t.part_off, n + 1, np.uint32)
            comp_off = _view(t.comp_off, n + 1, np.uint32)
            out.append((t.chrom.decode("ascii"), dict(
                pos=_view(t.pos, n, np.int64), site=_view(t.site, n, np.int64), strand=_view(t.strand, n, np.uint8),
                part_off=part_off, part_pos=_view(t.part_pos, int(part_off[-1]), np.int64),
                comp_off=comp_off, comp_pos=_view(t.comp_pos, int(comp_off[-1]), np.int64))))
        return out

    def answers(self, idx, site, beta1, beta2_simple):
        site, beta1, b2 = _arr(site, np.int64), _arr(beta1, np.uint32), _arr(beta2_simple, np.uint32)
        _check(lib().spl_combine_answers(self._h, ctypes.c_int32(idx), ctypes.c_int64(site.shape[0]), _ptr(site), _ptr(beta1), _ptr(b2)))

    def write(self, path, titles, cryptic):
        t = (ctypes.c_char_p * self.n)(*[s.encode("utf-8") for s in titles])
        _check(lib().spl_combine_write(self._h, os.fsencode(path), t, ctypes.c_int(1 if cryptic else 0)))
