"""``process`` -- the SpliSER sub-command whose Step 3 is the MI355X hot path.

Same call signature, same stdout banners (loosely), same ``<outputPath>.SpliSER.tsv`` as
SpliSER_v0_1_8.py:695-720.  Steps 0-2 run on the host (``sites.py`` / ``fast_sites.py``) while the BAM file is decoded on the
GPU(s) -- every device the stretch of the file that holds its own chromosomes; Step 3 -- the per-site ``checkBam`` loop plus
``findBeta2Counts`` / ``calculateSSE`` (``processSites``, :681-692) -- is one counting pass (range, literal, scan + SSE
kernels) per shard on each GPU, over reads that never left it (``native.py`` -> libspliser_hip.so).

Deliberate deviations from the reference, all on the failure side:
  * an unreadable / truncated / non-BAM alignment file is an error here; the reference ignores samtools'
    exit status and silently reports zero beta counts (SpliSER_v0_1_8.py:422-427);
  * ``-g GENE`` with a gene that is not in the annotation is an error here; the reference crashes with
    AttributeError at :283.
"""
import os
import sys
import threading
import time

import numpy as np

from . import fast_sites, native, samio, shard, sites, tsv


def _log(msg):
    print(msg)
    sys.stdout.flush()


class _SamSource(object):
    """Reads from SAM text (small inputs / fixtures)."""

    def __init__(self, path):
        self.ref_names, self._sets = samio.read_sam(path)

    def reads(self, chrom):
        return self._sets.get(chrom)


def open_alignments(path, threads=0, stream=False, defer=False):
    """BAM (BGZF) through the native decoder; plain SAM text through the Python reader.  ``stream=True``: the BAM decoder
    returns after the header and decodes in the background (``native.BamFile``); ``defer=True``: nothing is decoded until
    somebody asks (``BamFile.decode_on_device``, or the first wait: host threads)."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:2] == b"\x1f\x8b":
        return native.BamFile(path, threads=threads, stream=stream, defer=defer)
    if magic[:1] == b"@" or b"\t" in open(path, "rb").readline():
        return _SamSource(path)
    raise native.SpliserNativeError(-5, "%s is neither BGZF/BAM nor SAM text" % path)


_closers = []


def wait_deferred_close():
    """``process`` hands the closing of its alignment file to a thread of its own and returns; this waits for all of them (a
    caller that times calls back to back, so that one call's unmapping does not run into the next one's opening).  Returns the
    seconds waited."""
    t0 = time.perf_counter()
    while _closers:
        _closers.pop().join()
    return time.perf_counter() - t0


def open_and_decode(path, devices, gpuDecode=None, threads=0):
    """The alignment file opened and its decode started: on the GPU(s) -- with several devices every one inflates and extracts
    the stretch of the file that holds its own references (``BamFile.decode_on_devices_async``), and counts them -- or, told so
    (``gpuDecode=False``), on host threads.  SAM text has one reader."""
    source = open_alignments(path, threads=threads, stream=True, defer=gpuDecode is not False)
    if isinstance(source, native.BamFile) and gpuDecode is not False:
        try:
            if len(devices) > 1:
                source.decode_on_devices_async(list(devices))
            else:
                source.decode_on_device_async(devices[0])
        except BaseException:       # (the share plan walks the whole directory and can fail on a damaged file: the mapping goes with it)
            source.close()
            raise
    return source


class _Replan(Exception):
    """A read reaches beyond the room its chromosome was given in the shard (planned from the BAM header)."""


def junction_consistency(arr, junctions, offset, is_stranded):
    """The junctions of the BED file (each partner edge of the table seen from its left end: position, partner position, alpha,
    SpliSER_v0_1_8.py:275-277, :341-355) against the junctions the reads themselves carry (``spl_junctions`` over the same read
    set: one count per N op, no anchor or intron-length filter).  -> list of (left, right, strand, BED alpha, reads in the BAM);
    regtools-style files filter by anchor length, so BED <= BAM is the normal case and BED > BAM means the two files do not
    belong together."""
    import numpy as np
    bam = {}
    for l, r, st, n in zip(junctions["left"].tolist(), junctions["right"].tolist(), junctions["strand"].tolist(), junctions["count"].tolist()):
        bam[(l - offset, r - offset, chr(st) if is_stranded else "?")] = n
    rows, seen = [], set()
    pos, off, ppos, cnt = arr.pos.tolist(), arr.part_off.tolist(), arr.part_pos.tolist(), arr.edge_cnt.tolist()
    for i in range(arr.n):
        st = arr.strand_text[i] if is_stranded else "?"
        for e in range(off[i], off[i + 1]):
            if ppos[e] > pos[i]:
                key = (pos[i], ppos[e], st)
                seen.add(key)
                rows.append(key + (cnt[e], bam.get(key, 0)))
    rows.extend(key + (0, n) for key, n in sorted(bam.items()) if key not in seen)
    return rows


def process_sites(table, source, q_chrom, is_stranded, stranded_type, is_beta2_cryptic, devices=(0,), combine_mode=0,
                  log=_log, timings=None, on_result=None, on_junctions=None, on_tables=None):
    """processSites (SpliSER_v0_1_8.py:681-692) for every chromosome.

    One thread and one context per device; the chromosomes of a device share ONE site table in one coordinate space
    (``shard.pack``) and are counted one after the other AS THEIR READS BECOME AVAILABLE: with a BAM file that is still being
    decoded a chromosome's reads go to the GPU (packed by the host, through the page-locked staging ring) as soon as the decoder
    has seen the first record of the next one, and its results are handed to ``on_result`` while later chromosomes decode.

    -> {chrom: (ChromArrays, results dict)} with results keys beta1, beta2_simple, beta2_cryptic,
    beta2_weighted, sse (numpy arrays in table row order).
    """
    stranded = native.STRANDED_CODE[stranded_type] if is_stranded else 0
    if is_stranded and stranded == 0:
        raise ValueError("strandedType must be 'fr' or 'rf' for a stranded analysis")
    is_bam = isinstance(source, native.BamFile)
    t_enter = time.perf_counter()
    items = {}
    for chrom in table.chrom_index:
        log("Processing region " + str(chrom))
        if not (q_chrom == chrom or q_chrom == "All"):
            continue
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        if is_bam:
            reads = None
            present = chrom in source._tid
        else:
            reads = source.reads(chrom)
            present = reads is not None
        if not present:
            log("  (no reference named %s in the alignment file: all beta counts are 0)" % chrom)
        items[chrom] = (arr, reads, present)
    if on_tables is not None:      # (whoever formats the rows later: these are the tables they will come from)
        on_tables([a for a, _, _ in items.values()])
    # Which device takes which chromosome is decided before the reads are known (they may still be on their way): the junction
    # read counts of the BED file say where the spliced reads are.
    weights = {c: (r.n if r is not None else int(a.alpha.sum())) + a.n for c, (a, r, _) in items.items()}
    shares = getattr(source, "shares", None) if is_bam else None
    listed = None    # the file is being decoded in shares: chromosome -> the shares (= positions in `devices`) that can hold reads of it
    if shares and len(devices) >= len(shares) and [d for d, _ in shares] == list(devices[:len(shares)]):
        # A share is a stretch of the FILE (spl_bam_share_plan: equal parts, cut at any BGZF block), so a chromosome's reads may lie
        # on several devices.  Each counts its stretch against the chromosome's whole site table; checkBam's counters only ever add
        # one per read (SpliSER_v0_1_8.py:519-559), so the partial beta1 / beta2Simple-reads / double-count arrays are added on the
        # host and findBeta2Counts + calculateSSE (:581-639) run once on the sums -- no read twice, nothing exchanged between devices.
        listed = {c: [k for k, (_, names) in enumerate(shares) if c in names] or [0] for c in items}
        plan = [[] for _ in devices]
        for c in items:
            for k in listed[c]:
                plan[k].append(c)
    else:
        plan = shard.assign(weights, len(devices))
    out, errors = {}, []
    lock = threading.Lock()
    partial = {}     # chromosome -> {share: (beta1, beta2s_reads, dbl)} of the devices that have counted their stretch of it

    def run(k_dev, device, chroms):
        try:
            if not chroms:
                return
            t0 = time.perf_counter()
            order = [c for c in table.chrom_index if c in chroms]
            if is_bam:      # the order of the file: a chromosome is complete when the next one begins
                order.sort(key=lambda c: source._tid.get(c, 1 << 30))
            for exact in ((False, True) if is_bam else (True,)):
                extents = None
                if is_bam and not exact:    # room by the header's reference lengths, and a little more: a read that hangs over
                    # the end of its reference (soft ends, simulated data) is no reason to plan again
                    extents = {c: (1, source.ref_lengths[source._tid[c]] + (1 << 20)) for c in order if c in source._tid}
                elif is_bam:                # ... or, should a read reach beyond that, by what the decoder has seen (whole file)
                    source.wait_all()
                    extents = {c: (1, max(1, source.wait_ref(c)[1])) for c in order if c in source._tid}
                shards = shard.pack([(c, items[c][0], items[c][1]) for c in order], concat_reads=False, extents=extents)
                if os.environ.get("SPL_PROCESS_TIMING"):
                    sys.stderr.write("[process] device %d: %s plan %.4f s after Step 3 began\n" % (device, "exact" if exact else "first", time.perf_counter() - t_enter))
                try:
                    _count_shards(k_dev, device, shards)
                    break
                except _Replan:
                    if exact:
                        raise
            if timings is not None:
                with lock:
                    timings["gpu_s"] = timings.get("gpu_s", 0.0) + time.perf_counter() - t0
        except Exception as exc:  # surfaced after join
            with lock:
                errors.append(exc)

    def _finish_split(ctx, chrom, pieces):
        """A chromosome whose reads were counted on several devices: the sums of their counters, then findBeta2Counts +
        calculateSSE on them (``spl_sse``, on the context of whichever device brought the last piece)."""
        arr = items[chrom][0]
        beta1, b2r, dbl = (sum(p[j].astype(np.int64) for p in pieces).astype(np.uint32) for j in range(3))
        b2s, b2c, b2w, sse = ctx.sse(native.SiteArrays.from_chrom(arr), beta1, b2r, dbl, is_beta2_cryptic)
        res = dict(beta1=beta1, beta2_simple=b2s, beta2_cryptic=b2c, beta2_weighted=b2w, sse=sse, beta2s_reads=b2r)
        with lock:
            out[chrom] = (arr, res)
        if on_result is not None:
            on_result(chrom, arr, res)

    def _count_shards(k_dev, device, shards):
        # the device decoder delivers every reference at once: nothing to stream chromosome by chromosome, so all chromosomes of
        # a shard are laid out as ONE read set and counted in one pass.  (Whether it did is asked when the first site table is
        # up: context and table take 10 ms that the decode's last kernels can run beside.)
        device_decode = is_bam and (getattr(source, "_device_thread", None) is not None or getattr(source, "_device_threads", None)) and on_junctions is None
        whole = None
        stamps = [] if os.environ.get("SPL_PROCESS_TIMING") else None   # (where a device thread's time goes, on stderr)

        def stamp(what):
            if stamps is not None:
                stamps.append((what, time.perf_counter()))
        stamp("start")
        with native.Context(device) as ctx:
            stamp("context")
            tables = []
            laid = {}      # shard index -> its read set, the layout kernels of its chromosomes on their stream
            try:
                # every shard's site table goes up first: building one is host work (the junction table) that fits beside the
                # decode, which is still running -- between two shards' counting passes it was 75 ms of an idle device
                for sh in shards:
                    tables.append(ctx.upload_sites(sh.sites))
                stamp("site tables up")
                def held(chrom):    # what this device holds of a chromosome: (reads, their largest end)
                    if not items[chrom][2]:
                        return 0, 0
                    return source.share_ref(k_dev, chrom) if listed is not None else source.wait_ref(chrom)

                def lay_out(k):
                    sh_k = shards[k]
                    dr_k = ctx.begin_reads(sum(held(c)[0] for c in sh_k.chroms))
                    laid[k] = dr_k
                    for chrom, off, limit in zip(sh_k.chroms, sh_k.offsets, sh_k.limits):
                        n_held, max_end = held(chrom)
                        if n_held:
                            if max_end > limit:
                                raise _Replan()
                            if listed is not None:
                                dr_k.add_bam_share(source, k_dev, chrom, off)
                            else:
                                dr_k.add_bam(source, chrom, off)
                    return dr_k
                for k_sh, (sh, ds) in enumerate(zip(shards, tables)):
                    if whole is None:
                        whole = bool(device_decode and source.join_decoders())
                        stamp("decoder joined")
                    if whole:
                        dr = laid[k_sh] if k_sh in laid else lay_out(k_sh)
                        stamp("reads laid out")
                        try:
                            dr.finish()
                            stamp("read set finished")
                            if k_sh + 1 < len(shards):
                                lay_out(k_sh + 1)  # (the next shard's segments are noted while this one is counted and brought down; since the fused
                                                   #  pass nothing is launched for them before their own finish())
                            ctx.count_launch(ds, dr, stranded, combine_mode)
                            ctx.sse_launch(ds, is_beta2_cryptic)
                            beta1, b2r, dbl = ds.counters()
                            b2s, b2c, b2w, sse = ds.sse_results()
                            stamp("counted, results down")
                        finally:
                            dr.free()
                            del laid[k_sh]
                        stamp("read set freed")
                        for chrom, (r0, r1), (e0, e1) in zip(sh.chroms, sh.site_rows, sh.edge_rows):
                            if listed is not None:
                                # who holds reads of this chromosome (every device thread asks the same finished decode: the same answer)
                                holders = [j for j in listed[chrom] if items[chrom][2] and source.share_ref(j, chrom)[0] > 0]
                                if len(holders) > 1:
                                    if k_dev not in holders:
                                        continue
                                    with lock:
                                        got = partial.setdefault(chrom, {})
                                        got[k_dev] = (beta1[r0:r1].copy(), b2r[r0:r1].copy(), dbl[e0:e1].copy())
                                        pieces = [got[j] for j in holders] if len(got) == len(holders) else None
                                    if pieces is not None:     # (the last piece: this thread adds them up)
                                        _finish_split(ctx, chrom, pieces)
                                    continue
                                if k_dev != (holders or listed[chrom])[0]:
                                    continue    # (listed here, but all its reads are on another device -- or it has none, and the first listed takes it)
                            res = dict(beta1=beta1[r0:r1].copy(), beta2_simple=b2s[r0:r1].copy(), beta2_cryptic=b2c[r0:r1].copy(),
                                       beta2_weighted=b2w[r0:r1].copy(), sse=sse[r0:r1].copy(), beta2s_reads=b2r[r0:r1].copy())
                            with lock:
                                out[chrom] = (items[chrom][0], res)
                            if on_result is not None:
                                on_result(chrom, items[chrom][0], res)
                        stamp("rows handed over")
                        if stamps is not None and sh is shards[-1]:
                            sys.stderr.write("[process] device %d: its shards %.4f s after Step 3 began, then %s\n" % (device, stamps[0][1] - t_enter, ", ".join(
                                "%s +%.4f" % (w, t - stamps[k - 1][1]) for k, (w, t) in enumerate(stamps) if k)))
                        continue
                    for chrom, off, limit, (r0, r1) in zip(sh.chroms, sh.offsets, sh.limits, sh.site_rows):
                        if listed is not None and listed[chrom][0] != k_dev:
                            continue    # (the host's threads decoded the file after all: a chromosome is whole, and the first device listed for it counts it)
                        with ctx.begin_reads() as dr:
                            if is_bam and items[chrom][2]:
                                _, max_end = source.wait_ref(chrom)
                                if max_end > limit:
                                    raise _Replan()
                                dr.add_bam(source, chrom, off)
                            elif items[chrom][1] is not None and items[chrom][1].n:
                                rs = items[chrom][1]
                                dr.add(native.ReadArrays(rs.pos, rs.flag, rs.cig_off, rs.cigar), off, getattr(rs, "max_end", None))
                            dr.finish()
                            ctx.count_launch(ds, dr, stranded, combine_mode)
                            ctx.sse_launch(ds, is_beta2_cryptic)
                            beta1, b2r, _ = ds.counters()
                            b2s, b2c, b2w, sse = ds.sse_results()
                            if on_junctions is not None and dr.n:
                                on_junctions(chrom, junction_consistency(items[chrom][0], dr.junctions(stranded), off, bool(stranded)))
                        res = dict(beta1=beta1[r0:r1].copy(), beta2_simple=b2s[r0:r1].copy(), beta2_cryptic=b2c[r0:r1].copy(),
                                   beta2_weighted=b2w[r0:r1].copy(), sse=sse[r0:r1].copy(), beta2s_reads=b2r[r0:r1].copy())
                        with lock:
                            out[chrom] = (items[chrom][0], res)
                        if on_result is not None:
                            on_result(chrom, items[chrom][0], res)
            finally:
                for dr_left in list(laid.values()):
                    dr_left.free()
                for ds in tables:
                    ds.free()

    threads = [threading.Thread(target=run, args=(k, dev, chroms)) for k, (dev, chroms) in enumerate(zip(devices, plan))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


class _TsvWriter(object):
    """outputBedFile (SpliSER_v0_1_8.py:641-664) while Step 3 is still running: chromosomes arrive in any order (one device
    thread each, the order of the BAM file), the file gets them in the table's order; the rows are formatted by the native
    library (tsv.format_chrom is the Python statement of the same format and what the CPU tests compare it with) on a thread
    of the writer's own -- the device threads hand a chromosome over and go on with the next one."""

    def __init__(self, output_path, table, cryptic):
        self.path = output_path + ".SpliSER.tsv"
        self.order = list(table.chrom_index)
        self.cryptic = cryptic
        self.ready = {}
        self.taken = set()     # chromosomes handed to the writing thread
        self.next = 0
        self.cond = threading.Condition()
        self.closing = False
        self.error = None
        self.seconds = 0.0
        self.prep = None
        with open(self.path, "w") as fh:
            fh.write(tsv.HEADER)
        self.thread = threading.Thread(target=self._run)
        self.thread.start()

    def expect(self, arrays):
        """The chromosomes' tables, known before their counts are: what the rows hold of them (texts, the table's own columns) is
        made ready now, on a thread of its own, beside the counting (``native.tsv_prepare``)."""
        def run():
            for arr in arrays:
                if self.closing:
                    return
                native.tsv_prepare(arr)
        self.prep = threading.Thread(target=run)
        self.prep.start()

    def add(self, chrom, arr, res):
        with self.cond:
            self.ready[chrom] = (arr, res)
            self.cond.notify()

    def _run(self):
        try:
            while True:
                with self.cond:
                    while True:
                        while self.next < len(self.order) and self.closing and self.order[self.next] not in self.ready:
                            self.next += 1      # (closing: a chromosome without a result has no rows)
                        if self.next >= len(self.order):
                            return
                        if self.order[self.next] in self.ready:
                            batch = []      # everything that is there and next in the table's order: formatted side by side
                            k = self.next
                            while k < len(self.order) and self.order[k] in self.ready:
                                batch.append(self.ready.pop(self.order[k]))
                                self.taken.add(self.order[k])
                                k += 1
                            break
                        self.cond.wait()
                t = time.perf_counter()
                native.tsv_append_many(self.path, batch, self.cryptic)   # (the file is appended to in order, by this thread only)
                self.seconds += time.perf_counter() - t
                self.next += len(batch)
        except BaseException as exc:   # surfaced by close()
            self.error = exc

    def close(self, expected):
        """Chromosomes that never got a result (no sites, other -c) are skipped; everything in ``expected`` must be there."""
        with self.cond:
            missing = [c for c in expected if c not in self.ready and c not in self.taken]
            self.closing = True
            self.cond.notify()
        self.thread.join()
        if self.prep is not None:
            self.prep.join()
        if self.error is not None:
            raise self.error
        if missing:
            raise RuntimeError("no results for %s" % missing)

    def abandon(self):
        """Stop without writing what is left (an error elsewhere)."""
        with self.cond:
            self.order = self.order[:self.next]
            self.closing = True
            self.cond.notify()
        self.thread.join()
        if self.prep is not None:
            self.prep.join()


def write_tsv(output_path, table, results, is_beta2_cryptic):
    """outputBedFile for a finished result set."""
    w = _TsvWriter(output_path, table, is_beta2_cryptic)
    for chrom, (arr, res) in results.items():
        w.add(chrom, arr, res)
    w.close(list(results))


def process(inBAM, inBed, outputPath, qGene="All", qChrom="All", maxIntronSize=0, annotationFile=None, aType="gene",
            isStranded=False, strandedType=None, isbeta2Cryptic=False, devices=(0,), threads=0, log=_log, checkJunctions=False,
            gpuDecode=None, keepReads=False):
    """SpliSER_v0_1_8.py:695-720, keyword-compatible with the reference's argparse dests.

    ``checkJunctions`` (this build only; changes no result): also derive every chromosome's junction table from the reads on
    the GPU and write ``<outputPath>.junctionCheck.tsv`` -- BED alpha against reads in the BAM per junction -- with a warning
    for every junction the BED file gives MORE reads than the BAM holds (the two files do not belong together).

    ``gpuDecode`` (this build only; changes no result): True = the BAM goes to the GPU as it is -- BGZF inflate, CRC32 and the
    extraction of POS / FLAG / CIGAR happen there (``spl_bam_decode_device``); files that path does not take (unsorted, CG-tag
    CIGARs, damaged) are decoded by the host threads all the same.  False = host threads.  None (default) = the GPU: with several
    devices every one decodes the stretch of the file that holds the references it then counts (``open_and_decode``).

    ``keepReads`` (this build only; changes no result): also leave ``<outputPath>.SpliSER.reads`` -- flag, POS and CIGAR of every
    placed record, all ``checkBam`` reads of an alignment -- which ``combine`` takes instead of decoding the BAM again, as long
    as it is still that BAM's (``readstore``)."""
    timings = {}
    t0 = time.perf_counter()
    # The alignment file does not depend on Steps 0-2: it is decoded on native threads while the site table is built here, and
    # goes on decoding while Step 3 counts the chromosomes that are complete.  An unreadable file is an error here already
    # (block directory and header are read by the opening call).
    source = open_and_decode(inBAM, devices, gpuDecode, threads)     # (the decode runs beside Steps 0-2, wherever it runs)
    keep = None      # (--keepReads: what the closing thread does first)
    try:
        t_open = time.perf_counter()
        table = _site_table(inBed, qGene, qChrom, maxIntronSize, annotationFile, aType, isStranded, strandedType, log)
        t1 = time.perf_counter()
        log("\n\nStep 3: Finding Beta reads")
        log("Processing sample 1 out of 1")
        writer = _TsvWriter(outputPath, table, isbeta2Cryptic)
        jrows = {}
        try:
            results = process_sites(table, source, qChrom, isStranded, strandedType, isbeta2Cryptic, devices=devices, log=log,
                                    timings=timings, on_result=writer.add, on_tables=writer.expect,
                                    on_junctions=(lambda c, rows: jrows.__setitem__(c, rows)) if checkJunctions else None)
            if isinstance(source, native.BamFile) and not source.wait_all():
                # records of an earlier reference after a later one: chromosomes were counted before they were complete.  (samtools
                # cannot index such a file, the reference could not have processed it at all.)  Everything is decoded by now: again.
                log("  (the alignment file is not sorted by reference: counting again from the complete decode)")
                writer.abandon()
                writer = _TsvWriter(outputPath, table, isbeta2Cryptic)
                jrows = {}
                results = process_sites(table, source, qChrom, isStranded, strandedType, isbeta2Cryptic, devices=devices, log=lambda m: None,
                                        timings=timings, on_result=writer.add, on_tables=writer.expect,
                                        on_junctions=(lambda c, rows: jrows.__setitem__(c, rows)) if checkJunctions else None)
        except BaseException:
            writer.abandon()
            raise
        if isinstance(source, native.BamFile) and gpuDecode is not False and source.decline_reason():
            # (said once, where the user reads it: the host's threads are several times slower than the device on files like this)
            log("  (the alignment file was decoded on host threads, not on the GPU: %s)" % source.decline_reason())
        t3 = time.perf_counter()
        log("\nOutputting .tsv file")
        writer.close(list(results))
        if checkJunctions:
            n_more = 0
            with open(outputPath + ".junctionCheck.tsv", "w") as fh:
                fh.write("Region\tLeft\tRight\tStrand\tBED_alpha\tBAM_reads\tStatus\n")
                for chrom in table.chrom_index:
                    for (l, r, st, a, b) in jrows.get(chrom, ()):
                        status = "equal" if a == b else ("bam_only" if a == 0 else ("bed_only" if b == 0 else ("bed<bam" if a < b else "bed>bam")))
                        n_more += a > b
                        fh.write("%s\t%d\t%d\t%s\t%d\t%d\t%s\n" % (chrom, l, r, st, a, b, status))
            if n_more:
                log("WARNING: %d junction(s) of %s carry more reads than %s holds for them -- do the two files belong together?" % (n_more, inBed, inBAM))
        if keepReads and isinstance(source, native.BamFile):
            # The kept reads -- every reference's arrays down from the device(s), then 19 bytes a read to the file -- are nothing the
            # .SpliSER.tsv waits for: they go out on the thread that closes the alignment file afterwards (wait_deferred_close waits
            # for both; `combine` does before it looks for such files; the interpreter does before it exits).  The file appears whole
            # or not at all (readstore.save writes beside its name and moves).
            from . import readstore

            def keep():
                t_keep = time.perf_counter()
                kept = [(name, source.reads(name)) for name in source.ref_names]
                t_down = time.perf_counter()
                readstore.save(outputPath + readstore.SUFFIX, inBAM, [(name, rs) for name, rs in kept if rs is not None])
                timings["keep_reads_s"] = time.perf_counter() - t_keep
                if os.environ.get("SPL_PROCESS_TIMING"):
                    sys.stderr.write("[process] kept reads: down from the device %.4f s, written %.4f s\n" % (t_down - t_keep, time.perf_counter() - t_down))
        t4 = time.perf_counter()
    except BaseException:
        keep = None
        raise
    finally:
        if hasattr(source, "close"):
            # (closing a decoded BAM gives gigabytes of read arrays back to the system -- a tenth of a second for 200 M reads --
            #  and nobody is waiting for that: on a thread of its own)
            def finish(keep=keep):
                try:
                    if keep is not None:
                        keep()
                except BaseException as exc:      # (said, not lost: the .SpliSER.tsv is written and right; `combine` decodes the BAM when the file is not there)
                    sys.stderr.write("process: the reads could not be kept (%s)\n" % exc)
                finally:
                    source.close()
            closer = threading.Thread(target=finish)
            closer.start()
            _closers[:] = [t for t in _closers if t.is_alive()] + [closer]
    timings["bam_decode"] = "device" if getattr(source, "on_device", False) else "host"
    if getattr(source, "share_bytes", None):     # (a decode in shares: the file bytes of each device's stretch -- the plan's balance)
        timings["share_bytes"] = [int(b) for b in source.share_bytes]
    timings.update(open_s=t_open - t0, site_table_s=t1 - t_open, step3_s=t3 - t1, write_tail_s=t4 - t3, write_s=writer.seconds,
                   close_s=time.perf_counter() - t4, total_s=time.perf_counter() - t0)
    if os.environ.get("SPL_PROCESS_TIMING"):
        sys.stderr.write("[process] %s\n" % ", ".join("%s %.4f" % (k, v) if isinstance(v, float) else "%s %s" % (k, v) for k, v in timings.items()))
    return timings


def _site_table(inBed, qGene, qChrom, maxIntronSize, annotationFile, aType, isStranded, strandedType, log):
    """Steps 0-2 (SpliSER_v0_1_8.py:700-712)."""
    log("Processing")
    log("Stranded Analysis {}".format(strandedType) if isStranded else "Unstranded Analysis")
    bins = sites.GeneBins()
    if annotationFile is not None:
        log("\n\nStep 0: Creating Genes from Annotation...")
        bins = sites.GeneBins.from_annotation(annotationFile, aType, qGene, log=log)
    log("\n\nPreparing Splice Site Arrays")
    log("\n\nStep 1: Finding Splice Sites / Counting Alpha reads...")
    log("Processing sample 1 out of 1")
    # one sample, no gene query, ordinary strands: the table follows from a few sorts (fast_sites.py, held to the
    # line-by-line builder by tests/test_fast_sites.py); anything else is built line by line
    table = fast_sites.build(bins, isStranded, inBed, q_chrom=qChrom, q_gene=qGene, max_intron=int(maxIntronSize))
    if table is None:
        table = sites.SiteTable(bins, is_stranded=isStranded)
        table.add_bed(inBed, q_chrom=qChrom, q_gene=qGene, max_intron=int(maxIntronSize))
    log("Sites assessed:\t" + str(table.assessed))
    log("Sites found:\t\t\t" + str(table.created))
    log("Sites assigned to a Gene:\t" + str(table.assigned))
    log("Sites:\t\t\t" + str(table.n_sites()))
    log("\n\nStep 2: Identifying Competitors of each splice site...")
    table.find_competitors()
    return table
