#!/usr/bin/env python3
"""bench.py -- throughput of the SpliSER `process` hot path (Step 3) on MI355X.

One "step" = ONE pass of the hot path over one synthetic sample whose reads are resident in HBM as the BAM-native arrays the
boundary takes and a decode on the device leaves -- pos, flag, cig_off, cigar: what checkBam reads from a SAM line
(SpliSER_v0_1_8.py:434-437), SURVEY.md 8(d)'s "device-resident SoA".  For every shard of the sample: the chunks' descriptors and
order (two small kernels over the CIGAR offsets), the FUSED range kernel (it reads the arrays itself: every CIGAR parsed once
into per-class records in LDS, the first half of the per-read walk, :436-512, and counted from there, the checkBam loop,
:408-559), the literal kernel and the scan with findBeta2Counts + calculateSSE (:581-639).  A step ends with a device-side
barrier: the next step's first kernel does not start before this step's last one has finished, as in a `process` run, which
counts a read set once (``--pipelined`` drops the barrier: many samples in a row).  After the timed region the same shards are
run the way the round timed them before the fused pass -- a LAYOUT kernel writing the records to memory, then the range kernel
over them (``roofline.two_kernels``; SPL_FUSED=0 makes that the step) -- and, as rounds 1-4 did, from the records on
(``roofline.count_only``).

Default workload = BASELINE.json configs[2], the largest single-GPU configuration ("synthetic human-scale: 200 M reads x 300 k
splice sites, HBM-roofline run"), synthesised from a seed (spliser_amd/synth.py) because there is no network and the reference
ships no data.

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards with no exchange step.  ``--scaling strong`` (the
default for N > 1): ONE sample, its chromosomes dealt to the ranks by read count; the line then reports the imbalance.
``--scaling weak`` (N = 1, or asked for): every rank processes its own sample of the same shape (seed + rank).  The only
collectives are the timing barrier and the reductions of the report.

Besides the resident-step figure the line carries (rank 0): the parity of the last timed step against the oracle on the whole
workload; a CPU baseline (the oracle on 1 thread and on all cores of this host; N = 1); and ``e2e`` -- the PRODUCT: BAM file ->
decode on the GPU(s) -> reads laid out -> kernels -> .SpliSER.tsv through spliser_amd.process, wall clock of the whole call, on
a file of constant SEQ / QUAL bytes and on one that deflates like a real library's, each with a table of the kernels' times from
the library's own events, the whole file text checked against the oracle, and the same work on the host's cores beside it
(``cpu_e2e``).  With N > 1 the legs call process(devices=range(N)): every GPU decodes and counts its own stretch of ONE file --
strong scaling of the product -- next to the same call on one GPU.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (before torch starts the HIP runtime: spl_create in csrc/spl_capi.cpp says why)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_inputs(args, rank, world):
    """-> (workload, table, items [(chrom, ChromArrays, ReadSet)], stranded).

    Strong scaling with several ranks: ONE sample, cut into `world` stretches of equal numbers of reads in file order, a chromosome
    cut anywhere (synth.strong_plan: what `process --gpus N` does to a BAM file); a rank generates the chromosomes its stretch
    touches and nothing else (the same reads the whole sample has there), builds the site tables of those, and its items are
    its PIECES: (chromosome, the chromosome's whole table, the rank's stretch of its reads).  args.strong_pieces then says which."""
    from spliser_amd import fast_sites, sites, synth
    cfg = synth.WORKLOADS[args.workload]
    stranded = args.stranded or ("fr" if cfg.get("paired") else None)
    seed = cfg["seed"] + (rank if args.scaling == "weak" else 0)
    pieces = None
    if args.scaling == "strong" and world > 1:
        n_genes = max(2, int(cfg["n_genes"] * (args.scale if args.scale < 1.0 else 1.0))) if args.genes is None else args.genes
        genome = synth.make_genome(cfg["chroms"], n_genes, cfg["intron"], seed=seed, alt_fraction=cfg.get("alt_fraction", 0.3) if args.alt_fraction is None else args.alt_fraction)
        expected = synth.expected_reads_per_chrom(genome, int(cfg["n_reads"] * args.scale))
        plan = synth.strong_plan(expected, world)
        pieces = plan[rank]
        wl = synth.Workload(args.workload, scale=args.scale, seed=seed, genome=genome, keep_chroms=sorted(set(c for c, _, _ in pieces)),
                            workers=max(1, min(32, (os.cpu_count() or 1) // max(world, 1))))
        args.strong_pieces = {"plan": plan, "expected_reads": [float(e) for e in expected]}
    cache = None
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)
        cache = os.path.join(args.cache, "%s_s%g_seed%d%s.npz" % (args.workload, args.scale, seed,
                                                                   "" if args.genes is None else "_g%d" % args.genes))
    if pieces is not None:
        pass
    elif cache and os.path.exists(cache):
        wl = synth.Workload.load(cache, args.workload)
    else:
        over = {} if args.alt_fraction is None else {"alt_fraction": args.alt_fraction}
        if args.genes is not None:
            over["n_genes"] = args.genes
        wl = synth.Workload(args.workload, scale=args.scale, seed=seed, **over,
                            workers=max(1, min(32, (os.cpu_count() or 1) // max(world, 1))))
        if cache:
            wl.save(cache)
    if args.soft_clips > 0:
        wl.reads = [synth.add_soft_clips(r, args.soft_clips, seed=100 + k) for k, r in enumerate(wl.reads)]
    tmp = tempfile.mkdtemp(prefix="spliser_bench_")
    bed = os.path.join(tmp, "junctions.bed")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
    table = fast_sites.build(sites.GeneBins(), bool(stranded), bed)   # the same table `process` builds (Steps 1-2)
    if table is None:
        table = sites.SiteTable(is_stranded=bool(stranded))
        table.add_bed(bed)
        table.find_competitors()
    shutil.rmtree(tmp, ignore_errors=True)
    items = []
    if pieces is not None:
        for ci, f0, f1 in pieces:
            c, rs = wl.genome.chrom_names[ci], wl.reads[ci]
            arr = table.chrom_arrays(c)
            if arr.n:
                items.append((c, arr, rs.take(int(f0 * rs.n), rs.n if f1 >= 1.0 else int(f1 * rs.n))))
        return wl, table, items, stranded
    for i, c in enumerate(wl.genome.chrom_names):
        arr = table.chrom_arrays(c)
        if arr.n:
            items.append((c, arr, wl.reads[i]))
    return wl, table, items, stranded


def lib_sha16():
    from spliser_amd import native
    h = hashlib.sha256()
    with open(native.LIB_PATH, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


KERNEL_SOURCES = ("spl_kernels.hip", "spl_device.h", "spl_pack.h", "spl_classify.h", "spl_pack.cpp", "spl_devpack.hip", "spl_devpack.h", "spl_layout_tile.h")


def kernel_src_sha16():
    """What the HBM traffic of the step's kernels depends on: the range kernel's sources and the layout kernel / packer that lays out what it reads,
    comments and white space aside (tools/traffic_json.py stamps a measurement with the same hash).  Host-side changes elsewhere
    in the library, and comments, leave it alone."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "spliser_amd", "csrc", name), "rb") as fh:
            h.update(strip_source(fh.read().decode("utf-8", "replace")).encode("utf-8"))
    return h.hexdigest()[:16]


INGEST_SOURCES = ("spl_inflate.hip", "spl_inflate_wave.h", "spl_wave.h", "spl_inflate.h", "spl_devpack.hip", "spl_devpack.h")


def ingest_src_sha16():
    """The sources of the device decode and the device packer (what the end-to-end legs run besides the counting kernels)."""
    h = hashlib.sha256()
    for name in INGEST_SOURCES:
        with open(os.path.join(ROOT, "spliser_amd", "csrc", name), "rb") as fh:
            h.update(strip_source(fh.read().decode("utf-8", "replace")).encode("utf-8"))
    return h.hexdigest()[:16]


def strip_source(text):
    """C / C++ source without comments, every run of white space one blank (string literals of these files hold no '//')."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def reference_measured():
    """What the REAL reference took in the build container (tools/time_reference.py, committed under profiles/): it cannot
    travel to the GPU box, so its `Total runtime (s)` is quoted from there -- indicative only: samtools is a SAM-text shim."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_time_reference.txt")))
    if not paths:
        return None
    try:
        with open(paths[-1]) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None
    return {"label": "INDICATIVE, measured in the build container (not on this box): the unmodified SpliSER_v0_1_8.py process, one "
                     "core; `samtools view` per site served by a SAM-text shim (oracle/refharness/samtools)",
            "from": os.path.basename(paths[-1]), "host": doc.get("host"),
            "cases": [{"name": c["name"], "reads": c["reads"],
                       "total_runtime_s_child_process_per_site": c["child process per site (samtools shim)"]["total_runtime_s"],
                       "total_runtime_s_in_process_replay": c["in-process replay"]["total_runtime_s"], "rows": c["in-process replay"]["rows"]}
                      for c in doc.get("cases", [])]}


def cpu_budget():
    """-> (hardware threads this process may run on, CPU-time quota of its cgroup in cores or None).  A box of this pool shows
    256 hardware threads and grants 16 cores' worth of CPU time: threads beyond the quota only get throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                 # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fh.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def run_oracle(items, scode, cryptic, threads):
    """The oracle over the whole workload.  -> (seconds, {chrom: (counters, sse results)})"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    outer = 1 if threads <= 1 else min(len(items), 8)
    oracle.set_threads(max(1, threads // outer))

    def one(item):
        c, arr, rd = item
        cnt = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                               rd.pos, rd.flag, rd.cig_off, rd.cigar, scode, 0)
        sse = oracle.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                               cnt[0], cnt[1], cnt[2], cryptic)
        return c, (cnt, sse)
    t = time.perf_counter()
    if outer == 1:
        res = dict(one(it) for it in items)
    else:  # (ctypes releases the GIL: chromosomes side by side, the site loop of each on its share of the cores)
        with ThreadPoolExecutor(outer) as pool:
            res = dict(pool.map(one, sorted(items, key=lambda it: -it[2].n)))
    dt = time.perf_counter() - t
    oracle.set_threads(1)
    return dt, res


def write_e2e_files(name, wl, stranded, seq_mode):
    """The BAM + BED + GFF files of one end-to-end leg in a directory of their own.  -> {"tmp", "prefix", "files_written_s"}"""
    from spliser_amd import native, synth
    tmp = tempfile.mkdtemp(prefix="spliser_e2e_")
    prefix = os.path.join(tmp, name)
    t = time.perf_counter()
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=bool(stranded))
    synth.write_gff(prefix + ".gff", wl.genome)
    # (seq_mode 2: the file `samtools sort` would leave -- htslib's default level 6, whole records per block, an aligner's fields)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=6 if seq_mode == 2 else 1, threads=0, seq_mode=seq_mode)
    return {"tmp": tmp, "prefix": prefix, "files_written_s": time.perf_counter() - t}


def cold_cli(files, stranded, cryptic, runs=2):
    """`python -m spliser_amd process ...` as a CHILD process, the way the reference is run (one command per interpreter,
    SpliSER_v0_1_8.py:1295-1361): interpreter start, imports, the library, the HIP context, the first call with nothing at hand,
    the .SpliSER.tsv.  The parent must not have touched the GPU yet (main() calls this before its first HIP call); the files are
    in the page cache (just written).  -> {"wall_s": [...], "out": path of the last run's file}"""
    import subprocess
    prefix = files["prefix"]
    walls, tails = [], []
    for k in range(runs):
        argv = [sys.executable, "-m", "spliser_amd", "process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-A", prefix + ".gff",
                "-o", prefix + ".cold%d" % k]
        if stranded:
            argv += ["--isStranded", "-s", stranded]
        if cryptic:
            argv += ["--beta2Cryptic"]
        t = time.perf_counter()
        r = subprocess.run(argv, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        walls.append(time.perf_counter() - t)
        tails.append(r.stdout[-300:])
        if r.returncode != 0:
            return {"wall_s": walls, "error": "exit %d: %s" % (r.returncode, r.stdout[-400:])}
    reported = [float(t.rsplit("Total runtime (s):", 1)[1].split()[0]) if "Total runtime (s):" in t else None for t in tails]
    return {"wall_s": walls, "reported_runtime_s": reported, "out": prefix + ".cold%d.SpliSER.tsv" % (runs - 1),
            "what": "wall clock of the child process around `python -m spliser_amd process -B -b -A -o` (fork to exit), started before "
                    "this process made its first HIP call; reported_runtime_s: the `Total runtime (s)` line the command prints, as the "
                    "reference does (SpliSER_v0_1_8.py:1360-1361)"}


def e2e_leg(name, wl, items, stranded, cryptic, seq_mode, reps, want, gpu_decode=None, devices=(0,), cpu_e2e=False, compare_devices=None,
            files=None, cold=None):
    """BAM + BED + GFF files of one workload -> spliser_amd.process.process (the CLI's function), timed.  The .SpliSER.tsv is
    compared, byte for byte, with the text the oracle's numbers give for the same sample (``want``: run_oracle's results).
    ``files``: written already (write_e2e_files); ``cold``: what cold_cli measured on them before the GPU was touched."""
    import statistics
    from spliser_amd import native, process, tsv
    if files is None:
        files = write_e2e_files(name, wl, stranded, seq_mode)
    tmp, prefix = files["tmp"], files["prefix"]
    out = {"workload": name}
    noop = lambda m: None   # noqa: E731
    try:
        out["files_written_s"] = files["files_written_s"]
        out["devices"] = list(devices)
        out["bam_decode_asked"] = {None: "default (on the GPU: every device its own stretch of the file)", False: "host threads (process --hostDecode)",
                                   True: "GPU (process --gpuDecode)"}[gpu_decode]
        out["bam_bytes"] = os.path.getsize(prefix + ".bam")
        out["bam_seq_qual"] = {0: "constant bytes (deflate to almost nothing)",
                               1: "pseudo-random bases, binned qualities in runs (deflate like a real library), zlib level 1, records cut at 0xff00",
                               2: "htslib-shaped: level 6, whole records per block, Illumina names, MAPQ / bin / mate fields, bases of a reference "
                                  "(overlapping reads share them), per-cycle NovaSeq quality bins, NH HI AS nM (XS:A, MD:Z) tags"}[seq_mode]
        out["bam"] = {0: "const", 1: "seq-like", 2: "htslib"}[seq_mode]      # (short forms for the compact line)
        n_reads = sum(r.n for r in wl.reads)

        def call(tag, devs):
            t0 = time.perf_counter()
            tm = process.process(prefix + ".bam", prefix + ".bed", prefix + ".out" + tag, annotationFile=prefix + ".gff",
                                 isStranded=bool(stranded), strandedType=stranded, isbeta2Cryptic=cryptic, log=noop,
                                 gpuDecode=gpu_decode, devices=tuple(devs))
            wall = time.perf_counter() - t0
            closing = process.wait_deferred_close()     # (the file's unmapping, on a thread of its own: not into the next call's opening)
            share = tm.pop("share_bytes", None)
            if share and len(devs) > 1:     # the plan of a decode in shares: how even the devices' stretches of the file are
                out["share_plan"] = {"shares": len(share), "file_bytes": share, "max_over_mean_bytes": max(share) / (sum(share) / len(share))}
            return dict(wall_s=wall, reads_per_sec=n_reads / wall, bam_decode=tm.pop("bam_decode", "host"),
                        stages={k2: round(v, 4) for k2, v in tm.items()}, deferred_close_s=round(closing, 4))
        import torch
        free_before = [torch.cuda.mem_get_info(d)[0] for d in devices]
        prof_first = bool(os.environ.get("SPL_BENCH_PROF_FIRST"))     # (diagnostic: the first call's kernels by the library's stopwatch)
        if prof_first:
            native.prof_enable(True)
        first = call("w", devices)     # the leg's first call: device memory for this file's sizes is not at hand yet (and, for the
        #                                first leg, nothing is): reported by itself, the timed calls are the ones behind it
        # (what the first call took from the driver and the process still holds -- live or in the library's pool for the next call)
        if prof_first:
            out["first_call_kernels"] = [{"kernel": k["kernel"], "calls": k["calls"], "ms": round(k["ms"], 2)} for k in native.prof_report()[:8]]
            native.prof_enable(False)
            sys.stderr.write("[bench] first call %.3f s, kernels: %s\n" % (first["wall_s"], out["first_call_kernels"]))
        out["first_call_device_gb"] = round(sum(b - torch.cuda.mem_get_info(d)[0] for b, d in zip(free_before, devices)) / 1e9, 2)
        runs = [call("%d" % k, devices) for k in range(reps)]
        best = min(runs, key=lambda r: r["wall_s"])
        med = statistics.median(r["wall_s"] for r in runs)
        nproc, quota = cpu_budget()
        out["host_cpu"] = {"nproc": nproc, "cpu_quota_cores": quota}
        out["decode"] = best["bam_decode"]
        out["bam_decode"] = {"host": "on host threads", "device": "on the GPU (BGZF inflate, CRC32, record extraction as kernels)"}[best["bam_decode"]]
        out.update(reads=n_reads, reads_per_sec=best["reads_per_sec"], wall_s=best["wall_s"], stages=best["stages"],
                   median_wall_s=med, median_reads_per_sec=n_reads / med, all_wall_s=[round(r["wall_s"], 4) for r in runs],
                   first_call_wall_s=first["wall_s"], runs=len(runs), deferred_close_s=best["deferred_close_s"],
                   what="process(): open BAM + BED/GFF -> Steps 0-2 on the host while the BAM decodes (GPU: the file's bytes up, Huffman "
                        "decoding, copies, CRC32, record extraction, reads laid out by kernels, a window of the stream at a time; host: "
                        "threads, packing, H2D through the staging ring) -> range + literal + scan/SSE kernels, D2H -> .SpliSER.tsv; wall "
                        "clock of the whole call in a process that has its GPU context and, from an earlier call, device memory of this file's "
                        "sizes at hand (first_call_wall_s: the leg's first call, which has neither yet and is not one of `runs`); "
                        "reads_per_sec / wall_s: the best of `runs` calls, median_*: their median; deferred_close_s: what the call left to a "
                        "thread of its own when it returned (the alignment file closed and unmapped), waited for before the next call")
        if compare_devices is not None:     # the same call on fewer devices: what the others bought
            one = [call("c%d" % k, compare_devices) for k in range(max(2, reps - 1))]
            b1 = min(r["wall_s"] for r in one)
            out["compared_with"] = {"devices": list(compare_devices), "wall_s": b1, "reads_per_sec": n_reads / b1,
                                    "median_wall_s": statistics.median(r["wall_s"] for r in one),
                                    "speedup_of_this_leg": b1 / best["wall_s"]}
        # one more call under the library's own stopwatch: where the device's time goes (never one of the timed calls)
        native.prof_enable(True)
        prof_run = call("p", devices)
        table = native.prof_report()
        native.prof_enable(False)
        for row in table:
            row["GBps"] = row["bytes"] / (row["ms"] * 1e-3) / 1e9 if row["ms"] > 0 else None
            row["frac_of_hbm_peak"] = row["GBps"] / HBM_PEAK_GBS if row["GBps"] is not None else None
        out["kernels"] = {"what": "every kernel launch of one more process() call between two events on its own stream (spl_prof_*): "
                                  "calls, summed ms, the bytes it was given to work on (algorithmic: its stretch of the inflated stream, "
                                  "its reads) and that as a rate against the 8 TB/s HBM peak; kernels of different streams overlap, "
                                  "so the ms do not add up to the call",
                          "wall_s_of_that_call": prof_run["wall_s"], "kernel_ms_sum": sum(r["ms"] for r in table), "table": table}
        # what the call as a whole moves, against the two things that can bound it: the link the file crosses (PCIe 5.0 x16:
        # 64 GB/s nominal; 56 GB/s is what single 32 MB copies reach on this stack, tools/pcie_rate.py) and the kernels
        inflated = next((r["bytes"] for r in table if r["kernel"] == "spl_crc32_kernel"), None)
        if best["bam_decode"] == "device" and inflated:
            file_rate = out["bam_bytes"] / best["wall_s"] / 1e9
            out["path"] = {"what": "the whole call on the bytes it has to move: the file over PCIe once (nothing else crosses it), its inflated "
                                   "stream through decode + copies + CRC32 + scan + extraction; a file that deflates 4x is bound by the "
                                   "link, one that deflates 49x by the copying kernel (DESIGN.md section 7)",
                           "file_GBps": file_rate, "pcie_peak_GBps": 56.0, "frac_of_pcie": file_rate / 56.0,
                           "inflated_bytes": inflated, "inflated_GBps": inflated / best["wall_s"] / 1e9}

        # parity of the FILE: its whole text against the rows the oracle's numbers give (same table, same format statement)
        tab = process._site_table(prefix + ".bed", "All", "All", 0, prefix + ".gff", "gene", bool(stranded), stranded, noop)
        text = [tsv.HEADER]
        for chrom in tab.chrom_index:
            if chrom not in want:
                continue
            (b1c, _, _), (b2s, b2c, b2w, sse) = want[chrom]
            text.extend(tsv.format_chrom(tab.chrom_arrays(chrom), dict(beta1=b1c, beta2_simple=b2s, beta2_cryptic=b2c, beta2_weighted=b2w, sse=sse), cryptic))
        expected = "".join(text)
        with open(prefix + ".out%d.SpliSER.tsv" % (reps - 1)) as fh:
            got = fh.read()
        out["tsv_rows"] = got.count("\n") - 1
        out["tsv_matches_oracle"] = bool(got == expected)
        out["tsv_compared"] = "the whole file, byte for byte, against tsv.format_chrom over the oracle's counters and doubles"
        if cold is not None:
            out["cold_cli"] = cold
            if "out" in cold:
                with open(cold["out"]) as fh:
                    out["cold_cli_matches"] = bool(fh.read() == expected)
                out["cold_cli_s"] = min(cold["wall_s"])
                out["cold_cli_first_s"] = cold["wall_s"][0]
        if cpu_e2e:     # the same work on the host's cores: decode (threads, libdeflate) + the oracle on all of them
            scode = native.STRANDED_CODE[stranded]
            n_threads = max(1, min(nproc, int(round(quota)) if quota else nproc))
            t0 = time.perf_counter()
            bam = native.BamFile(prefix + ".bam", threads=0)
            cpu_items = [(c, arr, bam.reads(c)) for c, arr, _ in items]
            t_dec = time.perf_counter() - t0
            t_or, got_cpu = run_oracle(cpu_items, scode, cryptic, n_threads)
            same = all(all(np.array_equal(x, y) for x, y in zip(got_cpu[c][0], want[c][0])) for c in want)
            del cpu_items
            bam.close()
            out["cpu_e2e"] = {"what": "the same file on the host alone: BGZF inflate + record extraction on the host's threads (libdeflate), then "
                                      "the oracle (oracle/spliser_oracle.c) on %d threads; no TSV" % n_threads,
                              "decode_s": t_dec, "oracle_s": t_or, "wall_s": t_dec + t_or, "reads_per_sec": n_reads / (t_dec + t_or),
                              "threads": n_threads, "same_counts": bool(same), "gpu_over_cpu": (t_dec + t_or) / best["wall_s"]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def make_combine_samples(args, n_samples=6):
    """BASELINE config 4: six samples of ONE A. thaliana-like genome (config 2's, seed 2), reads drawn with seeds 11-16, 15 % of
    the isoforms not expressed in a sample (its sample-specific gaps).  Generated before the GPU is touched (the generator forks)."""
    from spliser_amd import synth
    base = synth.make_genome(synth.WORKLOADS["arabidopsis"]["chroms"], synth.WORKLOADS["arabidopsis"]["n_genes"],
                             synth.WORKLOADS["arabidopsis"]["intron"], seed=synth.WORKLOADS["arabidopsis"]["seed"])
    return [synth.Workload("arabidopsis", scale=args.combine_scale, genome=base, read_seed=11 + k, silence=0.15,
                           workers=max(1, min(32, os.cpu_count() or 1))) for k in range(n_samples)]


def combine_leg(samples, threads):
    """config 4 end to end: `process` for each of the six samples (BAM -> .SpliSER.tsv), then `combine` over them (the files walked
    in lock-step, every site a sample does not list counted from that sample's BAM on the GPU, .combined.tsv).  The check: the
    same walk with the ORACLE answering the gap-fill queries from the decoded reads, its file byte for byte against the product's."""
    from oracle import oracle
    from spliser_amd import combine as cmb, native, process, synth
    tmp = tempfile.mkdtemp(prefix="spliser_cmb_")
    noop = lambda m: None   # noqa: E731
    out = {}
    try:
        t = time.perf_counter()
        lines, tsvs, bams, titles = [], [], [], []
        for k, wl in enumerate(samples):
            prefix = os.path.join(tmp, "s%d" % k)
            synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=False)
            native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=0, seq_mode=1)
            titles.append("S%d" % k)
            tsvs.append(prefix + ".SpliSER.tsv")
            bams.append(prefix + ".bam")
            lines.append("S%d\t%s\t%s\n" % (k, tsvs[-1], bams[-1]))
        sfile = os.path.join(tmp, "samples.tsv")
        with open(sfile, "w") as fh:
            fh.writelines(lines)
        t_files = time.perf_counter() - t
        n_reads = sum(sum(r.n for r in wl.reads) for wl in samples)
        t = time.perf_counter()
        per, kept_tm = [], []
        for k in range(len(samples)):
            t1 = time.perf_counter()
            tmk = process.process(bams[k], os.path.join(tmp, "s%d.bed" % k), os.path.join(tmp, "s%d" % k), log=noop, keepReads=True)
            per.append(time.perf_counter() - t1)      # (the call: its .SpliSER.tsv is there; its kept reads go out beside the next sample's call)
            kept_tm.append(tmk)
        process.wait_deferred_close()                 # ... and process_s ends when the last sample's are on the disk
        keep_s = sum(tmk.get("keep_reads_s", 0.0) for tmk in kept_tm)
        t_process = time.perf_counter() - t

        def timed_combine(out_name):
            best = None
            for rep in range(2):    # (the second call finds device memory of the samples' sizes at hand, like the e2e legs' timed calls)
                t = time.perf_counter()
                tm = cmb.combine(sfile, os.path.join(tmp, out_name), log=noop)
                wall = time.perf_counter() - t
                process.wait_deferred_close()
                if best is None or wall < best[0]:
                    best = (wall, tm)
                if rep == 0:
                    first = wall
            return best[0], best[1], first
        # `combine` as `process --keepReads` lets it run: every sample's flag / POS / CIGAR from the file process left (readstore:
        # only while it is that BAM's), no BAM decoded twice -- and, beside it, from the BAMs alone (what the reference's workflow
        # does, SpliSER_v0_1_8.py:903, and what this build does without --keepReads)
        wall, tm, first = timed_combine("all")
        os.environ["SPL_IGNORE_KEPT_READS"] = "1"
        try:
            wall_bams, tm_bams, _ = timed_combine("from_bams")
        finally:
            del os.environ["SPL_IGNORE_KEPT_READS"]
        with open(os.path.join(tmp, "all.combined.tsv"), "rb") as a, open(os.path.join(tmp, "from_bams.combined.tsv"), "rb") as b:
            kept_same = a.read() == b.read()
        # the checker: the same walk, the oracle's answers (test infrastructure; never timed as the product)
        t = time.perf_counter()
        oracle.set_threads(threads)
        n_q = 0
        with native.Combine(tsvs) as walk:
            walk.merge(cmb.region_order_from_runs(walk.region_runs()), False, "All", None)
            for idx in range(len(samples)):
                tabs = walk.tables(idx)
                if not tabs:
                    continue
                names = samples[idx].genome.chrom_names
                for chrom, tb in tabs:
                    rd = samples[idx].reads[names.index(chrom)]
                    b1, b2, _ = oracle.check_bam(tb["pos"], tb["strand"], tb["part_off"], tb["part_pos"], tb["comp_off"], tb["comp_pos"],
                                                 rd.pos, rd.flag, rd.cig_off, rd.cigar, 0, 1)
                    walk.answers(idx, tb["site"], b1, b2)
                    n_q += int(tb["pos"].shape[0])
            walk.write(os.path.join(tmp, "want.combined.tsv"), titles, False)
        oracle.set_threads(1)
        t_check = time.perf_counter() - t
        with open(os.path.join(tmp, "all.combined.tsv"), "rb") as a, open(os.path.join(tmp, "want.combined.tsv"), "rb") as b:
            same = a.read() == b.read()
        # ... and the native WALK itself (parsers, lock-step walk, writer: what the check above shares with the product) against the
        # Python statement of the same walk -- what the reference-generated goldens pin -- on the samples' rows of ONE chromosome
        # (the interpreter takes 50 s for all five), no gap fill on either side: the two files byte for byte
        t = time.perf_counter()
        small = min(samples[0].genome.chrom_names, key=lambda c: samples[0].genome.chrom_lengths[samples[0].genome.chrom_names.index(c)])
        cut = []
        for k, path in enumerate(tsvs):
            dst = os.path.join(tmp, "cut%d.tsv" % k)
            with open(path) as fh, open(dst, "w") as out_fh:
                for n_line, text in enumerate(fh):
                    if n_line == 0 or text.startswith(small + "\t"):
                        out_fh.write(text)
            cut.append(dst)
        rows = [cmb._parse_tsv(q) for q in cut]
        order = cmb.region_order(rows)
        merged = cmb.merge_sites(rows, order, len(titles), False, "All")
        cmb.write_combined(os.path.join(tmp, "cut_py.combined.tsv"), merged, titles, {}, False)
        with native.Combine(cut) as walk:
            walk.merge(cmb.region_order_from_runs(walk.region_runs()), False, "All", None)
            walk.write(os.path.join(tmp, "cut_native.combined.tsv"), titles, False)
        with open(os.path.join(tmp, "cut_py.combined.tsv"), "rb") as a, open(os.path.join(tmp, "cut_native.combined.tsv"), "rb") as b:
            walk_same = a.read() == b.read()
        t_walk_check = time.perf_counter() - t
        host = tm["parse_s"] + tm["merge_s"] + tm["write_s"]
        out = {"workload": "config 4: %d samples x %d reads (A. thaliana-like genome, seeds 11-%d, 15%% of the isoforms silent per sample)"
                           % (len(samples), n_reads // len(samples), 10 + len(samples)),
               "files_written_s": t_files, "process_s": t_process, "process_s_per_sample": [round(v, 4) for v in per],
               "combine_s": wall, "combine_first_call_s": first, "combine_from_bams_s": wall_bams, "gapfill_from_bams_s": tm_bams["gapfill_s"],
               "keep_reads_s": keep_s, "kept_reads_same_file": bool(kept_same),
               "walk": tm["walk"], "parse_s": tm["parse_s"], "merge_s": tm["merge_s"],
               "gapfill_s": tm["gapfill_s"], "write_s": tm["write_s"], "host_over_gpu": host / tm["gapfill_s"] if tm["gapfill_s"] else None,
               "sites": tm["sites"], "gap_sites": tm["gap_sites"], "queries": tm["queries"], "rows": tm["sites"] * len(samples),
               "combined_bytes": os.path.getsize(os.path.join(tmp, "all.combined.tsv")),
               "reads_per_sec": n_reads / (t_process + wall), "matches_oracle": bool(same), "queries_checked": n_q, "check_s": t_check,
               "walk_matches_python": bool(walk_same), "walk_check": "%s of the six files through combine.py's own parse / merge_sites / write_combined and through the native walk, no gap fill: %d sites, %.1f s" % (small, len(merged), t_walk_check),
               "what": "process --keepReads x 6 then combine, wall clock of the calls in this process (process_s includes writing the kept reads, keep_reads_s of it; "
                       "combine_s takes them, combine_from_bams_s decodes the six BAMs again); combine = parse the six .SpliSER.tsv (native, "
                       "a thread per file) + the lock-step walk (native) + gap fill (each sample's BAM decoded on the GPU once, its "
                       "queries through the range kernel in combine mode) + .combined.tsv (native); matches_oracle: the file against the "
                       "same walk with oracle/spliser_oracle.c answering every query from the reads that were written"}
        out["line"] = {k: out[k] for k in ("workload", "process_s", "combine_s", "parse_s", "merge_s", "gapfill_s", "write_s", "host_over_gpu",
                                           "queries", "rows", "reads_per_sec", "matches_oracle", "walk_matches_python", "combine_from_bams_s", "keep_reads_s",
                                           "kept_reads_same_file")}
        out["line"]["workload"] = "config4: %dx%dM reads" % (len(samples), n_reads // len(samples) // 1000000)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# ---- the line the driver parses ----------------------------------------------------------------------------------------------
LINE_LIMIT = 6144   # bytes; the driver keeps the last 8 KB of stdout and parses the LAST line: round 3's 23 KB line was cut


def _r(x, sig=5):
    """Numbers as the line carries them: floats to `sig` significant digits (ints, bools, None, strings untouched)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_e2e(leg):
    """One end-to-end leg as scalars only (the prose, the kernel tables and the stages stay in bench_detail.json)."""
    path = leg.get("path") or {}
    cpu = leg.get("cpu_e2e") or {}
    out = {"workload": leg.get("workload"), "bam": leg.get("bam"), "bam_bytes": leg.get("bam_bytes"),
           "decode": leg.get("decode"), "devices": len(leg.get("devices") or [0]) if len(leg.get("devices") or [0]) > 1 else None,
           "wall_s": leg.get("wall_s"), "median_wall_s": leg.get("median_wall_s"), "first_call_wall_s": leg.get("first_call_wall_s"), "first_call_device_gb": leg.get("first_call_device_gb"),
           "reads_per_sec": leg.get("reads_per_sec"), "file_GBps": path.get("file_GBps"), "frac_of_pcie": path.get("frac_of_pcie"),
           "tsv_matches_oracle": leg.get("tsv_matches_oracle"), "cpu_e2e_reads_per_sec": cpu.get("reads_per_sec")}
    for k in ("cold_cli_s", "cold_cli_matches"):
        if leg.get(k) is not None:
            out[k] = leg[k]
    cmp_ = leg.get("compared_with")
    if cmp_:
        out["one_device_wall_s"] = cmp_.get("wall_s")
    if leg.get("share_plan"):
        out["shares"], out["share_max_over_mean"] = leg["share_plan"]["shares"], leg["share_plan"]["max_over_mean_bytes"]
    return {k: v for k, v in out.items() if v is not None}


def compact_line(out, detail_name="bench_detail.json"):
    """The full result `out` -> the ONE line (< LINE_LIMIT bytes) that goes to stdout last: the contract's top-level keys,
    `roofline` and `cpu_baseline` as numbers, `parity`, one object of scalars per end-to-end leg.  Everything else is in
    `detail_name`.  Should a future field push the line past the limit, optional blocks are dropped from the back (never the
    contract's keys, `roofline` or `cpu_baseline`) until it fits: a line the driver can parse beats a complete one it cannot."""
    r = out.get("roofline") or {}
    cpu = out.get("cpu_baseline") or None
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "reads_per_sec", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "scale", "parallelism", "seed", "step") if k in cfg}
    line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg",
                                              "launches_timed", "algorithmic_bytes_per_launch")}
    lay, rng_, co = r.get("layout") or {}, r.get("range") or {}, r.get("count_only") or {}
    line["roofline"].update(layout={k: lay.get(k) for k in ("kernel_ms_avg", "frac", "arrays_read", "records_written")},
                            range={k: rng_.get(k) for k in ("kernel_ms_avg", "frac", "algorithmic_bytes_per_launch")},
                            range_alone_frac=(r.get("alone") or {}).get("frac"), path_frac=(r.get("path") or {}).get("frac"),
                            count_only={k: co.get(k) for k in ("ms_per_step", "reads_per_sec", "path_frac")},
                            range_traffic=r.get("range_traffic"), range_hbm_actual_frac=(r.get("hbm_actual") or {}).get("frac"),
                            traffic_from=r.get("traffic_from") or r.get("range_traffic_from"), fused=r.get("fused"),
                            two_kernels=(None if not r.get("two_kernels") else
                                         {k: r["two_kernels"].get(k) for k in ("ms_per_step", "layout_kernel_ms_avg", "range_kernel_ms_avg", "layout_frac", "range_frac",
                                                                                "path_frac", "same_counters")}),
                            lib_sha16=r.get("lib_sha16"), kernel_src_sha16=r.get("kernel_src_sha16"))
    if cpu:
        allc = cpu.get("all_cores") or {}
        line["cpu_baseline"] = {"value": cpu.get("value"), "unit": cpu.get("unit"), "reads_per_sec": cpu.get("reads_per_sec"),
                                "cores": cpu.get("cores"), "kind": cpu.get("kind"), "sample": cpu.get("sample_short") or cpu.get("sample"),
                                "all_cores": {"value": allc.get("value"), "reads_per_sec": allc.get("reads_per_sec"),
                                              "threads": allc.get("threads_used"), "seconds": allc.get("seconds")},
                                "reference_cost_model_s": (cpu.get("reference_cost_model") or {}).get("estimate_seconds")}
    else:
        line["cpu_baseline"] = None
    par = out.get("parity")
    line["parity"] = None if not par else {k: par.get(k) for k in ("bit_exact_vs_oracle", "reads", "sites")}
    line["e2e"] = None if out.get("e2e") is None else [compact_e2e(leg) for leg in out["e2e"]]
    for k in ("combine", "cold_cli"):
        if out.get(k) is not None:
            line[k] = out[k].get("line", out[k]) if isinstance(out[k], dict) else out[k]
    if out.get("other_steps"):
        line["other_steps"] = [{k: o.get(k) for k in ("workload", "reads", "ms_per_step", "kernel_ms_avg", "frac", "bit_exact_vs_oracle")} for o in out["other_steps"]]
    line["imbalance"] = out.get("imbalance")
    line["literal_kernel_reads"] = out.get("literal_kernel_reads")
    line["detail"] = detail_name
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    for victim in ("imbalance", "literal_kernel_reads", "cold_cli", "combine", "other_steps"):   # (never needed so far; see the docstring)
        if len(text) < LINE_LIMIT:
            break
        line.pop(victim, None)
        text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_LIMIT and line.get("e2e"):
        line["e2e"].pop()
        line["e2e_dropped_for_length"] = line.get("e2e_dropped_for_length", 0) + 1
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out):
    """Detail first (a file beside bench.py -- under gpurun_out/ too when that exists -- and an earlier stdout line that starts
    with a word, so that nothing mistakes it for THE line), then the compact line, last."""
    detail = json.dumps(out)
    name = os.environ.get("SPL_BENCH_DETAIL", "bench_detail.json")
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, os.path.basename(name)), "w") as fh:
                    fh.write(detail + "\n")
            except OSError:
                pass
    sys.stdout.write("bench_detail " + detail + "\n")
    sys.stdout.write(compact_line(out, os.path.basename(name)) + "\n")
    sys.stdout.flush()


def resident_step(ctx, items, stranded, cryptic, steps, warmup, threads):
    """The timed step of the line for ANOTHER sample, in the same process: the BAM-native arrays of `items` resident in HBM, per
    shard chunk map + order, the fused range kernel, literal, scan/SSE, a barrier between steps; the last step's counters and
    doubles against the oracle on the whole sample.  -> scalars for the line's `other_steps`."""
    from spliser_amd import native, shard
    scode = native.STRANDED_CODE[stranded]
    shards = shard.pack(items, concat_reads=False)
    dev = []
    for sh in shards:
        soa = ctx.upload_soa([rd for rd, _ in sh.read_segments])
        dev.append((ctx.upload_sites(sh.sites), ctx.layout_read_segments(soa, [shift for _, shift in sh.read_segments]), soa))
    ctx.sync()
    alg = sum(native.algorithmic_bytes(ds, dr) for ds, dr, _ in dev)
    fused = all(dr.layout_bytes()[1] == 0 for _, dr, _ in dev)

    def step():
        for ds, dr, _ in dev:
            dr.relayout()
            ctx.count_launch(ds, dr, scode, 0, 0)
            ctx.sse_launch(ds, cryptic)
        ctx.pass_barrier()
    for _ in range(warmup):
        step()
    ctx.sync()
    ctx.kernel_timing_begin(steps * len(dev))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    elapsed = (time.perf_counter() - t0) / steps
    kms = ctx.kernel_timing_collect(steps * len(dev) + 8)
    counts = [ds.counters() for ds, _, _ in dev]
    sses = [ds.sse_results() for ds, _, _ in dev]
    for ds, dr, soa in dev:
        dr.free()
        soa.free()
        ds.free()
    _, want = run_oracle(items, scode, cryptic, threads)
    exact = True
    for sh, cnts, sse in zip(shards, counts, sses):
        for chrom, (r0, r1), (e0, e1) in zip(sh.chroms, sh.site_rows, sh.edge_rows):
            (w1, w2, w3), wsse = want[chrom]
            exact &= np.array_equal(cnts[0][r0:r1], w1) and np.array_equal(cnts[1][r0:r1], w2) and np.array_equal(cnts[2][e0:e1], w3)
            exact &= all(np.array_equal(g[r0:r1], w) for g, w in zip(sse, wsse))
    n_reads = sum(rd.n for _, _, rd in items)
    k_avg = float(np.mean(kms)) if kms else float("nan")
    per_launch = alg / max(len(dev), 1)
    return {"reads": n_reads, "sites": sum(arr.n for _, arr, _ in items), "shards": len(dev), "stranded": stranded, "cryptic": bool(cryptic), "fused": bool(fused),
            "ms_per_step": elapsed * 1e3, "reads_per_sec": n_reads / elapsed, "kernel_ms_avg": k_avg,
            "frac": per_launch / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBS if kms else None, "path_frac": alg / elapsed / 1e9 / HBM_PEAK_GBS,
            "bit_exact_vs_oracle": bool(exact)}


def rank_pieces(items, world, rank, expected=None):
    """The pieces rank `rank` of `world` works on in a strong-scaling run, from the WHOLE sample's items [(chromosome, table, reads)]
    in file order: the sample's reads cut into `world` stretches of equal numbers (synth.strong_plan over `expected` reads per
    chromosome -- the real numbers when none are given), a chromosome cut anywhere -> [(chromosome, the chromosome's whole table,
    the rank's stretch of its reads)].  Every read is in exactly one rank's pieces; what a rank counts are PARTIAL counters of its
    chromosomes (checkBam only ever adds one per read, SpliSER_v0_1_8.py:519-559: the ranks' counters add up to the sample's).
    main() does the same without ever holding the whole sample (build_inputs); this is the statement the CPU tests hold."""
    from spliser_amd import synth
    exp = [float(rd.n) for _, _, rd in items] if expected is None else list(expected)
    out = []
    for ci, f0, f1 in synth.strong_plan(exp, world)[rank]:
        c, arr, rd = items[ci]
        out.append((c, arr, rd.take(int(f0 * rd.n), rd.n if f1 >= 1.0 else int(f1 * rd.n))))
    return out


def reduce_report(dist, rank, world, n_reads, n_sites, elapsed, device, exact=None):
    """The report's only collectives: MAX of the ranks' times, SUM of their units, the per-rank table (imbalance) -- and, where
    every rank has checked its own counters against the oracle (`exact`), whether ALL did (imbalance["all_ranks_exact"]).
    -> (elapsed over ranks, reads of all ranks, sites of all ranks, imbalance)."""
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = None
    if exact is not None:
        e = torch.tensor([1.0 if exact else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(e, op=dist.ReduceOp.MIN)
        ok = bool(e.item() > 0.5)
    c = torch.tensor([float(n_reads), float(n_sites)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    per = torch.zeros(world, 2, dtype=torch.float64, device=device)
    per[rank, 0], per[rank, 1] = float(n_reads), float(elapsed)
    dist.all_reduce(per, op=dist.ReduceOp.SUM)
    imbalance = {"reads_per_rank": [int(v) for v in per[:, 0].tolist()],
                 "seconds_per_rank": [round(float(v), 6) for v in per[:, 1].tolist()],
                 "max_over_mean_reads": float(per[:, 0].max() / per[:, 0].mean()) if float(per[:, 0].sum()) > 0 else None}
    if ok is not None:
        imbalance["all_ranks_exact"] = ok
    return float(t.item()), float(c[0].item()), float(c[1].item()), imbalance


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1) before this process has made any HIP call -- a process that has touched the GPU must
    not be replaced or forked --, pass its output through (its last line is THE line) and leave with its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="human", choices=["arabidopsis", "human", "mouse_stranded", "single_gene"])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the workload's read count (debug)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default) = one sample, chromosomes dealt to the ranks; weak = a sample per rank")
    ap.add_argument("--share-devices", action="store_true", help="N > 1 on a box with fewer than N GPUs: rank r on device r mod the "
                    "number there are, the report's reductions over gloo (runs the multi-rank path; its numbers are not a measurement)")
    ap.add_argument("--stranded", default=None, choices=[None, "fr", "rf"])
    ap.add_argument("--beta2Cryptic", action="store_true")
    ap.add_argument("--pipelined", action="store_true", help="no barrier between steps: the tail of a step runs beside the "
                    "first range kernel of the next (many samples in a row)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", default="auto", choices=["auto", "off"], help="auto: the decode-inclusive leg for this workload "
                    "(and for arabidopsis when the workload is human)")
    ap.add_argument("--e2e-seq-mode", type=int, default=None, choices=[0, 1, 2], help="the e2e BAMs: 0 constant SEQ / QUAL bytes, "
                    "1 pseudo-random bases + binned qualities (10x the file), 2 what an aligner + samtools would write; default: all three")
    ap.add_argument("--e2e-reps", type=int, default=3)
    ap.add_argument("--no-small-leg", action="store_true", help="no A. thaliana end-to-end legs beside the human-scale ones")
    ap.add_argument("--combine", default="auto", choices=["auto", "on", "off"], help="auto: the config-4 leg (six samples, process x 6 + combine) "
                    "with the default workload at N = 1; on: with any workload (N = 1)")
    ap.add_argument("--combine-scale", type=float, default=1.0, help="fraction of 20 M reads per combine sample (debug)")
    ap.add_argument("--no-other-steps", action="store_true", help="no resident steps of the other configurations (mouse paired fr + cryptic, A. thaliana, "
                    "this workload with soft clips) behind the timed one")
    ap.add_argument("--no-cold-cli", action="store_true", help="no `python -m spliser_amd process` child processes before the GPU is touched")
    ap.add_argument("--kernel", default="ranges", choices=["ranges", "ranges_agg", "pairs"],
                    help="ranges = default product path; pairs = the literal per-(read, site) kernel")
    ap.add_argument("--alt-fraction", type=float, default=None, help="(experiment) fraction of genes with alternative isoforms")
    ap.add_argument("--genes", type=int, default=None, help="(experiment) number of genes the reads are spread over: fewer genes = "
                    "deeper coverage per site")
    ap.add_argument("--soft-clips", type=float, default=0.0, help="(experiment) fraction of reads that get leading / trailing "
                    "soft clips, as a local aligner reports them")
    ap.add_argument("--cache", default=None, help="directory to cache the generated sample in (.npz); a cached "
                    "sample is loaded instead of regenerated (use under rocprofv3: no generator worker processes)")
    ap.add_argument("--copies", type=int, default=1, help="(experiment) device copies of the sample; step k works on copy "
                    "k mod copies, so that nothing a step read can still be in a cache when it is read again")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                 "bench.py --gpus %d ...), or run `python bench.py --gpus %d` alone and it starts its ranks itself"
                 % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"

    # torch first: it brings a HIP runtime of its own, and whichever libamdhip64 a process loads first is the one everything
    # else in it gets -- libspliser_hip.so (loaded by the host helpers below, long before a GPU is touched) would otherwise pull
    # in the system's, and torch's other libraries do not work with that one.  (Importing does not initialise the GPU.)
    import torch
    from spliser_amd import native, shard, synth

    # ---- synthetic sample of this rank (generated BEFORE the GPU is touched: the generator forks) -----
    t_gen = time.perf_counter()
    wl, table, items, stranded = build_inputs(args, rank, world)   # (strong scaling, N > 1: the rank's own stretch of ONE sample, and nothing else of it)
    strong = getattr(args, "strong_pieces", None)
    # N > 1: rank 0's end-to-end legs (process(devices = all the node's GPUs) on ONE file, after the collectives) have a sample of
    # their own -- config 2's shape, five chromosomes: what whole-reference shares could give five GPUs at most
    wl_e2e = None
    if args.e2e == "auto" and rank == 0 and world > 1:
        small = argparse.Namespace(**vars(args))
        small.workload, small.cache, small.scaling = "arabidopsis", None, "weak"
        wl_e2e = build_inputs(small, 0, 1)
    # the e2e leg of the smaller configuration wants a sample of its own: generated now, for the same reason
    wl_small = None
    if args.e2e == "auto" and rank == 0 and world == 1 and args.workload == "human" and args.scale == 1.0 and not args.no_small_leg:
        small = argparse.Namespace(**vars(args))
        small.workload, small.cache = "arabidopsis", None
        wl_small = build_inputs(small, 0, 1)
    full_default = args.e2e == "auto" and rank == 0 and world == 1 and args.workload == "human" and args.scale == 1.0
    # ... and so do the OTHER configurations whose resident step rides in the line (`other_steps`): config 5's shape (mouse, paired
    # flags, `fr` + cryptic: the STRANDED fused instantiation), config 2 (the small leg's sample, above), and this workload with a
    # local aligner's soft clips on three reads in ten
    others = []
    wl_mouse_e2e = None
    if rank == 0 and world == 1 and args.workload == "human" and not args.no_other_steps and args.kernel == "ranges" and args.soft_clips == 0:
        ms = argparse.Namespace(**vars(args))
        ms.workload, ms.cache, ms.stranded = "mouse_stranded", None, None
        wl_m, _, items_m, stranded_m = build_inputs(ms, 0, 1)
        others.append(("mouse_stranded fr+cryptic", items_m, stranded_m, True))
        del wl_m
        if full_default and args.e2e_seq_mode is None:     # ... and a quarter of it as a FILE: paired flags through the device decoder, `fr` + cryptic end to end
            ms.scale = 0.25
            wl_mouse_e2e = build_inputs(ms, 0, 1)
        if wl_small is not None:
            others.append(("arabidopsis", wl_small[2], wl_small[3], args.beta2Cryptic))
        from spliser_amd import synth as _synth
        clipped = [(c, arr, _synth.add_soft_clips(rd, 0.3, seed=100 + k)) for k, (c, arr, rd) in enumerate(items)]
        others.append(("%s clipped30" % args.workload, clipped, stranded, args.beta2Cryptic))
    cmb_samples = make_combine_samples(args) if ((full_default and args.combine == "auto") or (args.combine == "on" and rank == 0 and world == 1)) else None
    shards = shard.pack(items, concat_reads=False)
    n_reads = sum(rd.n for _, _, rd in items)
    n_sites = sum(arr.n for _, arr, _ in items)
    if strong is not None:     # (a cut chromosome's sites are counted against by several ranks: each its share of them, so that the ranks' sums are the sample's)
        share = {wl.genome.chrom_names[ci]: (f1 - f0) for ci, f0, f1 in strong["plan"][rank]}
        n_sites = sum(arr.n * share.get(c, 1.0) for c, arr, _ in items)
    t_gen = time.perf_counter() - t_gen

    # ---- the command line as users run it, BEFORE this process touches the GPU: the files of the sequence-like legs are written now
    # (host code) and `python -m spliser_amd process` runs on them as a child, a fresh interpreter each (cold_cli)
    pre_files, cold = {}, {}
    if args.e2e == "auto" and rank == 0 and world == 1 and not args.no_cold_cli and (args.e2e_seq_mode in (None, 1)):
        pre_files[args.workload] = write_e2e_files(args.workload, wl, stranded, 1)
        cold[args.workload] = cold_cli(pre_files[args.workload], stranded, args.beta2Cryptic)
        if wl_small is not None:
            pre_files["arabidopsis"] = write_e2e_files("arabidopsis", wl_small[0], wl_small[3], 1)
            cold["arabidopsis"] = cold_cli(pre_files["arabidopsis"], wl_small[3], args.beta2Cryptic)

    dist = None
    # One rank per GPU.  A box with fewer GPUs than ranks is refused, unless --share-devices asks for the ranks to take turns on
    # what is there (rank r on device r mod the number of devices): the N > 1 path of this file run end to end on a one-GPU box;
    # RCCL does not take two ranks on one device, so the report's reductions then go over gloo on host tensors.
    n_dev = torch.cuda.device_count()    # (counts devices without initialising the GPU)
    shared = False
    if world > n_dev:
        if not args.share_devices or n_dev < 1:
            sys.exit("bench.py: --gpus %d, but %d GPU(s) visible here: one rank per GPU is what is measured (--share-devices lets the "
                     "ranks share the devices there are: a test of the path, not a measurement)" % (world, n_dev))
        shared = True
    device_id = local_rank % max(n_dev, 1)
    torch.cuda.set_device(device_id)
    red_device = "cpu" if shared else "cuda"
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_id))

    scode = native.STRANDED_CODE[stranded]
    kflags = {"pairs": native.OPT_PAIR_KERNEL, "ranges_agg": native.OPT_WAVE_AGGREGATION, "ranges": 0}[args.kernel]
    ctx = native.Context(device_id)
    t_up = time.perf_counter()
    # per shard: the site table, the reads' BAM-native arrays in device memory (spl_soa_upload), and a read set laid out from them
    # once already (spl_reads_finish: record slots, chunk list, queues exist; every timed step lays the records out again)
    copies = []
    for _ in range(max(1, args.copies)):
        cp = []
        for sh in shards:
            soa = ctx.upload_soa([rd for rd, _ in sh.read_segments])
            cp.append((ctx.upload_sites(sh.sites), ctx.layout_read_segments(soa, [shift for _, shift in sh.read_segments]), soa))
        copies.append(cp)
    ctx.sync()
    t_up = time.perf_counter() - t_up
    dev = copies[0]
    step_no = [0]
    alg_bytes = sum(native.algorithmic_bytes(ds, dr) for ds, dr, _ in dev)
    layout_bytes = [dr.layout_bytes() for _, dr, _ in dev]      # (arrays read, records written) per shard

    def step(layout=True):
        step_no[0] += 1
        for ds, dr, _ in copies[(args.steps + args.warmup - step_no[0]) % len(copies)]:   # (the last step works on copy 0)
            if layout:
                dr.relayout()
            ctx.count_launch(ds, dr, scode, 0, kflags)
            ctx.sse_launch(ds, args.beta2Cryptic)
        if not args.pipelined:
            ctx.pass_barrier()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.kernel_timing_begin(args.steps * len(dev))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    layout_ms = ctx.layout_timing_collect(args.steps * len(dev) + 8)
    kernel_ms = ctx.kernel_timing_collect(args.steps * len(dev) + 8)
    # the results of the LAST TIMED step, taken before anything else is launched: what the parity field below compares
    gpu_counts = [ds.counters() for ds, _, _ in dev]
    gpu_sse = [ds.sse_results() for ds, _, _ in dev]
    literal_reads = sum(dr.literal_queue_size() for _, dr, _ in dev) if args.kernel != "pairs" else None
    # outside the timed region: (a) the step as rounds 1-4 timed it -- from the records on, no layout -- for continuity
    # (roofline.count_only); (b) the range kernel with nothing beside it (a sync after every launch, so that the tail of a pass
    # is over before the next range kernel starts)
    # WEAK beside strong (N > 1, strong mode): a step of N passes over the rank's own stretch -- every GPU counts as many reads a
    # step as the one GPU of an N = 1 run does, in launches of 1/N the size -- timed the same way (barrier, MAX over ranks below)
    weak_elapsed = None
    if dist is not None and args.scaling == "strong" and world > 1:
        fence()
        tw = time.perf_counter()
        for _ in range(args.steps):
            for _rep in range(world):
                for ds, dr, _ in dev:
                    dr.relayout()
                    ctx.count_launch(ds, dr, scode, 0, kflags)
                    ctx.sse_launch(ds, args.beta2Cryptic)
            if not args.pipelined:
                ctx.pass_barrier()
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        weak_elapsed = time.perf_counter() - tw
        wt = torch.tensor([weak_elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        weak_elapsed = float(wt.item())
        same_w = all(all(np.array_equal(a, b) for a, b in zip(ds.counters(), g)) for (ds, _, _), g in zip(dev, gpu_counts))
        we = torch.tensor([1.0 if same_w else 0.0], dtype=torch.float64, device=red_device)
        dist.all_reduce(we, op=dist.ReduceOp.MIN)
        weak_same = bool(we.item() > 0.5)
    step_no[0] = 0
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(layout=False)
    ctx.sync()
    count_only_s = (time.perf_counter() - t1) / args.steps
    ctx.kernel_timing_begin(5 * len(dev))
    for _ in range(5):
        for ds, dr, _ in dev:
            ctx.count_launch(ds, dr, scode, 0, kflags)
            ctx.sse_launch(ds, args.beta2Cryptic)
            ctx.sync()
    kernel_ms_alone = ctx.kernel_timing_collect(5 * len(dev) + 8)
    info = ctx.launch_info()
    # (c) a fused pass (no records in memory: layout_bytes says 0 written) beside what it replaced: the same shards as read sets
    # with records in memory (SPL_FUSED=0), every step the layout kernel and the range kernel -- rounds 1-5's two kernels
    fused = all(b == 0 for _, b in layout_bytes) and args.kernel == "ranges"
    two_kernels = None
    if fused:
        os.environ["SPL_FUSED"] = "0"
        try:
            packed = [(ds, ctx.layout_read_segments(soa, [shift for _, shift in sh.read_segments])) for (ds, _, soa), sh in zip(dev, shards)]
        finally:
            del os.environ["SPL_FUSED"]
        n2 = max(2, min(args.steps, 10))

        def step2():
            for ds, dr in packed:
                dr.relayout()
                ctx.count_launch(ds, dr, scode, 0, kflags)
                ctx.sse_launch(ds, args.beta2Cryptic)
            ctx.pass_barrier()
        for _ in range(2):
            step2()
        ctx.sync()
        ctx.kernel_timing_begin(n2 * len(packed))
        t2 = time.perf_counter()
        for _ in range(n2):
            step2()
        ctx.sync()
        t2 = (time.perf_counter() - t2) / n2
        l2 = ctx.layout_timing_collect(n2 * len(packed) + 8)
        k2 = ctx.kernel_timing_collect(n2 * len(packed) + 8)
        lb2 = [dr.layout_bytes() for _, dr in packed]
        same2 = all(all(np.array_equal(a, b) for a, b in zip(ds.counters(), g)) for (ds, _), g in zip(packed, gpu_counts))
        two_kernels = {"ms_per_step": t2 * 1e3, "layout_kernel_ms_avg": float(np.mean(l2)) if l2 else None, "range_kernel_ms_avg": float(np.mean(k2)) if k2 else None,
                       "layout_bytes_per_launch": sum(a + b for a, b in lb2) / max(len(lb2), 1), "same_counters": bool(same2), "steps": n2}
        for _, dr in packed:
            dr.free()
    for cp in copies:
        for ds, dr, soa in cp:
            dr.free()
            soa.free()
            ds.free()

    tot_reads, tot_sites = float(n_reads), float(n_sites)
    imbalance = None
    my_elapsed = elapsed
    if dist is not None:
        exact_rank = None
        if strong is not None:
            # every rank holds its own stretch against the oracle: the PARTIAL counters of its pieces (and what findBeta2Counts +
            # calculateSSE make of them) are what the oracle gives for the same reads against the same tables
            nproc, quota = cpu_budget()
            _, want_mine = run_oracle(items, scode, args.beta2Cryptic, max(1, int((quota or nproc) // world)))
            exact_rank = True
            for sh, cnts, sses in zip(shards, gpu_counts, gpu_sse):
                for chrom, (r0, r1), (e0, e1) in zip(sh.chroms, sh.site_rows, sh.edge_rows):
                    (w1, w2, w3), wsse = want_mine[chrom]
                    exact_rank &= np.array_equal(cnts[0][r0:r1], w1) and np.array_equal(cnts[1][r0:r1], w2) and np.array_equal(cnts[2][e0:e1], w3)
                    exact_rank &= all(np.array_equal(g[r0:r1], w) for g, w in zip(sses, wsse))
        elapsed, tot_reads, tot_sites, imbalance = reduce_report(dist, rank, world, n_reads, n_sites, my_elapsed, red_device, exact=exact_rank)
        if weak_elapsed is not None:
            imbalance["weak_beside"] = {"value": tot_sites * world * args.steps / weak_elapsed, "reads_per_sec": tot_reads * world * args.steps / weak_elapsed,
                                        "ms_per_step": weak_elapsed / args.steps * 1e3, "same_counters": weak_same,
                                        "what": "weak scaling beside the strong figure: a step = every rank counts its own stretch %d times (the reads one GPU "
                                                "counts a step at N = 1, in launches of 1/%d the size); barrier + MAX over ranks as for `value`" % (world, world)}
        if strong is not None:
            imbalance["split"] = "one sample cut into %d stretches of equal expected reads in file order, chromosomes cut anywhere; every rank generated its own stretch only" % world
        # the collectives are over: what follows (parity, the end-to-end legs over ALL the node's GPUs) is rank 0's alone, the other
        # ranks give their GPUs back
        dist.barrier()
        dist.destroy_process_group()
        dist = None
        if rank != 0:
            ctx.close()
            return

    # ---- whole-workload parity check + CPU baseline (rank 0; the baseline only at N = 1) -------------
    # The oracle is the checker here, never the thing measured as "value".
    cpu = None
    parity = None
    e2e = None
    combine_res = None
    other_steps = None
    if rank == 0:
        t_cpu1, want = run_oracle(items, scode, args.beta2Cryptic, 1)
        exact = True
        for sh, cnts, sses in zip(shards, gpu_counts, gpu_sse):
            for chrom, (r0, r1), (e0, e1) in zip(sh.chroms, sh.site_rows, sh.edge_rows):
                (w1, w2, w3), wsse = want[chrom]
                exact &= np.array_equal(cnts[0][r0:r1], w1) and np.array_equal(cnts[1][r0:r1], w2) and np.array_equal(cnts[2][e0:e1], w3)
                exact &= all(np.array_equal(g[r0:r1], w) for g, w in zip(sses, wsse))
        parity = {"reads": n_reads, "sites": n_sites, "bit_exact_vs_oracle": bool(exact), "of": "the last timed step",
                  "checked": "beta1, beta2Simple(reads), double counts, beta2Simple, beta2Cryptic, beta2Weighted, SSE"}
        if imbalance is not None and "all_ranks_exact" in imbalance:     # (strong scaling: every rank checked its own stretch's partial counters)
            parity.update(reads=int(tot_reads), sites=int(round(tot_sites)), bit_exact_vs_oracle=bool(exact and imbalance["all_ranks_exact"]),
                          of="the last timed step, every rank its own stretch of the sample (partial counters of cut chromosomes)")
        if world == 1 and not args.no_cpu_baseline:
            nproc, quota = cpu_budget()
            n_threads = max(1, min(nproc, int(round(quota)) if quota else nproc))
            t_all, want_all = run_oracle(items, scode, args.beta2Cryptic, n_threads)
            same = all(all(np.array_equal(a, b) for a, b in zip(want[c][0], want_all[c][0])) for c in want)
            pairs = int(sum(int(w[0].sum()) + int(w[1].sum()) for w, _ in want.values()))
            cpu = {"value": n_sites / t_cpu1, "unit": "splice sites/s", "reads_per_sec": n_reads / t_cpu1, "cores": 1,
                   "kind": "port",
                   "sample_short": "whole workload, one pass of oracle/spliser_oracle.c, %.1f s" % t_cpu1,
                   "sample": "the whole workload (%d reads x %d sites), one pass (%.1f s) of oracle/spliser_oracle.c, the "
                             "site-centric C restatement of checkBam + findBeta2Counts + calculateSSE, on 1 thread" % (n_reads, n_sites, t_cpu1),
                   "all_cores": {"value": n_sites / t_all, "reads_per_sec": n_reads / t_all, "nproc": nproc,
                                 "cpu_quota_cores": quota, "threads_used": n_threads,
                                 "threads": "min(nproc, the cgroup's CPU quota) threads: up to 8 chromosomes side by side, the site "
                                            "loop of each on its share of them (OpenMP)",
                                 "seconds": t_all, "same_counts_as_1_thread": bool(same)},
                   "reference_cost_model": {"estimate_seconds": n_sites * 1.7e-3 + pairs * 7.1e-6,
                                            "label": "ESTIMATE, not a measurement: S*1.7 ms (one samtools spawn per site) + P*7.1 us "
                                                     "(Python per counted (read, site) pair, P >= %d) for SpliSER v0.1.8 on one core "
                                                     "(BASELINE.md; SpliSER_v0_1_8.py:422, :427-559)" % pairs},
                   "reference_measured": reference_measured()}
        other_steps = []
        if others:
            nproc, quota = cpu_budget()
            n_threads = max(1, min(nproc, int(round(quota)) if quota else nproc))
            while others:
                name, o_items, o_stranded, o_cryptic = others.pop(0)
                res = resident_step(ctx, o_items, o_stranded, o_cryptic, max(5, min(args.steps, 20)), 2, n_threads)
                res["workload"] = name
                other_steps.append(res)
                del o_items
        if args.e2e == "auto":
            nproc, quota = cpu_budget()
            n_threads = max(1, min(nproc, int(round(quota)) if quota else nproc))
            # (N > 1: the sequence-like file only -- rank 0 runs these legs alone after the collectives, on all the node's GPUs and
            #  once more on one, and the driver's window for the scaling run is not known to be longer than the one-GPU run's)
            modes = [args.e2e_seq_mode] if args.e2e_seq_mode is not None else ([1, 2, 0] if world == 1 else [1])
            devs = tuple(d % max(n_dev, 1) for d in range(world))     # (--share-devices: the ranks' devices, as the ranks took them)
            e2e = []
            for q in modes if world == 1 else []:     # (first the file that deflates like a real library's: the leg that says what the product does)
                e2e.append(e2e_leg(args.workload, wl, items, stranded, args.beta2Cryptic, q, args.e2e_reps, want, devices=devs,
                                   cpu_e2e=(q == 1 and not args.no_cpu_baseline),
                                   files=pre_files.pop(args.workload, None) if q == 1 else None, cold=cold.get(args.workload) if q == 1 else None))
            if wl_e2e is not None:     # N > 1: ONE file on all the node's GPUs, every device its own stretch of it -- beside the same call on one
                wl2, _, items2, stranded2 = wl_e2e
                _, want2 = run_oracle(items2, native.STRANDED_CODE[stranded2], args.beta2Cryptic, n_threads)
                for q in modes:
                    e2e.append(e2e_leg("arabidopsis", wl2, items2, stranded2, args.beta2Cryptic, q, args.e2e_reps, want2, devices=devs, compare_devices=(0,)))
            if wl_small is not None:
                wl2, _, items2, stranded2 = wl_small
                _, want2 = run_oracle(items2, native.STRANDED_CODE[stranded2], args.beta2Cryptic, n_threads)
                for q in modes:
                    e2e.append(e2e_leg("arabidopsis", wl2, items2, stranded2, args.beta2Cryptic, q, args.e2e_reps, want2, cpu_e2e=(q == 1 and not args.no_cpu_baseline),
                                       files=pre_files.pop("arabidopsis", None) if q == 1 else None, cold=cold.get("arabidopsis") if q == 1 else None))
                if 1 in modes:  # ... and once with the host decoder asked for
                    e2e.append(e2e_leg("arabidopsis", wl2, items2, stranded2, args.beta2Cryptic, 1, args.e2e_reps, want2, gpu_decode=False))
            if wl_mouse_e2e is not None:     # config 5's shape as a file (25 M reads, flags 99 / 147 / 83 / 163): --isStranded -s fr --beta2Cryptic
                wl3, _, items3, stranded3 = wl_mouse_e2e
                _, want3 = run_oracle(items3, native.STRANDED_CODE[stranded3], True, n_threads)
                e2e.append(e2e_leg("mouse_stranded", wl3, items3, stranded3, True, 1, args.e2e_reps, want3))
        if cmb_samples is not None:
            nproc, quota = cpu_budget()
            combine_res = combine_leg(cmb_samples, max(1, min(nproc, int(round(quota)) if quota else nproc)))
    for f in pre_files.values():     # (files of legs that did not run)
        shutil.rmtree(f["tmp"], ignore_errors=True)
    ctx.close()

    if rank == 0:
        # HBM bytes per launch from the PMC counters cannot be collected inside this process; they come from a committed
        # rocprofv3 --pmc run of this command (profiles/*_traffic.json, made by tools/prof_pmc.sh) and are quoted only for the
        # workload AND the kernel sources (kernel + packer) they were measured with.
        traffic, traffic_from, layout_traffic = None, None, None
        import glob
        sha, ksha = lib_sha16(), kernel_src_sha16()
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_traffic.json")), reverse=True):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("kernel_src_sha16") == ksha and tj.get("workload") == args.workload and args.scale == 1.0 and args.kernel == "ranges":
                traffic, traffic_from = tj["hbm_bytes_per_launch"], os.path.basename(tpath)
                layout_traffic = (tj.get("layout") or {}).get("hbm_bytes_per_launch")
                break
        k_avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        l_avg_ms = float(np.mean(layout_ms)) if layout_ms else float("nan")
        # one step = len(dev) launches of each kernel; bytes per launch and time per launch are both averaged over launches
        bytes_per_launch = alg_bytes / max(len(dev), 1)
        soa_per_launch = sum(a for a, _ in layout_bytes) / max(len(dev), 1)
        rec_per_launch = sum(b for _, b in layout_bytes) / max(len(dev), 1)
        range_gbs = bytes_per_launch / (k_avg_ms * 1e-3) / 1e9 if kernel_ms else float("nan")
        layout_gbs = (soa_per_launch + rec_per_launch) / (l_avg_ms * 1e-3) / 1e9 if layout_ms else float("nan")
        ms_per_step = elapsed / args.steps * 1e3
        path_gbs = alg_bytes / (my_elapsed / args.steps) / 1e9
        range_obj = {"kernel": "spl_count_%s_kernel%s" % (args.kernel.split("_")[0], "<FUSED>" if fused else ""), "kernel_ms_avg": k_avg_ms, "launches_timed": len(kernel_ms),
                     "algorithmic_bytes_per_launch": bytes_per_launch, "achieved": range_gbs, "frac": range_gbs / HBM_PEAK_GBS,
                     "what": ("SURVEY 8(d)'s bytes (the BAM-native inputs once, the outputs once) / the fused range kernel's mean duration: the kernel reads the "
                              "BAM-native arrays itself and makes a tile's records in LDS -- no layout kernel, no records in memory" if fused else
                              "SURVEY 8(d)'s bytes (the BAM-native inputs once, the outputs once) / the range kernel's mean duration: rounds 1-4's roofline.frac")}
        layout_obj = {"kernel": "spl_layout_kernel", "kernel_ms_avg": l_avg_ms if layout_ms else None, "launches_timed": len(layout_ms),
                      "algorithmic_bytes_per_launch": soa_per_launch + rec_per_launch, "arrays_read": soa_per_launch, "records_written": rec_per_launch,
                      "achieved": layout_gbs if layout_ms else None, "frac": layout_gbs / HBM_PEAK_GBS if layout_ms else None,
                      "what": ("not launched: the pass is fused (see range)" if fused else
                               "the arrays read once (10 B a read + 4 B an op) + the records written once / the layout kernel's mean duration")}
        if two_kernels is not None:
            tk = two_kernels
            tk["what"] = ("the same shards as read sets with records in memory (SPL_FUSED=0): layout kernel + range kernel every step, what rounds 1-5 "
                          "timed; frac: each kernel's bytes (layout: arrays read + records written; range: SURVEY 8(d)'s) / its duration / the HBM peak")
            tk["layout_frac"] = tk["layout_bytes_per_launch"] / (tk["layout_kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS if tk["layout_kernel_ms_avg"] else None
            tk["range_frac"] = bytes_per_launch / (tk["range_kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS if tk["range_kernel_ms_avg"] else None
            tk["path_frac"] = alg_bytes / (tk["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        # THE roofline object is the dominant kernel's: whichever of the two takes longer per launch
        dom, other = (layout_obj, range_obj) if (layout_ms and l_avg_ms >= k_avg_ms) else (range_obj, layout_obj)
        out = {
            "metric": "splice sites/sec (+ reads/sec) processed",
            "value": tot_sites * args.steps / elapsed,
            "unit": "splice sites/s",
            "reads_per_sec": tot_reads * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "int32/u32 counters, f64 SSE", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d splice sites%s, %d chromosomes in %d shard(s) on rank 0, 150 bp, %s"
                                   % (args.workload, int(tot_reads), int(tot_sites), " per GPU" if args.scaling == "weak" and world > 1 else "",
                                      len(items), len(dev), stranded or "unstranded"),
                       "scale": args.scale, "parallelism": "chromosome/sample shards, no collectives", "seed": synth.WORKLOADS[args.workload]["seed"],
                       "step": "BAM-native arrays resident in HBM -> %s + literal + scan/SSE kernels of every shard, %s"
                               % ("chunk map + order, fused range kernel (records made in LDS)" if fused else "layout + range",
                                  "no barrier between steps (pipelined)" if args.pipelined else "barrier between steps")},
            "roofline": {"bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["frac"], "traffic": traffic if dom is range_obj else layout_traffic, "traffic_from": traffic_from,
                         "kernel": dom["kernel"], "kernel_ms_avg": dom["kernel_ms_avg"], "launches_timed": dom["launches_timed"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                         "what": dom["what"],
                         "layout": layout_obj, "range": range_obj, "second_kernel": None if fused else other["kernel"], "fused": fused, "two_kernels": two_kernels,
                         "range_traffic": traffic, "range_traffic_from": traffic_from,
                         "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"],
                         "alone": (None if not kernel_ms_alone else
                                   {"kernel_ms_avg": sum(kernel_ms_alone) / len(kernel_ms_alone),
                                    "frac": bytes_per_launch / (sum(kernel_ms_alone) / len(kernel_ms_alone) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "how": "the range kernel: 5 more launches after the timed region, each followed by a sync"}),
                         "path": {"what": "SURVEY 8(d)'s algorithmic bytes of a step / time of a step: every kernel of the pass (layout included) and the gaps between them",
                                  "achieved": path_gbs, "frac": path_gbs / HBM_PEAK_GBS},
                         "count_only": {"what": ("the step without the chunk map and order kernels (they need the CIGAR offsets only)" if fused else
                                                 "the step as rounds 1-4 timed it: from the records on (range + literal + scan), no layout"),
                                        "ms_per_step": count_only_s * 1e3, "reads_per_sec": n_reads / count_only_s,
                                        "path_frac": alg_bytes / count_only_s / 1e9 / HBM_PEAK_GBS},
                         "hbm_actual": (None if not traffic or not kernel_ms else
                                        {"what": "the range kernel: HBM bytes per launch by the PMC counters (range_traffic) / the kernel's time: what the "
                                                 "memory system really moved, beside the algorithmic rate",
                                         "GBps": traffic / (k_avg_ms * 1e-3) / 1e9, "frac": traffic / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}),
                         "lib_sha16": sha, "kernel_src_sha16": ksha, "ingest_src_sha16": ingest_src_sha16()},
            "cpu_baseline": cpu,
            "parity": parity,
            "e2e": e2e,
            "combine": combine_res,
            "other_steps": other_steps or None,
            "imbalance": imbalance,
            "literal_kernel_reads": literal_reads,
            "gen_seconds": t_gen, "upload_seconds": t_up,
        }
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
