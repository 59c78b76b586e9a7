#!/usr/bin/env python3
"""bench.py -- throughput of the SpliSER `process` hot path (Step 3) on MI355X.

One "step" = one pass of the hot path over one synthetic sample whose SoA is already resident in HBM:
spl_count_kernel (the checkBam loop, SpliSER_v0_1_8.py:408-559) + spl_sse_kernel (findBeta2Counts +
calculateSSE, :581-639) for every shard of the sample.  Default workload = BASELINE.json configs[1]
("A. thaliana whole-genome process, ~20M 150 bp reads, 1 MI355X"), synthesised from a seed
(spliser_amd/synth.py) because there is no network and the reference ships no data.

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards with no exchange step, so
every rank processes its own sample of the same shape (weak scaling, seed + rank); the only collectives
are the timing barrier and the MAX reduction of the elapsed time.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="arabidopsis", choices=["arabidopsis", "human", "mouse_stranded", "single_gene"])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the workload's read count (debug)")
    ap.add_argument("--stranded", default=None, choices=[None, "fr", "rf"])
    ap.add_argument("--beta2Cryptic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from spliser_amd import native, shard, sites, synth

    # ---- synthetic sample of this rank (generated BEFORE the GPU is touched: the generator forks) -----
    t_gen = time.perf_counter()
    cfg = synth.WORKLOADS[args.workload]
    stranded = args.stranded or ("fr" if cfg.get("paired") else None)
    cache = None
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)
        cache = os.path.join(args.cache, "%s_s%g_seed%d%s.npz" % (args.workload, args.scale, cfg["seed"] + rank,
                                                                   "" if args.genes is None else "_g%d" % args.genes))
    if cache and os.path.exists(cache):
        wl = synth.Workload.load(cache, args.workload)
    else:
        over = {} if args.alt_fraction is None else {"alt_fraction": args.alt_fraction}
        if args.genes is not None:
            over["n_genes"] = args.genes
        wl = synth.Workload(args.workload, scale=args.scale, seed=cfg["seed"] + rank, **over,
                            workers=max(1, min(8, (os.cpu_count() or 1) // max(world, 1))))
        if cache:
            wl.save(cache)
    if args.soft_clips > 0:
        wl.reads = [synth.add_soft_clips(r, args.soft_clips, seed=100 + k) for k, r in enumerate(wl.reads)]
    tmp = tempfile.mkdtemp(prefix="spliser_bench_")
    bed = os.path.join(tmp, "junctions.bed")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
    from spliser_amd import fast_sites
    table = fast_sites.build(sites.GeneBins(), bool(stranded), bed)   # the same table `process` builds (Steps 1-2)
    if table is None:
        table = sites.SiteTable(is_stranded=bool(stranded))
        table.add_bed(bed)
        table.find_competitors()
    names = wl.genome.chrom_names
    items = []
    for i, c in enumerate(names):
        arr = table.chrom_arrays(c)
        if arr.n:
            items.append((c, arr, wl.reads[i]))
    shards = shard.pack(items)
    n_reads = sum(sh.reads.n for sh in shards)
    n_sites = sum(sh.sites.n for sh in shards)
    t_gen = time.perf_counter() - t_gen

    import torch
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    scode = native.STRANDED_CODE[stranded]
    kflags = {"pairs": native.OPT_PAIR_KERNEL, "ranges_agg": native.OPT_WAVE_AGGREGATION, "ranges": 0}[args.kernel]
    ctx = native.Context(local_rank)
    copies = [[(ctx.upload_sites(sh.sites), ctx.upload_reads(sh.reads)) for sh in shards] for _ in range(max(1, args.copies))]
    dev = copies[0]
    step_no = [0]
    alg_bytes = sum(native.algorithmic_bytes(ds, dr) for ds, dr in dev)

    def step():
        step_no[0] += 1
        for ds, dr in copies[(args.steps + args.warmup - step_no[0]) % len(copies)]:   # (the last step works on copy 0)
            ctx.count_launch(ds, dr, scode, 0, kflags)
            ctx.sse_launch(ds, args.beta2Cryptic)

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
    kernel_ms = ctx.kernel_timing_collect(args.steps * len(dev) + 8)
    # outside the timed region: the same kernel with nothing beside it (a sync after every launch, so that the tail of a pass is
    # over before the next range kernel starts) -- what the tail stream costs the kernel it overlaps with
    ctx.kernel_timing_begin(5 * len(dev))
    for _ in range(5):
        for ds, dr in dev:
            ctx.count_launch(ds, dr, scode, 0, kflags)
            ctx.sse_launch(ds, args.beta2Cryptic)
            ctx.sync()
    kernel_ms_alone = ctx.kernel_timing_collect(5 * len(dev) + 8)
    literal_reads = sum(dr.literal_queue_size() for _, dr in dev) if args.kernel != "pairs" else None
    info = ctx.launch_info()

    tot_reads, tot_sites = float(n_reads), float(n_sites)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([tot_reads, tot_sites], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_reads, tot_sites = float(c[0].item()), float(c[1].item())

    # ---- whole-workload parity check + CPU baseline (rank 0; the baseline only at N = 1) -------------
    # The oracle is the checker here, never the thing measured as "value".  It is so much faster than the
    # Python reference (tens of M reads/s on one core) that the bounded sample is the ENTIRE workload,
    # repeated until ~cpu-seconds of CPU work have been timed.
    cpu = None
    parity = None
    if rank == 0:
        from oracle import oracle
        gpu_counts = [ds.counters() for ds, _ in dev]          # results of the last timed step
        gpu_sse = [ds.sse_results() for ds, _ in dev]

        def run_oracle():
            res = {}
            t = time.perf_counter()
            for c, arr, rd in items:
                cnt = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                       rd.pos, rd.flag, rd.cig_off, rd.cigar, scode, 0)
                sse = oracle.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                                       cnt[0], cnt[1], cnt[2], args.beta2Cryptic)
                res[c] = (cnt, sse)
            return time.perf_counter() - t, res

        t_cpu, want = run_oracle()
        exact = True
        for sh, cnts, sses in zip(shards, gpu_counts, gpu_sse):
            for chrom, (r0, r1), (e0, e1) in zip(sh.chroms, sh.site_rows, sh.edge_rows):
                (w1, w2, w3), wsse = want[chrom]
                exact &= np.array_equal(cnts[0][r0:r1], w1) and np.array_equal(cnts[1][r0:r1], w2) and np.array_equal(cnts[2][e0:e1], w3)
                exact &= all(np.array_equal(g[r0:r1], w) for g, w in zip(sses, wsse))
        parity = {"reads": n_reads, "sites": n_sites, "bit_exact_vs_oracle": bool(exact),
                  "checked": "beta1, beta2Simple(reads), double counts, beta2Simple, beta2Cryptic, beta2Weighted, SSE"}
        if world == 1 and not args.no_cpu_baseline:
            times = [t_cpu]
            while sum(times) < args.cpu_seconds and len(times) < 50:
                times.append(run_oracle()[0])
            best = min(times)
            cpu = {"value": n_sites / best, "unit": "splice sites/s", "reads_per_sec": n_reads / best, "cores": 1,
                   "kind": "port",
                   "sample": "the whole workload (%d reads x %d sites), oracle/spliser_oracle.c site-centric C "
                             "restatement on 1 thread, best of %d passes (%.2f s each, %.1f s total)"
                             % (n_reads, n_sites, len(times), best, sum(times))}

    for ds, dr in dev:
        dr.free()
        ds.free()
    ctx.close()

    if rank == 0:
        # HBM bytes per launch from the PMC counters cannot be collected inside this process; they come from the committed
        # rocprofv3 --pmc passes of the same command (profiles/), and only for the exact workload they were measured on.
        traffic = None
        import glob
        tfiles = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_traffic.json")))
        tpath = tfiles[-1] if tfiles else ""  # the newest committed measurement (file names sort by round and build)
        if (args.workload == "arabidopsis" and args.scale == 1.0 and args.kernel == "ranges" and not stranded
                and args.alt_fraction is None and args.genes is None and args.soft_clips == 0 and os.path.exists(tpath)):
            with open(tpath) as fh:
                traffic = json.load(fh)["hbm_bytes_per_launch"]
        k_avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        # one step = len(dev) launches; bytes per launch and time per launch are both averaged over launches
        bytes_per_launch = alg_bytes / max(len(dev), 1)
        achieved = bytes_per_launch / (k_avg_ms * 1e-3) / 1e9 if kernel_ms else float("nan")
        out = {
            "metric": "splice sites/sec (+ reads/sec) processed",
            "value": tot_sites * args.steps / elapsed,
            "unit": "splice sites/s",
            "reads_per_sec": tot_reads * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32/u32 counters, f64 SSE", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d splice sites per GPU, %d chromosomes in %d shard(s), 150 bp, %s"
                                   % (args.workload, n_reads, n_sites, len(items), len(dev), stranded or "unstranded"),
                       "scale": args.scale, "parallelism": "chromosome/sample shards, no collectives", "seed": cfg["seed"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "spl_count_%s_kernel" % args.kernel.split("_")[0], "kernel_ms_avg": k_avg_ms, "launches_timed": len(kernel_ms),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"],
                         "alone": (None if not kernel_ms_alone else
                                   {"kernel_ms_avg": sum(kernel_ms_alone) / len(kernel_ms_alone),
                                    "frac": bytes_per_launch / (sum(kernel_ms_alone) / len(kernel_ms_alone) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "how": "5 more launches after the timed region, each followed by a sync"}),
                         "concurrent": (None if os.environ.get("SPL_TAIL_STREAM", "1")[:1] == "0" or args.kernel == "pairs" else
                                        "the literal kernel and the scan of the launch before run beside this kernel on a stream of "
                                        "their own: its duration includes what it yields to them (alone: SPL_TAIL_STREAM=0)")},
            "cpu_baseline": cpu,
            "parity": parity,
            "literal_kernel_reads": literal_reads,
            "gen_seconds": t_gen,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
