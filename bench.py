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
    ap.add_argument("--cache", default=None, help="directory to cache the generated sample in (.npz); a cached "
                    "sample is loaded instead of regenerated (use under rocprofv3: no generator worker processes)")
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
        cache = os.path.join(args.cache, "%s_s%g_seed%d.npz" % (args.workload, args.scale, cfg["seed"] + rank))
    if cache and os.path.exists(cache):
        wl = synth.Workload.load(cache, args.workload)
    else:
        wl = synth.Workload(args.workload, scale=args.scale, seed=cfg["seed"] + rank,
                            workers=max(1, min(8, (os.cpu_count() or 1) // max(world, 1))))
        if cache:
            wl.save(cache)
    tmp = tempfile.mkdtemp(prefix="spliser_bench_")
    bed = os.path.join(tmp, "junctions.bed")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
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
    ctx = native.Context(local_rank)
    dev = [(ctx.upload_sites(sh.sites), ctx.upload_reads(sh.reads)) for sh in shards]
    alg_bytes = sum(native.algorithmic_bytes(ds, dr) for ds, dr in dev)

    def step():
        for ds, dr in dev:
            ctx.count_launch(ds, dr, scode, 0)
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
    info = ctx.launch_info()

    tot_reads, tot_sites = float(n_reads), float(n_sites)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([tot_reads, tot_sites], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_reads, tot_sites = float(c[0].item()), float(c[1].item())

    # ---- parity spot check + CPU baseline (rank 0, N = 1 only for the baseline) ---------------------
    cpu = None
    parity = None
    if rank == 0:
        from oracle import oracle
        c0, arr0, reads0 = max(items, key=lambda it: it[2].n)
        probe = min(reads0.n, 200_000)

        def sample(m):
            off = reads0.cig_off[: m + 1]
            return reads0.pos[:m], reads0.flag[:m], off, reads0.cigar[: int(off[-1])]

        def run_oracle(m):
            p, f, o, g = sample(m)
            t = time.perf_counter()
            cnt = oracle.check_bam(arr0.pos, arr0.strand, arr0.part_off, arr0.part_pos, arr0.comp_off, arr0.comp_pos,
                                   p, f, o, g, scode, 0)
            sse = oracle.beta2_sse(arr0.pos, arr0.part_off, arr0.part_pos, arr0.part_site, arr0.alpha, arr0.edge_cnt,
                                   cnt[0], cnt[1], cnt[2], args.beta2Cryptic)
            return time.perf_counter() - t, cnt, sse

        t_probe, cnt, sse = run_oracle(probe)
        m = probe
        if world == 1 and not args.no_cpu_baseline:
            rate = probe / max(t_probe, 1e-6)
            m = int(min(reads0.n, max(probe, rate * args.cpu_seconds)))
            t_cpu, cnt, sse = run_oracle(m)
            cpu = {"value": m / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sites_per_sec": arr0.n / t_cpu,
                   "sample": "first %d coordinate-sorted reads of %s x its %d sites, oracle/spliser_oracle.c "
                             "(site-centric C restatement, 1 thread), %.1f s" % (m, c0, arr0.n, t_cpu)}
        p, f, o, g = sample(m)
        s_arr = native.SiteArrays.from_chrom(arr0)
        got = ctx.count(s_arr, native.ReadArrays(p, f, o, g), scode, 0)
        got_sse = ctx.sse(s_arr, got[0], got[1], got[2], args.beta2Cryptic)
        exact = all(np.array_equal(a, b) for a, b in zip(got, cnt)) and all(np.array_equal(a, b) for a, b in zip(got_sse, sse))
        parity = {"reads": m, "sites": arr0.n, "bit_exact_vs_oracle": bool(exact)}

    for ds, dr in dev:
        dr.free()
        ds.free()
    ctx.close()

    if rank == 0:
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "spl_count_kernel", "kernel_ms_avg": k_avg_ms, "launches_timed": len(kernel_ms),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "grid": info["grid"], "block": info["block"], "lds_bytes": info["lds_bytes"]},
            "cpu_baseline": cpu,
            "parity": parity,
            "gen_seconds": t_gen,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
