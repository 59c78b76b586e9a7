"""N > 1 paths on CPU: chromosome -> device assignment, shard packing round trip, and world_size-2 gloo runs of the
rank/shard logic bench.py uses -- weak scaling (every rank owns its own sample) and STRONG scaling, bench.py's default for
N > 1 (one sample, its chromosomes dealt to the ranks: bench.rank_items, bench.reduce_report); the only collectives are the
timing barrier and the MAX/SUM reductions of the report."""
import os
import socket
import sys

import numpy as np
import pytest

from spliser_amd import samio, shard, sites, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assign_is_balanced_and_complete():
    w = {"chr%d" % i: n for i, n in enumerate([900, 850, 400, 390, 380, 100, 90, 80, 5, 1])}
    for n_dev in (1, 2, 4, 8):
        bins = shard.assign(w, n_dev)
        assert len(bins) == n_dev
        flat = sorted(c for b in bins for c in b)
        assert flat == sorted(w)
        loads = [sum(w[c] for c in b) for b in bins]
        assert max(loads) <= sum(w.values()) / n_dev + max(w.values())
    assert shard.assign({}, 4) == [[], [], [], []]


def _table(wl, tmp_path):
    bed = str(tmp_path / "j.bed")
    synth.write_bed(bed, wl.genome.chrom_names, wl.junctions)
    t = sites.SiteTable(is_stranded=False)
    t.add_bed(bed)
    t.find_competitors()
    return t


def test_pack_offsets_keep_chromosomes_apart(tmp_path, oracle_lib):
    """Packing several chromosomes into one coordinate space must not change any counter: run the oracle on the
    packed shard and on every chromosome alone."""
    wl = synth.Workload("arabidopsis", scale=0.002, seed=3, workers=1)
    table = _table(wl, tmp_path)
    items = [(c, table.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names) if table.chrom_arrays(c).n]
    shards = shard.pack(items)
    assert len(shards) == 1
    sh = shards[0]
    assert all(b > a for a, b in zip(sh.offsets, sh.offsets[1:]))
    s, r = sh.sites, sh.reads
    assert np.all(np.diff(s.pos.astype(np.int64)) >= 0) and np.all(np.diff(r.pos.astype(np.int64)) >= 0)
    got = oracle_lib.check_bam(s.pos, s.strand, s.part_off, s.part_pos, s.comp_off, s.comp_pos, r.pos, r.flag, r.cig_off, r.cigar)
    for (chrom, arr, rd), (r0, r1), (e0, e1) in zip(items, sh.site_rows, sh.edge_rows):
        want = oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                    rd.pos, rd.flag, rd.cig_off, rd.cigar)
        assert np.array_equal(got[0][r0:r1], want[0]) and np.array_equal(got[1][r0:r1], want[1])
        assert np.array_equal(got[2][e0:e1], want[2])
        # partner rows are re-based into the packed table
        ps = s.part_site[e0:e1]
        assert np.array_equal(np.where(ps >= 0, ps - r0, -1), arr.part_site)


def test_pack_splits_when_coordinates_overflow():
    big = 1_500_000_000
    items = []
    for name in ("a", "b", "c"):
        arr = sites.ChromArrays()
        arr.chrom, arr.n = name, 2
        arr.pos = np.array([10, big], np.int64)
        arr.strand = np.array([43, 43], np.uint8)
        arr.part_off = np.array([0, 1, 2], np.uint32)
        arr.part_pos = np.array([big, 10], np.int64)
        arr.part_site = np.array([1, 0], np.int32)
        arr.edge_cnt = np.array([1, 1], np.int64)
        arr.comp_off = np.zeros(3, np.uint32)
        arr.comp_pos = np.zeros(0, np.int64)
        arr.alpha = np.array([1, 1], np.int64)
        items.append((name, arr, samio.ReadSet.from_records([(0, 5, "50M")])))
    shards = shard.pack(items)
    assert len(shards) == 3
    assert [sh.chroms for sh in shards] == [["a"], ["b"], ["c"]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank owns its own sample (seed + rank), exactly like bench.py
    wl = synth.Workload("single_gene", n_reads=3000, seed=40 + rank, workers=1)
    reads = wl.reads[0]
    import tempfile
    tmp = tempfile.mkdtemp()
    synth.write_bed(os.path.join(tmp, "j.bed"), wl.genome.chrom_names, wl.junctions)
    t = sites.SiteTable()
    t.add_bed(os.path.join(tmp, "j.bed"))
    t.find_competitors()
    arr = t.chrom_arrays("Chr1")
    cnt = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                           reads.pos, reads.flag, reads.cig_off, reads.cigar)
    dist.barrier()
    elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    tot = torch.tensor([float(reads.n), float(arr.n), float(cnt[0].sum())], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        out.put((elapsed.item(), tot.tolist()))
    dist.destroy_process_group()


def test_two_rank_gloo_weak_scaling_reduction():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    elapsed, tot = out.get()
    assert elapsed == 2.0                      # MAX over ranks
    assert tot[0] == 6000.0 and tot[1] > 0     # reads summed over both samples


def _strong_worker(rank, world, port, out):
    """What a rank of `bench.py --gpus 2` does with the sample, the oracle standing in for the device: the same sample on every
    rank, bench.rank_items picks the rank's chromosomes, bench.reduce_report makes the report."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import tempfile
    import torch.distributed as dist
    import bench
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = synth.Workload("arabidopsis", scale=0.003, seed=21, workers=1)
    tmp = tempfile.mkdtemp()
    synth.write_bed(os.path.join(tmp, "j.bed"), wl.genome.chrom_names, wl.junctions)
    t = sites.SiteTable()
    t.add_bed(os.path.join(tmp, "j.bed"))
    t.find_competitors()
    items = [(c, t.chrom_arrays(c), wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names) if t.chrom_arrays(c).n]
    mine = bench.rank_items(items, "strong", world, rank)
    res = {}
    for c, arr, rd in mine:
        res[c] = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, rd.pos, rd.flag, rd.cig_off, rd.cigar)
    dist.barrier()
    n_reads, n_sites = sum(rd.n for _, _, rd in mine), sum(arr.n for _, arr, _ in mine)
    report = bench.reduce_report(dist, rank, world, n_reads, n_sites, 0.5 + rank, "cpu")
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({c: [a.tolist() for a in v] for c, v in res.items()}, gathered, dst=0)   # (the TEST's own collective, not bench.py's)
    if rank == 0:
        whole = {c: [a.tolist() for a in oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                                         rd.pos, rd.flag, rd.cig_off, rd.cigar)] for c, arr, rd in items}
        out.put((report, gathered, whole, {c: rd.n for c, _, rd in items}, sum(arr.n for _, arr, _ in items)))
    dist.destroy_process_group()


def test_two_rank_gloo_strong_split_union_is_the_whole():
    """bench.py's default for N > 1: ONE sample, its chromosomes dealt to the ranks.  Every chromosome goes to exactly one rank,
    the union of the ranks' counters is the oracle's on the whole sample, and the report carries the imbalance."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    (elapsed, tot_reads, tot_sites, imbalance), gathered, whole, reads_by_chrom, n_sites = out.get()
    n_reads = sum(reads_by_chrom.values())
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert elapsed == 1.5                                   # MAX over ranks
    assert tot_reads == float(n_reads) and tot_sites == float(n_sites)
    assert not (set(gathered[0]) & set(gathered[1]))        # no chromosome twice ...
    union = dict(gathered[0])
    union.update(gathered[1])
    assert union == whole                                   # ... none missing, every counter the whole sample's
    assert len(gathered[0]) >= 1 and len(gathered[1]) >= 1
    assert imbalance["reads_per_rank"] == [sum(reads_by_chrom[c] for c in g) for g in gathered]
    assert sum(imbalance["reads_per_rank"]) == n_reads and imbalance["seconds_per_rank"] == [0.5, 1.5]
    assert 1.0 <= imbalance["max_over_mean_reads"] < 2.0


def test_bench_refuses_a_world_that_is_not_its_gpus(tmp_path):
    """`--gpus N` is what the driver passes: a launcher that made another number of ranks is an error, not a silent one-GPU run."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.001"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
