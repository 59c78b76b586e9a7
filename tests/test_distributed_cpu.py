"""N > 1 paths on CPU: chromosome -> device assignment, shard packing round trip, and world_size-2 gloo runs of the
rank/shard logic bench.py uses -- weak scaling (every rank owns its own sample) and STRONG scaling, bench.py's default for
N > 1 (one sample cut into equal stretches of reads, chromosomes cut anywhere, partial counters that add up: bench.rank_pieces,
bench.reduce_report); the only collectives are the timing barrier and the MAX/MIN/SUM reductions of the report."""
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
    """What a rank of `bench.py --gpus 2` does with the sample, the oracle standing in for the device: ONE sample cut into `world`
    stretches of equal numbers of reads in file order -- chromosomes cut anywhere (bench.rank_pieces = synth.strong_plan) --, the
    rank counts its pieces against its chromosomes' WHOLE tables, bench.reduce_report makes the report."""
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
    mine = bench.rank_pieces(items, world, rank)
    res = {}
    for c, arr, rd in mine:
        res[c] = oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, rd.pos, rd.flag, rd.cig_off, rd.cigar)
    dist.barrier()
    n_reads = sum(rd.n for _, _, rd in mine)
    whole_n = {c: rd.n for c, _, rd in items}
    n_sites = sum(arr.n * rd.n / max(whole_n[c], 1) for c, arr, rd in mine)
    report = bench.reduce_report(dist, rank, world, n_reads, n_sites, 0.5 + rank, "cpu", exact=True)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({c: ([a.tolist() for a in v], rd.n) for (c, _, rd), v in zip(mine, res.values())}, gathered, dst=0)   # (the TEST's own collective, not bench.py's)
    if rank == 0:
        whole = {c: [a.tolist() for a in oracle.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos,
                                                         rd.pos, rd.flag, rd.cig_off, rd.cigar)] for c, arr, rd in items}
        out.put((report, gathered, whole, {c: rd.n for c, _, rd in items}, sum(arr.n for _, arr, _ in items)))
    dist.destroy_process_group()


def test_two_rank_gloo_strong_split_cuts_a_chromosome_and_the_sums_are_the_whole():
    """bench.py's default for N > 1 (and what `process --gpus N` does to a BAM): ONE sample in stretches of equal numbers of reads,
    a chromosome cut anywhere.  Two ranks: every read on exactly one, ONE chromosome on both -- and the ranks' partial counters of
    it, added, are the oracle's on the whole chromosome (checkBam only ever adds one per read, SpliSER_v0_1_8.py:519-559); the
    report carries the balance."""
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
    assert tot_reads == float(n_reads) and abs(tot_sites - n_sites) < 1e-6 * n_sites + 1e-9
    both = set(gathered[0]) & set(gathered[1])
    assert len(both) == 1                                   # the cut falls inside one chromosome: both ranks count it ...
    summed = {}
    for g in gathered:
        for c, (counters, n) in g.items():
            if c in summed:
                summed[c] = ([(np.array(a) + np.array(b)).tolist() for a, b in zip(summed[c][0], counters)], summed[c][1] + n)
            else:
                summed[c] = (counters, n)
    assert {c: v[0] for c, v in summed.items()} == whole    # ... and the sums are the whole sample's counters, every chromosome's
    assert {c: v[1] for c, v in summed.items()} == reads_by_chrom      # every read on exactly one rank
    assert imbalance["reads_per_rank"] == [sum(n for _, n in g.values()) for g in gathered]
    assert sum(imbalance["reads_per_rank"]) == n_reads and imbalance["seconds_per_rank"] == [0.5, 1.5]
    assert 1.0 <= imbalance["max_over_mean_reads"] < 1.001 and imbalance["all_ranks_exact"] is True


def test_strong_plan_and_a_rank_that_generates_its_own_stretch_only():
    """synth.strong_plan cuts by the EXPECTED reads per chromosome (known before any read is made), and a rank generates the
    chromosomes of its stretch only (Workload(keep_chroms=...)): the same reads the whole sample has there.  Eight ranks on a
    five-chromosome genome: every rank has work, the pieces tile every chromosome exactly, and rank 3's pieces made from its own
    partial sample are the pieces cut from the whole one."""
    wl = synth.Workload("arabidopsis", scale=0.004, seed=23, workers=1)
    expected = synth.expected_reads_per_chrom(wl.genome, wl.n_reads)
    actual = np.array([r.n for r in wl.reads], float)
    assert np.all(np.abs(expected - actual) < 0.03 * actual + 50)
    plan = synth.strong_plan(expected, 8)
    assert len(plan) == 8 and all(plan)
    for c in range(5):
        cuts = sorted((f0, f1) for pieces in plan for ci, f0, f1 in pieces if ci == c)
        assert cuts[0][0] == 0.0 and cuts[-1][1] == 1.0 and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    exp_share = [sum(expected[ci] * (f1 - f0) for ci, f0, f1 in pieces) for pieces in plan]
    assert max(exp_share) / (sum(exp_share) / 8) < 1.0001
    mine = plan[3]
    part = synth.Workload("arabidopsis", scale=0.004, seed=23, workers=1, genome=wl.genome, keep_chroms=sorted(set(c for c, _, _ in mine)))
    for ci, f0, f1 in mine:
        a, b = wl.reads[ci], part.reads[ci]
        assert a.n == b.n
        pa = a.take(int(f0 * a.n), a.n if f1 >= 1.0 else int(f1 * a.n))
        pb = b.take(int(f0 * b.n), b.n if f1 >= 1.0 else int(f1 * b.n))
        assert pa.n > 0 and np.array_equal(pa.pos, pb.pos) and np.array_equal(pa.flag, pb.flag) and np.array_equal(pa.cigar, pb.cigar) and np.array_equal(pa.cig_off, pb.cig_off)
    assert all(part.reads[c].n == 0 for c in range(5) if c not in set(ci for ci, _, _ in mine))


def test_bench_refuses_a_world_that_is_not_its_gpus(tmp_path):
    """`--gpus N` is what the driver passes: a launcher that made another number of ranks is an error, not a silent one-GPU run."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.001"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
