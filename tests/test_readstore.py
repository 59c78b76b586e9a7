"""``<out>.SpliSER.reads`` (spliser_amd/readstore.py): what ``process --keepReads`` leaves for ``combine`` -- flag, POS and CIGAR of
every read, all ``checkBam`` reads of an alignment (SpliSER_v0_1_8.py:434-437) -- comes back as it was written, and is taken
ONLY while it is still its BAM's: a BAM that has changed since (size, time, edges), a truncated or foreign file, another
version are all ignored (the BAM is then decoded as always)."""
import os

import numpy as np
import pytest

from spliser_amd import native, readstore, samio, synth


def _sample(tmp_path, seed=5):
    wl = synth.Workload("arabidopsis", scale=0.001, seed=seed, workers=1)
    bam = str(tmp_path / "s.bam")
    native.build()
    native.write_bam(bam, wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    return wl, bam


def test_round_trip_and_key(tmp_path):
    wl, bam = _sample(tmp_path)
    path = str(tmp_path / "s.SpliSER.reads")
    sets = [(c, wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names)] + [("empty", samio.ReadSet.empty())]
    readstore.save(path, bam, sets)
    rs = readstore.open_if_fresh(path, bam)
    assert rs is not None and rs.n_reads == sum(s.n for _, s in sets)
    for name, want in sets:
        got = rs.reads(name)
        assert got.n == want.n and got.max_end == want.max_end
        for k in ("pos", "flag", "cigar"):
            assert np.array_equal(getattr(got, k), getattr(want, k)[:len(getattr(got, k))])
        assert np.array_equal(got.cig_off, want.cig_off[:want.n + 1])
    assert rs.reads("no_such_reference") is None
    assert readstore.path_for_tsv("/x/y.SpliSER.tsv") == "/x/y.SpliSER.reads" and readstore.path_for_tsv("/x/y.tsv") is None
    rs.close()


def test_a_stale_or_foreign_file_is_ignored(tmp_path):
    wl, bam = _sample(tmp_path)
    path = str(tmp_path / "s.SpliSER.reads")
    sets = [(c, wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names)]
    readstore.save(path, bam, sets)
    assert readstore.open_if_fresh(path, bam) is not None
    assert readstore.open_if_fresh(str(tmp_path / "missing.SpliSER.reads"), bam) is None
    # the BAM touched (same bytes, later time): no longer this file's
    st = os.stat(bam)
    os.utime(bam, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    assert readstore.open_if_fresh(path, bam) is None
    os.utime(bam, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert readstore.open_if_fresh(path, bam) is not None
    # the BAM rewritten with other reads but -- forced -- the same size and time: its edges differ
    other = str(tmp_path / "o.bam")
    wl2 = synth.Workload("arabidopsis", scale=0.001, seed=6, workers=1)
    native.write_bam(other, wl2.genome.chrom_names, wl2.genome.chrom_lengths, wl2.reads, level=1, threads=2, seq_mode=1)
    data = open(other, "rb").read()
    size = os.path.getsize(bam)
    patched = (data + b"\0" * size)[:size]
    with open(bam, "wb") as fh:
        fh.write(patched)
    os.utime(bam, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert readstore.open_if_fresh(path, bam) is None
    # a truncated file, a file of another version, a file that is something else
    wl3, bam3 = wl, str(tmp_path / "t.bam")
    native.write_bam(bam3, wl3.genome.chrom_names, wl3.genome.chrom_lengths, wl3.reads, level=1, threads=2, seq_mode=1)
    readstore.save(path, bam3, sets)
    whole = open(path, "rb").read()
    with open(path, "wb") as fh:
        fh.write(whole[:len(whole) // 2])
    assert readstore.open_if_fresh(path, bam3) is None
    with open(path, "wb") as fh:
        fh.write(whole[:8] + b"\x07\0\0\0" + whole[12:])
    assert readstore.open_if_fresh(path, bam3) is None
    with open(path, "wb") as fh:
        fh.write(b"Region\tSite\n" * 100)
    assert readstore.open_if_fresh(path, bam3) is None
    with open(path, "wb") as fh:
        fh.write(whole)
    assert readstore.open_if_fresh(path, bam3) is not None


@pytest.mark.gpu
def test_combine_takes_kept_reads_and_ignores_stale_ones(tmp_path):
    """process --keepReads x 3, then combine: the .combined.tsv is byte for byte what combine makes from the BAMs; with one BAM
    replaced after its reads were kept, that sample's BAM is decoded again (the answer follows the BAM, not the stale file)."""
    from spliser_amd import combine as cmb, process
    base = synth.make_genome(synth.WORKLOADS["arabidopsis"]["chroms"], synth.WORKLOADS["arabidopsis"]["n_genes"],
                             synth.WORKLOADS["arabidopsis"]["intron"], seed=synth.WORKLOADS["arabidopsis"]["seed"])
    samples = [synth.Workload("arabidopsis", scale=0.01, genome=base, read_seed=31 + k, silence=0.15, workers=1) for k in range(3)]
    lines, logs = [], []
    for k, wl in enumerate(samples):
        prefix = str(tmp_path / ("s%d" % k))
        synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions, stranded=False)
        native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
        tm = process.process(prefix + ".bam", prefix + ".bed", prefix, keepReads=True, log=lambda m: None)
        assert os.path.exists(prefix + ".SpliSER.tsv")       # (the call's own output is there when it returns ...)
        process.wait_deferred_close()                        # (... the kept reads when the thread that closes the alignment file is through)
        assert "keep_reads_s" in tm and os.path.exists(prefix + ".SpliSER.reads")
        lines.append("S%d\t%s.SpliSER.tsv\t%s.bam\n" % (k, prefix, prefix))
    process.wait_deferred_close()
    sfile = str(tmp_path / "samples.tsv")
    open(sfile, "w").writelines(lines)
    cmb.combine(sfile, str(tmp_path / "kept"), log=logs.append)
    assert sum("reads kept by process" in m for m in logs) == 3
    os.environ["SPL_IGNORE_KEPT_READS"] = "1"
    try:
        cmb.combine(sfile, str(tmp_path / "bams"), log=lambda m: None)
    finally:
        del os.environ["SPL_IGNORE_KEPT_READS"]
    assert open(str(tmp_path / "kept.combined.tsv"), "rb").read() == open(str(tmp_path / "bams.combined.tsv"), "rb").read()
    # sample 1's BAM replaced by sample 2's reads: its kept reads are stale, the new BAM is what counts
    wl = samples[2]
    native.write_bam(str(tmp_path / "s1.bam"), wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=2, seq_mode=1)
    logs = []
    cmb.combine(sfile, str(tmp_path / "stale"), log=logs.append)
    assert sum("reads kept by process" in m for m in logs) == 2
    os.environ["SPL_IGNORE_KEPT_READS"] = "1"
    try:
        cmb.combine(sfile, str(tmp_path / "stale_bams"), log=lambda m: None)
    finally:
        del os.environ["SPL_IGNORE_KEPT_READS"]
    assert open(str(tmp_path / "stale.combined.tsv"), "rb").read() == open(str(tmp_path / "stale_bams.combined.tsv"), "rb").read()
    assert open(str(tmp_path / "stale.combined.tsv"), "rb").read() != open(str(tmp_path / "kept.combined.tsv"), "rb").read()
    process.wait_deferred_close()


def test_a_damaged_payload_is_not_taken(tmp_path):
    """a file of the right size and the right key whose payload is damaged -- a flipped byte among the CIGAR ops, CIGAR
    offsets that step back or end elsewhere than the reference's number of ops, negative counts in the header -- is not a source
    of reads: ``open_if_fresh`` gives None (and ``combine`` decodes the BAM)."""
    import json
    import struct
    wl, bam = _sample(tmp_path)
    path = str(tmp_path / "s.SpliSER.reads")
    sets = [(c, wl.reads[i]) for i, c in enumerate(wl.genome.chrom_names)]
    readstore.save(path, bam, sets)
    whole = bytearray(open(path, "rb").read())
    assert readstore.open_if_fresh(path, bam) is not None
    # one byte of the last array (the CIGAR ops)
    bad = bytearray(whole)
    bad[len(bad) - 200] ^= 0x10
    open(path, "wb").write(bad)
    assert readstore.open_if_fresh(path, bam) is None
    # the header's counts made negative (same length of text: "n": 12 -> "n": -2 is not, so: rewrite the header whole)
    head_len = struct.unpack("<I", whole[12:16])[0]
    head = json.loads(whole[16:16 + head_len].decode("utf-8"))
    head["refs"][0]["n"] = -head["refs"][0]["n"]
    text = json.dumps(head).encode("utf-8")
    if len(text) <= head_len:
        text = text + b" " * (head_len - len(text))
        open(path, "wb").write(whole[:16] + text + whole[16 + head_len:])
        assert readstore.open_if_fresh(path, bam) is None
    # CIGAR offsets that step back, the checksum made to fit them (a writer's bug rather than a damaged disk)
    rs = sets[0][1]
    off = rs.cig_off[:rs.n + 1].copy()
    off[rs.n // 2] = off[rs.n // 2 + 1] + 7
    broken = [(sets[0][0], samio.ReadSet(rs.pos, rs.flag, off, rs.cigar, max_end=rs.max_end))] + sets[1:]
    readstore.save(path, bam, broken)
    assert readstore.open_if_fresh(path, bam) is None
    readstore.save(path, bam, sets)
    assert readstore.open_if_fresh(path, bam) is not None
