"""The BASELINE.json configurations that are not the default bench line, at a scale a test can afford, through the product's
own entry points (CLI functions, BAM files, several contexts on one GPU standing in for several GPUs) against the oracle:

  configs[2]  human-scale genome (24 chromosomes, two shards of the int32 coordinate space), unstranded
  configs[3]  six-sample `combine` over an A. thaliana-like genome, chromosomes / samples dealt to several devices
  configs[4]  mouse-like genome, paired flags 99/147/83/163, --isStranded -s fr --beta2Cryptic, several devices
plus what the streaming ingest must survive: a BAM that is not sorted by reference, reads that reach beyond the reference
length of the header."""
import os

import numpy as np
import pytest

import helpers
from spliser_amd import cli, combine as cmb, native, process as proc, samio, sites, synth, tsv

pytestmark = pytest.mark.gpu


def _oracle_tsv(oracle_lib, bed, wl, stranded, cryptic, gff=None):
    """The .SpliSER.tsv text the reference's rules give for a workload: host Steps 0-2 (golden-pinned) + the oracle."""
    bins = sites.GeneBins.from_annotation(gff, "gene", "All") if gff else sites.GeneBins()
    table = sites.SiteTable(bins, is_stranded=bool(stranded))
    table.add_bed(bed)
    table.find_competitors()
    scode = native.STRANDED_CODE[stranded]
    text = [tsv.HEADER]
    names = wl.genome.chrom_names
    for chrom in table.chrom_index:
        arr = table.chrom_arrays(chrom)
        if arr.n == 0:
            continue
        rd = wl.reads[names.index(chrom)]
        cnt = oracle_lib.check_bam(arr.pos, arr.strand, arr.part_off, arr.part_pos, arr.comp_off, arr.comp_pos, rd.pos, rd.flag,
                                   rd.cig_off, rd.cigar, scode, 0)
        b2s, b2c, b2w, sse = oracle_lib.beta2_sse(arr.pos, arr.part_off, arr.part_pos, arr.part_site, arr.alpha, arr.edge_cnt,
                                                  cnt[0], cnt[1], cnt[2], cryptic)
        text.extend(tsv.format_chrom(arr, dict(beta1=cnt[0], beta2_simple=b2s, beta2_cryptic=b2c, beta2_weighted=b2w, sse=sse), cryptic))
    return "".join(text)


def _files(wl, prefix, seq_mode=0):
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions)
    synth.write_gff(prefix + ".gff", wl.genome)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, wl.genome.chrom_lengths, wl.reads, level=1, threads=4, seq_mode=seq_mode)


@pytest.mark.parametrize("devices", ["0", "0,0,0"])
def test_config3_human_scale_process(devices, tmp_path, oracle_lib):
    wl = synth.Workload("human", scale=0.01, workers=4)       # 2 M reads, 24 chromosomes of hg38 lengths: two shards
    prefix = str(tmp_path / "h")
    _files(wl, prefix)
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix, "-A", prefix + ".gff", "--devices", devices]) == 0
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, None, False, prefix + ".gff")


def test_config3_on_eight_contexts_decodes_in_eight_shares(tmp_path, oracle_lib, monkeypatch):
    """The target machine has eight GPUs (here: eight contexts on one).  A human-shaped file of 24 chromosomes is cut into eight
    stretches of equal size in file bytes -- anywhere, so chromosomes lie across contexts; every context inflates, extracts and
    counts its own stretch, the partial counters of a cut chromosome are added; the file is the oracle's."""
    wl = synth.Workload("human", scale=0.02, workers=4)       # 4 M reads
    prefix = str(tmp_path / "h8")
    _files(wl, prefix, seq_mode=1)
    seen = {}
    real = native.BamFile.decode_on_devices_async

    def spy(self, devs):
        seen["plan"] = real(self, devs)
        seen["bytes"] = list(self.share_bytes)
        return seen["plan"]
    monkeypatch.setattr(native.BamFile, "decode_on_devices_async", spy)
    tm = proc.process(prefix + ".bam", prefix + ".bed", prefix, annotationFile=prefix + ".gff", log=lambda m: None, devices=(0,) * 8)
    assert tm["bam_decode"] == "device"
    plan = seen["plan"]
    assert len(plan) == 8 and all(names for _, names in plan)
    assert sorted(set(c for _, names in plan for c in names)) == sorted(wl.genome.chrom_names)
    assert max(seen["bytes"]) <= 1.05 * sum(seen["bytes"]) / 8, seen["bytes"]
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, None, False, prefix + ".gff")


@pytest.mark.parametrize("stranded", [None, "fr"])
@pytest.mark.parametrize("n_ctx", [8, 3])
def test_five_chromosomes_on_eight_contexts(tmp_path, oracle_lib, monkeypatch, n_ctx, stranded):
    """whole-reference shares gave an A. thaliana file (five chromosomes) five shares at most.  Shares are stretches of
    the FILE now: `process(devices=(0,) * 8)` makes eight, none empty, every chromosome's reads on two or three contexts, each
    counted against the chromosome's whole table, the partial beta1 / beta2Simple / double counts added on the host before
    findBeta2Counts + calculateSSE run once on the sums -- and the .SpliSER.tsv is the oracle's, unstranded and `fr` + cryptic."""
    wl = synth.Workload("arabidopsis", scale=0.05, seed=23, workers=4)       # 1 M reads
    if stranded:
        for r in wl.reads:       # paired-end flags, so that both strands' windows are used
            r.flag[:] = np.random.default_rng(5).choice(np.array([99, 147, 83, 163], np.uint16), size=r.n)
    prefix = str(tmp_path / "a8")
    _files(wl, prefix, seq_mode=1)
    seen = {}
    real = native.BamFile.decode_on_devices_async
    real_join = native.BamFile.join_decoders

    def spy(self, devs):
        seen["plan"] = real(self, devs)
        return seen["plan"]

    def spy_join(self):
        ok = real_join(self)
        if ok and "held" not in seen and getattr(self, "shares", None):
            seen["held"] = [[self.share_ref(k, c)[0] for c in wl.genome.chrom_names] for k in range(len(self.shares))]
        return ok
    monkeypatch.setattr(native.BamFile, "decode_on_devices_async", spy)
    monkeypatch.setattr(native.BamFile, "join_decoders", spy_join)
    tm = proc.process(prefix + ".bam", prefix + ".bed", prefix, annotationFile=prefix + ".gff", log=lambda m: None, devices=(0,) * n_ctx,
                      isStranded=bool(stranded), strandedType=stranded, isbeta2Cryptic=bool(stranded))
    assert tm["bam_decode"] == "device"
    assert len(seen["plan"]) == n_ctx
    held = np.array(seen["held"])
    assert held.shape == (n_ctx, 5) and np.all(held.sum(axis=1) > 0)                     # no share without reads
    assert held.sum(axis=0).tolist() == [r.n for r in wl.reads]                         # every read on exactly one context
    assert np.any((held > 0).sum(axis=0) > 1)                                            # chromosomes ARE cut
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, stranded, bool(stranded), prefix + ".gff")


@pytest.mark.parametrize("gpu_decode", [False, True])
@pytest.mark.parametrize("stranded", [None, "fr"])
def test_large_read_set_chunks(tmp_path, oracle_lib, monkeypatch, stranded, gpu_decode):
    """Read sets of 64 M reads and more are cut into chunks of 4096 reads (other instantiations of the range kernel, longer
    per-wave lists, another queue entry format): forced here on a sample a test can afford, host and device decode."""
    monkeypatch.setenv("SPL_FORCE_CHUNK", "4096")
    wl = synth.Workload("mouse_stranded" if stranded else "human", scale=0.01, workers=4)
    prefix = str(tmp_path / "b")
    _files(wl, prefix, seq_mode=1)
    argv = ["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix]
    argv += ["--isStranded", "-s", "fr", "--beta2Cryptic"] if stranded else []
    argv += ["--gpuDecode"] if gpu_decode else []
    assert cli.main(argv) == 0
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, stranded, bool(stranded))


@pytest.mark.parametrize("devices", ["0", "0,0"])
def test_config5_mouse_stranded_cryptic_process(devices, tmp_path, oracle_lib):
    wl = synth.Workload("mouse_stranded", scale=0.02, workers=4)    # 2 M reads with flags 99/147/83/163
    assert set(np.unique(np.concatenate([r.flag for r in wl.reads]))) == {83, 99, 147, 163}
    prefix = str(tmp_path / "m")
    _files(wl, prefix, seq_mode=1)
    argv = ["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix, "--isStranded", "-s", "fr", "--beta2Cryptic",
            "--devices", devices]
    assert cli.main(argv) == 0
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, "fr", True)


@pytest.mark.parametrize("devices", ["0", "0,0,0,0", "0,0,0,0,0,0,0,0"])
def test_config4_six_sample_combine(devices, tmp_path, oracle_lib):
    """Six samples of one genome (seeds 11-16: every sample finds its own subset of the rare junctions), `process` each,
    `combine` all; the gap fill against the oracle's counts for the same queries."""
    titles, tsvs, bams = [], [], []
    genome_seed = 7
    for k in range(6):
        wl = synth.Workload("arabidopsis", scale=0.005, seed=genome_seed, workers=2)
        # same genome, another draw of reads: re-sample the reads with the sample's seed
        rb = synth.make_reads(wl.genome, wl.n_reads, seed=11 + k)
        wl.reads = synth.split_by_chrom(rb, len(wl.genome.chrom_names))
        wl.junctions = synth.junction_table([rb])
        prefix = str(tmp_path / ("s%d" % k))
        _files(wl, prefix)
        assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix]) == 0
        titles.append("S%d" % k)
        tsvs.append(prefix + ".SpliSER.tsv")
        bams.append(prefix + ".bam")
    sfile = str(tmp_path / "samples.tsv")
    with open(sfile, "w") as fh:
        for t, a, b in zip(titles, tsvs, bams):
            fh.write("%s\t%s\t%s\n" % (t, a, b))
    assert cli.main(["combine", "-S", sfile, "-o", str(tmp_path / "all"), "--devices", devices]) == 0
    # expectation: the host walk (golden-pinned) with the oracle answering the gap-fill queries
    rows = [cmb._parse_tsv(p) for p in tsvs]
    merged = cmb.merge_sites(rows, cmb.region_order(rows), 6, False, "All")
    results, n_queries = {}, 0
    for idx, queries in cmb.gap_queries(merged).items():
        source = proc.open_alignments(bams[idx])
        table = cmb._QueryTable(queries)
        for chrom in table.chrom_index:
            s, r = table.chrom_arrays(chrom), source.reads(chrom)
            b1, b2, _ = oracle_lib.check_bam(s.pos, s.strand, s.part_off, s.part_pos, s.comp_off, s.comp_pos, r.pos, r.flag, r.cig_off,
                                             r.cigar, 0, 1)
            for j, si in enumerate(table.site_index[chrom]):
                results[(si, idx)] = (int(b1[j]), int(b2[j]))
                n_queries += 1
    want = str(tmp_path / "want.combined.tsv")
    cmb.write_combined(want, merged, titles, results, False)
    assert n_queries > 500
    assert open(str(tmp_path / "all.combined.tsv")).read() == open(want).read()


def test_process_bam_not_sorted_by_reference(tmp_path, oracle_lib):
    """Records of an earlier reference after a later one: chromosomes were counted before they were complete -- the run must
    notice and count again from the complete decode."""
    wl = synth.Workload("arabidopsis", scale=0.01, seed=9, workers=2)
    names, lens = wl.genome.chrom_names, wl.genome.chrom_lengths
    # file order: Chr1 (first half), Chr2..Chr5, Chr1 (second half)
    r1 = wl.reads[0]
    h = r1.n // 2

    def part(rs, a, b):
        return samio.ReadSet(rs.pos[a:b], rs.flag[a:b], rs.cig_off[a:b + 1] - rs.cig_off[a], rs.cigar[int(rs.cig_off[a]):int(rs.cig_off[b])])
    order = [(0, part(r1, 0, h))] + [(i, wl.reads[i]) for i in range(1, 5)] + [(0, part(r1, h, r1.n))]
    prefix = str(tmp_path / "u")
    # a BAM whose reference dictionary lists Chr1 twice under different slots is not possible: write the stretches as
    # separate files and splice the BGZF blocks (concatenated BGZF members are a BGZF file; one header, then records)
    import struct
    import zlib

    def records(tid, rs):
        out = bytearray()
        for k in range(rs.n):
            ops = rs.cigar[int(rs.cig_off[k]):int(rs.cig_off[k + 1])]
            out += struct.pack("<iiiBBHHHiiii", 32 + 2 + 4 * len(ops), tid, int(rs.pos[k]) - 1, 2, 60, 4680, len(ops), int(rs.flag[k]), 0, -1, -1, 0)
            out += b"r\0" + ops.astype("<u4").tobytes()
        return bytes(out)

    def bgzf(raw):
        out = bytearray()
        for a in range(0, len(raw), 0xff00):
            chunk = raw[a:a + 0xff00]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            out += struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(body) + 25)
            out += body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return bytes(out)
    text = "@HD\tVN:1.6\tSO:unsorted\n"
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(names))
    for n, ln in zip(names, lens):
        head += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    with open(prefix + ".bam", "wb") as fh:
        fh.write(bgzf(head))
        for tid, rs in order:
            fh.write(bgzf(records(tid, rs)))
        fh.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    synth.write_bed(prefix + ".bed", names, wl.junctions)
    bam = native.BamFile(prefix + ".bam", stream=True)
    assert bam.wait_all() is False
    assert bam.reads("Chr1").n == r1.n and np.array_equal(bam.reads("Chr1").pos, r1.pos)
    bam.close()
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix, "--threads", "2"]) == 0
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, None, False)


def test_process_reads_beyond_the_reference_length(tmp_path, oracle_lib):
    """The shard is planned from the header's reference lengths while the file decodes; a header that understates them must
    not let one chromosome's reads spill into the next one's coordinates."""
    wl = synth.Workload("arabidopsis", scale=0.005, seed=4, workers=2)
    prefix = str(tmp_path / "b")
    synth.write_bed(prefix + ".bed", wl.genome.chrom_names, wl.junctions)
    native.write_bam(prefix + ".bam", wl.genome.chrom_names, [1000] * 5, wl.reads, level=1, threads=2)   # every reference "1 kb long"
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix]) == 0
    assert open(prefix + ".SpliSER.tsv").read() == _oracle_tsv(oracle_lib, prefix + ".bed", wl, None, False)


def test_process_check_junctions(tmp_path, capsys):
    """--checkJunctions: the junction table of the BAM (device) against the BED file's alpha.  A BED file made from the very
    reads agrees junction by junction; one that claims more reads than the BAM holds is called out."""
    wl = synth.Workload("arabidopsis", scale=0.005, seed=12, workers=2)
    prefix = str(tmp_path / "j")
    _files(wl, prefix)
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix, "--checkJunctions"]) == 0
    rows = [l.rstrip("\n").split("\t") for l in open(prefix + ".junctionCheck.tsv")][1:]
    assert len(rows) == len(wl.junctions[0]) and all(r[6] == "equal" and r[4] == r[5] for r in rows)
    plain = open(prefix + ".SpliSER.tsv").read()
    # tamper: the first junction gets 5 more reads in the BED file than the BAM has, the last one is dropped
    lines = open(prefix + ".bed").read().split("\n")
    f = lines[1].split("\t")
    f[4] = str(int(f[4]) + 5)
    lines[1] = "\t".join(f)
    with open(prefix + ".2.bed", "w") as fh:
        fh.write("\n".join(lines[:-2]) + "\n")
    capsys.readouterr()
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".2.bed", "-o", prefix + "2", "--checkJunctions"]) == 0
    assert "WARNING: 1 junction(s)" in capsys.readouterr().out
    status = [l.rstrip("\n").split("\t")[6] for l in open(prefix + "2.junctionCheck.tsv")][1:]
    assert status.count("bed>bam") == 1 and status.count("bam_only") == 1 and status.count("equal") == len(rows) - 2
    # and the flag changes nothing in the product's own output
    assert cli.main(["process", "-B", prefix + ".bam", "-b", prefix + ".bed", "-o", prefix + "3"]) == 0
    assert open(prefix + "3.SpliSER.tsv").read() == plain
