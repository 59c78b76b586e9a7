"""bench.py's last stdout line is what the driver parses: it has to be ONE short JSON line whatever the run measured
(round 3's 23 KB line was cut off by the driver's 8 KB tail and nothing of the round's measurement survived)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def canned(n_legs=9, world=1):
    leg = {"workload": "human", "bam": "seq-like", "bam_bytes": 14341234567, "decode": "device", "devices": list(range(world)),
           "wall_s": 0.43312345678, "median_wall_s": 0.4434567891, "first_call_wall_s": 0.80123456, "first_call_device_gb": 13.41, "reads_per_sec": 461234567.891,
           "path": {"what": "prose " * 60, "file_GBps": 33.123456789, "frac_of_pcie": 0.59123456, "pcie_peak_GBps": 56.0},
           "tsv_matches_oracle": True, "cpu_e2e": {"what": "prose " * 40, "reads_per_sec": 40912345.678},
           "what": "prose " * 200, "stages": {"a_s": 0.1, "b_s": 0.2}, "cold_cli_s": 0.7123456, "cold_cli_matches": True,
           "kernels": {"what": "prose " * 50, "table": [{"kernel": "spl_k%d" % k, "calls": 5, "ms": 1.2345678, "bytes": 123456789} for k in range(9)]}}
    return {
        "metric": "splice sites/sec (+ reads/sec) processed", "value": 334710996.8159196, "unit": "splice sites/s",
        "reads_per_sec": 222796814793.11566, "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.8976789016742259,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32/u32 counters, f64 SSE", "data": "synthetic",
        "config": {"workload": "human: 200000000 reads x 300463 splice sites, 24 chromosomes in 2 shard(s) on rank 0, 150 bp, unstranded",
                   "scale": 1.0, "parallelism": "chromosome/sample shards, no collectives", "seed": 3, "step": "BAM-native arrays resident in HBM -> layout + range + literal + scan/SSE kernels of every shard, barrier between steps"},
        "roofline": {"bound": "hbm", "achieved": 4640.32429710635, "peak": 8000.0, "unit": "GB/s", "frac": 0.5800405371382937,
                     "traffic": 1493338546, "traffic_from": "r03P_traffic.json", "kernel": "spl_count_ranges_kernel",
                     "kernel_ms_avg": 0.4034253999590874, "launches_timed": 40, "algorithmic_bytes_per_launch": 1872024685.5,
                     "layout": {"kernel": "spl_layout_kernel", "kernel_ms_avg": 0.6123456789, "launches_timed": 40, "algorithmic_bytes_per_launch": 3120000000.5,
                                "arrays_read": 1870000000.5, "records_written": 1250000000.0, "achieved": 5095.123456, "frac": 0.636890432, "what": "prose " * 20},
                     "range": {"kernel": "spl_count_ranges_kernel", "kernel_ms_avg": 0.4034253999590874, "launches_timed": 40,
                               "algorithmic_bytes_per_launch": 1872024685.5, "achieved": 4640.32429710635, "frac": 0.5800405371382937, "what": "prose " * 20},
                     "count_only": {"what": "prose " * 20, "ms_per_step": 0.8976789016742259, "reads_per_sec": 222796814793.11566, "path_frac": 0.5213514214293551},
                     "range_traffic": 1493338546, "range_traffic_from": "r03P_traffic.json", "second_kernel": "spl_count_ranges_kernel", "fused": True,
                     "two_kernels": {"ms_per_step": 2.3104123456, "layout_kernel_ms_avg": 0.669312345, "range_kernel_ms_avg": 0.388223456, "layout_bytes_per_launch": 3130000000.5,
                                     "same_counters": True, "steps": 10, "what": "prose " * 30, "layout_frac": 0.58528123, "range_frac": 0.60276123, "path_frac": 0.20256123},
                     "grid": 15856, "block": 256, "lds_bytes": 17404,
                     "alone": {"kernel_ms_avg": 0.39, "frac": 0.5949139256781943, "how": "prose " * 20},
                     "path": {"what": "prose " * 20, "achieved": 4170.8, "frac": 0.5213514214293551},
                     "hbm_actual": {"what": "prose " * 20, "GBps": 3701.6, "frac": 0.4627059135813722},
                     "lib_sha16": "c6c68c61bd907ffa", "kernel_src_sha16": "69beb4938a3f2c1f", "ingest_src_sha16": "b8c7ddf82408e0e2"},
        "cpu_baseline": {"value": 69925.73301240719, "unit": "splice sites/s", "reads_per_sec": 46545320.39712524, "cores": 1,
                         "kind": "port", "sample": "prose " * 40, "sample_short": "whole workload, one pass of oracle/spliser_oracle.c, 4.3 s",
                         "all_cores": {"value": 256805.8, "reads_per_sec": 170940063.4, "nproc": 256, "cpu_quota_cores": 16.0,
                                       "threads_used": 16, "threads": "prose " * 20, "seconds": 1.17, "same_counts_as_1_thread": True},
                         "reference_cost_model": {"estimate_seconds": 585.8379056, "label": "prose " * 40},
                         "reference_measured": {"label": "prose " * 40, "cases": [{"name": "x", "reads": 1}] * 4}},
        "parity": {"reads": 200000000, "sites": 300463, "bit_exact_vs_oracle": True, "of": "the last timed step", "checked": "prose " * 10},
        "e2e": [dict(leg) for _ in range(n_legs)],
        "combine": {"line": {"workload": "combine6: 6 x 20 M reads", "process_s": 1.234567, "combine_s": 2.345678, "parse_s": 0.3,
                             "merge_s": 0.4, "gapfill_s": 0.5, "write_s": 0.6, "queries": 123456, "rows": 1500000, "matches_oracle": True},
                    "what": "prose " * 100},
        "other_steps": [{"workload": "mouse_stranded fr+cryptic", "reads": 100000000, "sites": 229000, "shards": 2, "stranded": "fr", "cryptic": True, "fused": True,
                         "ms_per_step": 1.0871234, "reads_per_sec": 9.2e10, "kernel_ms_avg": 0.48412345, "frac": 0.2421234, "path_frac": 0.2151234, "bit_exact_vs_oracle": True},
                        {"workload": "arabidopsis", "reads": 20000000, "sites": 180000, "shards": 1, "stranded": None, "cryptic": False, "fused": True,
                         "ms_per_step": 0.3011234, "reads_per_sec": 6.6e10, "kernel_ms_avg": 0.2211234, "frac": 0.2101234, "path_frac": 0.1561234, "bit_exact_vs_oracle": True},
                        {"workload": "human clipped30", "reads": 200000000, "sites": 300463, "shards": 2, "stranded": None, "cryptic": False, "fused": True,
                         "ms_per_step": 2.2011234, "reads_per_sec": 9.1e10, "kernel_ms_avg": 1.0211234, "frac": 0.2301234, "path_frac": 0.2101234, "bit_exact_vs_oracle": True}],
        "imbalance": None if world == 1 else {"reads_per_rank": [25000000] * world, "seconds_per_rank": [0.123456] * world,
                                              "max_over_mean_reads": 1.2345678},
        "literal_kernel_reads": 41234, "gen_seconds": 12.3, "upload_seconds": 1.2}


def check(text, want_legs):
    assert "\n" not in text
    assert len(text.encode("utf-8")) < bench.LINE_LIMIT, len(text)
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["config"]["workload"].startswith("human")
    assert "model" not in d["config"]
    assert d["parity"]["bit_exact_vs_oracle"] is True
    assert len(d["e2e"]) == want_legs
    for leg in d["e2e"]:
        assert all(not isinstance(v, (dict, list)) for v in leg.values()), leg
    return d


def test_compact_line_is_short_and_parses():
    d = check(bench.compact_line(canned(n_legs=7)), 7)
    assert abs(d["value"] - 334710996.8159196) / 334710996.8159196 < 1e-4       # (5 significant digits)
    assert d["e2e"][0]["frac_of_pcie"] == 0.59123 and d["e2e"][0]["cpu_e2e_reads_per_sec"] == 40912000.0
    assert d["combine"]["matches_oracle"] is True
    assert d["roofline"]["fused"] is True and d["roofline"]["two_kernels"]["same_counters"] is True
    assert "what" not in d["roofline"]["two_kernels"] and d["e2e"][0]["first_call_device_gb"] == 13.41
    # the other configurations' resident steps ride in the line: config 5's shape, config 2, the clipped sample
    assert [o["workload"] for o in d["other_steps"]] == ["mouse_stranded fr+cryptic", "arabidopsis", "human clipped30"]
    assert all(o["bit_exact_vs_oracle"] is True and o["ms_per_step"] and o["frac"] for o in d["other_steps"])


def test_compact_line_eight_ranks():
    d = check(bench.compact_line(canned(n_legs=4, world=8)), 4)
    assert len(d["imbalance"]["reads_per_rank"]) == 8


def test_compact_line_sheds_before_it_breaks():
    text = bench.compact_line(canned(n_legs=30))
    assert len(text) < bench.LINE_LIMIT
    d = json.loads(text)
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"] and d["e2e_dropped_for_length"] > 0


def test_no_baseline_no_e2e():
    out = canned(0)
    out["cpu_baseline"], out["e2e"], out["parity"] = None, None, None
    d = json.loads(bench.compact_line(out))
    assert d["cpu_baseline"] is None and d["e2e"] is None


def test_emit_prints_the_compact_line_last(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(canned(n_legs=5))
    lines = capsys.readouterr().out.strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("bench_detail {")
    check(lines[-1], 5)
    with open(tmp_path / "bench_detail.json") as fh:
        assert json.load(fh)["e2e"][0]["kernels"]["table"]


def test_the_committed_traffic_measurement_is_of_these_kernel_sources():
    """bench.py quotes `roofline.traffic` (HBM bytes per launch by the PMC counters) from profiles/*_traffic.json only when the file was
    measured with the kernel sources it runs (kernel_src_sha16: comments and white space aside).  A change to spl_kernels.hip,
    spl_device.h, spl_pack.h, spl_classify.h or spl_pack.cpp without a new `tools/prof_round.sh` run leaves the driver's line with
    `traffic: null` -- which is how round 4 nearly went out (an experiment's switch added after the profile run)."""
    import glob
    sha = bench.kernel_src_sha16()
    have = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_traffic.json")):
        with open(path) as fh:
            t = json.load(fh)
        if t.get("workload") == "human":
            have.append(t.get("kernel_src_sha16"))
    assert sha in have, "no profiles/*_traffic.json for the human workload was measured with the current kernel sources (%s): run tools/prof_round.sh" % sha
