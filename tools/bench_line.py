#!/usr/bin/env python3
"""One-line digest of bench.py JSON logs:  tools/bench_line.py <log> [<log> ...]"""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d["roofline"]
        alone = r.get("alone") or {}
        e2e = d.get("e2e") or []
        print("%s | %s | ms/step %.4f | range kernel %.4f ms (frac %.3f), alone %.4f (%.3f) | path frac %.3f | reads/s %.3g | parity %s | literal %s%s" % (
            path.split("/")[-1], d["config"]["workload"].split(":")[0], d["ms_per_step"], r["kernel_ms_avg"], r["frac"],
            alone.get("kernel_ms_avg", float("nan")), alone.get("frac", float("nan")), r.get("path", {}).get("frac", float("nan")),
            d["reads_per_sec"], (d.get("parity") or {}).get("bit_exact_vs_oracle"), d.get("literal_kernel_reads"),
            "".join(" | e2e %s %.1f M reads/s" % (e["workload"], e["reads_per_sec"] / 1e6) for e in e2e)))
