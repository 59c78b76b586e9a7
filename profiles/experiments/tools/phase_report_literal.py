#!/usr/bin/env python3
"""Timeline of the literal kernel from a SPL_PHASE_LITERAL dump: per-wave stamps (100 MHz wall clock -> microseconds)."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
raw = raw[raw[:, 7] > 0]
t = raw[:, :4].astype(np.int64)
t0 = t[:, 0].min()
work = t[:, 1] > 0
print("waves %d, with work %d" % (len(t), work.sum()))
us = (t[work] - t0) / 100.0
print("kernel span %.1f us; waves start at p50 %.1f / p90 %.1f / max %.1f us" % (us[:, 3].max(), np.percentile(us[:, 0], 50),
                                                                           np.percentile(us[:, 0], 90), us[:, 0].max()))
for name, a, b in (("queue counters + index + packed words", 0, 1), ("ops walked once (extent)", 1, 2), ("rival tables, updates", 2, 3), ("lifetime", 0, 3)):
    d = us[:, b] - us[:, a]
    print("  %-40s mean %7.2f us   p50 %7.2f   p90 %7.2f   max %7.2f" % (name, d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max()))
paths = raw[work][:, 4:7].astype(np.int64)
life = us[:, 3] - us[:, 0]
print("reads by path: table %d, closed form %d, general walk %d" % tuple(paths.sum(axis=0)))
order = np.argsort(-life)[:12]
print("slowest waves: lifetime us / table / closed form / general")
for k in order:
    print("   %6.1f  %3d %3d %3d" % (life[k], paths[k, 0], paths[k, 1], paths[k, 2]))
for name, sel in (("waves with only table-path reads", (paths[:, 1] + paths[:, 2]) == 0), ("waves with a closed-form read", paths[:, 1] > 0),
                  ("waves with a general-walk read", paths[:, 2] > 0)):
    if sel.any():
        print("  %-36s n %5d  lifetime mean %6.2f  p90 %6.2f  max %6.2f" % (name, sel.sum(), life[sel].mean(), np.percentile(life[sel], 90), life[sel].max()))
