#!/usr/bin/env python3
"""Imbalance between the waves of a range-kernel workgroup, from a dump of a build with -DSPL_PHASE_TIMING -DSPL_PHASE_WAVES
-DSPL_PHASE_SET=0xc1: when the first / last wave left its loop, when the last wave finished its list, the longest list."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
raw = raw[raw[:, 7] > 0]
t = raw.astype(np.int64)
us = lambda a, b: (t[:, b] - t[:, a]) / 100.0   # noqa: E731
rows = (("start -> first wave leaves its loop", 0, 1), ("first -> last wave leaves its loop", 1, 3),
        ("last loop end -> last twice-spliced list end", 3, 5), ("-> last once-spliced pass end", 5, 4), ("last loop end -> last list end", 3, 4), ("last list end -> workgroup end (barrier, handover, flush)", 4, 7), ("lifetime", 0, 7))
for name, a, b in rows:
    d = us(a, b)
    print("  %-58s mean %6.2f us   p50 %6.2f   p90 %6.2f" % (name, d.mean(), np.percentile(d, 50), np.percentile(d, 90)))

