#!/usr/bin/env python3
"""Timeline of the range kernel from a SPL_PHASE_TIMING dump (see spl_kernels.hip): per-workgroup phase durations
(100 MHz wall clock -> microseconds), how many workgroups were resident at a time and where (XCD / CU)."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
raw = raw[raw[:, 7] > 0]
hw = raw[:, 2].copy()
t = raw.astype(np.int64)
t[:, 2] = 0  # slot 2 carries HW_ID / XCC_ID, not a stamp
have = [k for k in range(8) if (t[:, k] > 0).all()]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
span = us[:, 7].max()
label = {0: "start", 1: "loads ahead issued, LDS zeroed, barrier passed", 3: "read 1 begins", 4: "read 2 begins", 5: "read 3 begins",
         6: "last read done (wave 0)", 7: "barrier, queue handed over, LDS flushed"}
if len(sys.argv) > 2 and sys.argv[2] == "tail":  # built with -DSPL_PHASE_TAIL: slots 3, 4 are epilogue stamps
    label.update({3: "barrier after the loop passed", 4: "queue handed over", 7: "LDS flushed"})
    order = [0, 1, 6, 3, 4, 7]
    have = [k for k in order if k in have]
print("workgroups %d   kernel span %.1f us   stamps %s" % (len(t), span, have))
for a, b in zip(have[:-1], have[1:]):
    d = us[:, b] - us[:, a]
    print("  %-62s mean %7.2f us   p50 %7.2f   p90 %7.2f" % ("-> " + label[b], d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
life = us[:, 7] - us[:, 0]
print("  %-44s mean %7.2f us   p50 %7.2f   p90 %7.2f" % ("workgroup lifetime", life.mean(), np.percentile(life, 50), np.percentile(life, 90)))
print("  mean resident workgroups %.1f (= sum of lifetimes / span)" % (life.sum() / span))
# residency over time
edges = np.linspace(0, span, 11)
print("  resident workgroups at 10%% steps of the span: %s" % " ".join(
    "%d" % int(((us[:, 0] <= x) & (us[:, 7] > x)).sum()) for x in edges[1:-1]))
hwid = (hw & 0xffffffff).astype(np.int64)
xcc = ((hw >> 32) & 0xf).astype(np.int64)
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
where = xcc * 1000 + se * 100 + sh * 10 + cu  # one number per physical CU
ids, counts = np.unique(where, return_counts=True)
print("  CUs seen %d   workgroups per CU min %d / mean %.1f / max %d" % (len(ids), counts.min(), counts.mean(), counts.max()))
print("  workgroups per XCD: %s" % " ".join("%d" % int((xcc == x).sum()) for x in range(8)))
mid = span / 2
res = np.array([int(((where == c) & (us[:, 0] <= mid) & (us[:, 7] > mid)).sum()) for c in ids])
print("  resident per CU at mid-span: min %d / mean %.2f / max %d" % (res.min(), res.mean(), res.max()))
busy = np.array([life[where == c].sum() for c in ids])
print("  per-CU sum of lifetimes / span: min %.2f / mean %.2f / max %.2f" % (busy.min() / span, busy.mean() / span, busy.max() / span))
last = np.array([us[where == c, 7].max() for c in ids])
print("  per-CU last workgroup ends at: min %.1f / mean %.1f / max %.1f us" % (last.min(), last.mean(), last.max()))
