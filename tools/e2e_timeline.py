#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace directory of tools/e2e_profile.py: the copies and kernels of the
`process` call on one time axis.   tools/e2e_timeline.py <dir> [--last-call]
--last-call: the traced process made several calls and slept a second before the last one (e2e_profile.py --pause 1): only what
lies behind the longest silence in the trace is summarised -- a call of a process that has its context and its memory."""
import csv
import glob
import sys


def load(pattern):
    rows = []
    for path in glob.glob(sys.argv[1] + "/**/" + pattern, recursive=True):
        rows += list(csv.DictReader(open(path)))
    return rows


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    return None


kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]) for r in load("*kernel_trace.csv")]
copies = []
for r in load("*memory_copy_trace.csv"):
    size = col(r, "Bytes", "Size", "bytes")
    copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), col(r, "Direction", "Kind", "Name") or "?", int(size) if size else 0))
kernels.sort()
copies.sort()
if not kernels:
    sys.exit("no kernels in the trace")
if "--last-call" in sys.argv[2:]:
    ev = sorted([(a, b) for a, b, _ in kernels] + [(c[0], c[1]) for c in copies])
    cut, gap, end = ev[0][0], 0, ev[0][1]
    for a, b in ev[1:]:
        if a - end > gap:
            gap, cut = a - end, a
        end = max(end, b)
    kernels = [k for k in kernels if k[0] >= cut]
    copies = [c for c in copies if c[0] >= cut]
    print("(the last call of the traced process: what lies behind a silence of %.2f s)" % (gap / 1e9))
t0 = min([k[0] for k in kernels] + [c[0] for c in copies])
t1 = max([k[1] for k in kernels] + [c[1] for c in copies])


def union(iv):
    out, cur = [], None
    for a, b in sorted(iv):
        if cur and a <= cur[1]:
            cur[1] = max(cur[1], b)
        else:
            cur = [a, b]
            out.append(cur)
    return out


def overlap(u, v):
    i = j = 0
    tot = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if b > a:
            tot += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return tot


ku = union([(a, b) for a, b, _ in kernels])
h2d = [c for c in copies if "HOST_TO_DEVICE" in c[2].upper() or "H2D" in c[2].upper() or "HOSTTODEVICE" in c[2].upper()]
d2h = [c for c in copies if c not in h2d]
hu = union([(a, b) for a, b, _, _ in h2d])
busy = lambda u: sum(b - a for a, b in u)
print("process() on the device: %.1f ms from the first copy/kernel to the last" % ((t1 - t0) / 1e6))
print("kernels: %d launches, %.2f ms busy (union)" % (len(kernels), busy(ku) / 1e6))
by = {}
for a, b, n in kernels:
    e = by.setdefault(n, [0, 0])
    e[0] += 1
    e[1] += b - a
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("    %-40s %5d x %9.1f us = %8.2f ms" % (n[:40], c, t / c / 1e3, t / 1e6))
print("H2D copies: %d, %.1f MB, %.2f ms busy (union); D2H/other copies: %d, %.1f MB" % (
    len(h2d), sum(c[3] for c in h2d) / 1e6, busy(hu) / 1e6, len(d2h), sum(c[3] for c in d2h) / 1e6))
long_ = sorted(c[1] - c[0] for c in h2d if c[1] - c[0] >= 300_000)
if long_:
    print("H2D copies of 0.3 ms and more (the staging ring's 32 MiB pieces; this trace format has no sizes): %d, duration min / median / max "
          "= %.2f / %.2f / %.2f ms -- a full piece at the median = %.1f GB/s (SPL_STAGE_TIMING=1 prints exact rates)" % (
              len(long_), long_[0] / 1e6, long_[len(long_) // 2] / 1e6, long_[-1] / 1e6, (32 << 20) / long_[len(long_) // 2]))
print("copy time with a kernel running beside it: %.2f ms of %.2f ms H2D (%.0f %%); kernel time under a copy: %.0f %%" % (
    overlap(hu, ku) / 1e6, busy(hu) / 1e6, 100.0 * overlap(hu, ku) / max(busy(hu), 1), 100.0 * overlap(hu, ku) / max(busy(ku), 1)))
both = union([(a, b) for a, b in ku] + [(a, b) for a, b in hu] + [(a, b) for a, b, _, _ in d2h])
print("device idle (no copy, no kernel) inside the call: %.1f ms of %.1f ms -- the host side (BAM decode, packing) is the clock" % (
    ((t1 - t0) - busy(both)) / 1e6, (t1 - t0) / 1e6))
print("\ntimeline (ms from the first event): H2D bursts and kernel bursts, merged when less than 0.2 ms apart")
events = []
for name, u in (("H2D", hu), ("kernels", ku)):
    cur = None
    for a, b in u:
        if cur and a - cur[1] < 200_000:
            cur[1] = b
        else:
            cur = [a, b]
            events.append((cur, name))
for (a, b), name in sorted(events, key=lambda e: e[0][0])[:240]:
    print("  %9.2f .. %9.2f  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, name))
