#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace directory of tools/e2e_profile.py: the LAST `process` call's copies and
kernels on one time axis.   tools/e2e_timeline.py <dir>"""
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
# the last call = everything after the longest pause between two kernel launches
gaps = [(kernels[i + 1][0] - kernels[i][1], i) for i in range(len(kernels) - 1)]
cut = kernels[max(gaps)[1] + 1][0] if gaps else kernels[0][0]
first_spl = min(k[0] for k in kernels if k[0] >= cut)
copies = [c for c in copies if c[0] >= first_spl - 50_000_000]   # (the site table goes up just before the first kernel)
kernels = [k for k in kernels if k[0] >= cut]
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
print("last process() call on the device: %.1f ms from the first copy/kernel to the last" % ((t1 - t0) / 1e6))
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
big = [c for c in h2d if c[3] >= (8 << 20)]
if big:
    rates = sorted(c[3] / (c[1] - c[0]) for c in big)
    print("H2D pieces >= 8 MiB as they ran: %d, rate min / median / max = %.1f / %.1f / %.1f GB/s (%.1f MB in %.2f ms = %.1f GB/s overall)" % (
        len(big), rates[0], rates[len(rates) // 2], rates[-1], sum(c[3] for c in big) / 1e6, sum(c[1] - c[0] for c in big) / 1e6,
        sum(c[3] for c in big) / sum(c[1] - c[0] for c in big)))
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
for (a, b), name in sorted(events, key=lambda e: e[0][0])[:80]:
    print("  %9.2f .. %9.2f  %s" % ((a - t0) / 1e6, (b - t0) / 1e6, name))
