#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV of a multi-stream run into a timeline summary:
GPU-busy fraction, how many kernels execute concurrently, per-kernel summed / average durations.
Usage: python tools/timeline.py <kernel_trace.csv> [window_ms] > profiles/<tag>_timeline.txt"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    t1 = max(e[1] for e in ev)
    sel = [e for e in ev if e[0] >= t1 - window * 1e6]
    busy, cs, ce = 0, None, None
    for s, e, _ in sel:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    span = sel[-1][1] - sel[0][0]
    print("window: last %.1f ms of the run; GPU busy (union of kernel intervals) %.1f %%" % (span / 1e6, 100 * busy / span))
    pts = []
    for s, e, _ in sel:
        pts += [(s, 1), (e, -1)]
    pts.sort()
    lvl, last, hist = 0, pts[0][0], collections.Counter()
    for t, d in pts:
        hist[lvl] += t - last
        last = t
        lvl += d
    print("fraction of the window with N kernels executing:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print("%-46s %6s %10s %10s %7s %9s" % ("kernel", "calls", "sum_ms", "avg_ms", "%sum", "%of_wall"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print("%-46s %6d %10.3f %10.4f %6.1f%% %8.1f%%" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e6, 100 * v[1] / tot, 100 * v[1] / span))
    print("sum of kernel durations %.1f ms = %.2f x the window" % (tot / 1e6, tot / span))


if __name__ == "__main__":
    main()
