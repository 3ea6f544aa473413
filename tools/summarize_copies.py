#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py's streamed pass ON THE GPU BOX: every
host-to-device copy of a witness (>= 1 MiB) with its start, duration and rate, how many of them overlap, and per 250 ms
window the copy time, the kernel busy time and the number of level-1 launches (a proxy for proofs per window).
Usage: python tools/summarize_copies.py <dir with run_results.db> <out.txt>"""
import os
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    path = None
    for root, _, files in os.walk(src):
        for f in files:
            if f.endswith(".db"):
                path = os.path.join(root, f)
    c = sqlite3.connect(path)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    lines = []
    mc = [n for n in names if "memory_cop" in n and not n.startswith("rocpd_")]
    lines.append("views: %s" % ", ".join(n for n in names if not n.startswith("rocpd_")))
    if not mc:
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        return
    view = mc[0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    lines.append("%s columns: %s" % (view, ", ".join(cols)))
    size_col = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    rows = c.execute("select name, start, end, %s from %s order by start" % (size_col or "0", view)).fetchall()
    big = [(n, s, e, b) for n, s, e, b in rows if (b or 0) >= (1 << 20)]
    small = [(n, s, e, b) for n, s, e, b in rows if (b or 0) < (1 << 20)]
    lines.append("copies: %d (>= 1 MiB: %d)" % (len(rows), len(big)))
    if not big:
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        return
    t0 = big[0][1]
    import statistics as st
    d = [(e - s) / 1e6 for _, s, e, _ in big]
    lines.append("big copies: duration ms min %.2f med %.2f p90 %.2f max %.2f; rate GB/s med %.1f" % (
        min(d), st.median(d), sorted(d)[int(len(d) * 0.9)], max(d), st.median(b / (e - s) for _, s, e, b in big)))
    ds = [(e - s) / 1e3 for _, s, e, _ in small]
    if ds:
        lines.append("small copies: %d, duration us med %.1f p99 %.1f max %.1f" % (len(ds), st.median(ds), sorted(ds)[int(len(ds) * 0.99)], max(ds)))
        kinds = {}
        for n, s, e, b in small:
            kinds.setdefault(n, []).append((e - s) / 1e3)
        for n, v in kinds.items():
            lines.append("   %s: %d, med %.1f us, max %.1f us" % (n, len(v), st.median(v), max(v)))
    # windows of 250 ms over the big-copy span
    kern = c.execute("select name, start, end from kernels order by start").fetchall()
    W = 250e6
    tend = big[-1][2]
    nwin = int((tend - t0) / W) + 1
    copy_ms = [0.0] * nwin
    ncopy = [0] * nwin
    busy = [0.0] * nwin
    l1 = [0] * nwin
    for _, s, e, _ in big:
        i = int((s - t0) / W)
        if 0 <= i < nwin:
            copy_ms[i] += (e - s) / 1e6
            ncopy[i] += 1
    for n, s, e in kern:
        i = int((s - t0) / W)
        if 0 <= i < nwin:
            busy[i] += (e - s) / 1e6
            if "msm_accum_seg_kernel<true>" in n or "msm_accum_seg_kernel<1" in n:
                l1[i] += 1
    lines.append("window(250ms)  big_copies  copy_ms  kernel_ms_sum  level1_launches")
    for i in range(nwin):
        lines.append("%4d %4d %8.1f %9.1f %5d" % (i, ncopy[i], copy_ms[i], busy[i], l1[i]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
    print("...")


if __name__ == "__main__":
    main()
