#!/usr/bin/env python3
"""One proof alone on the GPU, kernel by kernel: reads the rocpd database of
  rocprofv3 --kernel-trace -d <dir> -o run -- python3 bench.py --steps 1 --warmup 1 --concurrency 1 --regions 1 \
      --no-cpu-baseline --no-stream-pass --no-serial-latency --no-k22
takes the LAST whole proof made with the default (lanes) key — the interval between the starts of the last two
random-polynomial kernels, the first kernel of every proof — and reports
  * its wall time and the fraction of it with 0, 1, 2, 3+ kernels resident (union of [start, end) intervals),
  * sum of kernel durations / wall,
  * per stream: busy time,
  * with -v: every dispatch (start offset, duration, stream, grid, name).
Usage: python tools/timeline_single_proof.py <dir-or-db> [-v] [proof-index-from-the-end, default 1]"""
import collections
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv
    d = args[0]
    back = int(args[1]) if len(args) > 1 else 1
    path = d if d.endswith(".db") else sorted(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True))[0]
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, stream_id, grid_x * grid_y * grid_z / "
                     "(workgroup_x * workgroup_y * workgroup_z), queue_id from kernels order by start").fetchall()
    marks = [s for n, s, e, st, g, q in rows if "chacha20_fr_random" in n]
    if len(marks) < back + 1:
        raise SystemExit("need at least %d random-polynomial kernels in the trace" % (back + 1))
    a, b = marks[-back - 1], marks[-back]
    sel = [r for r in rows if a <= r[1] < b]
    end = max(r[2] for r in sel)
    wall = end - a
    ev = []
    for n, s, e, st, g, q in sel:
        ev += [(s, 1), (e, -1)]
    ev.sort()
    depth, last, hist = 0, a, collections.Counter()
    for t, dl in ev:
        hist[min(depth, 4)] += t - last
        last = t
        depth += dl
    print("one proof: %d dispatches on %d streams, first kernel start -> last kernel end %.3f ms" % (len(sel), len({r[3] for r in sel}), wall / 1e6))
    for k in sorted(hist):
        print("  %s kernels resident: %5.1f %% of the time" % (("%d+" % k) if k == 4 else str(k), 100.0 * hist[k] / wall))
    print("  >= 2 kernels resident: %.1f %%" % (100.0 * sum(v for k, v in hist.items() if k >= 2) / wall))
    tot = sum(e - s for n, s, e, st, g, q in sel)
    print("sum of kernel durations %.3f ms = %.2f x the wall time" % (tot / 1e6, tot / wall))
    per_stream = collections.Counter()
    for n, s, e, st, g, q in sel:
        per_stream[st] += e - s
    for st, v in per_stream.most_common():
        print("  stream %-4s busy %.3f ms" % (st, v / 1e6))
    by = collections.Counter()
    cnt = collections.Counter()
    for n, s, e, st, g, q in sel:
        by[short(n)] += e - s
        cnt[short(n)] += 1
    print("%-40s %6s %10s" % ("kernel", "calls", "sum ms"))
    for k, v in by.most_common(14):
        print("%-40s %6d %10.3f" % (k[:40], cnt[k], v / 1e6))
    if verbose:
        print("%9s %8s %6s %5s %8s  %s" % ("start ms", "dur ms", "stream", "queue", "wgs", "kernel"))
        for n, s, e, st, g, q in sel:
            print("%9.3f %8.3f %6s %5s %8d  %s" % ((s - a) / 1e6, (e - s) / 1e6, st, q, g, short(n)[:60]))


if __name__ == "__main__":
    main()
