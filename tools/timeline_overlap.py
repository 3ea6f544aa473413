#!/usr/bin/env python3
"""What bounds the throughput with several proofs in flight? Reads the rocpd database of
  rocprofv3 --kernel-trace -d <dir> -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-stream-pass
and reports, over the steady part of the run (the last 60 % of the kernel launches by time):
  * the fraction of wall time with 0, 1, 2, 3, 4+ kernels resident on the GPU (union of [start, end) intervals);
  * sum of kernel durations per proof against the same sum measured with one proof in flight (how much the kernels
    stretch when they share the chip);
  * the kernels with the largest share of the time during which they were the ONLY kernel on the chip.
Usage: python tools/timeline_overlap.py <dir-with-run_results.db> [proofs-in-window]"""
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
    d = sys.argv[1]
    path = d if d.endswith(".db") else sorted(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True))[0]
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + 0.4 * (t1 - t0)
    rows = [r for r in rows if r[1] >= lo]
    ev = []
    for name, s, e in rows:
        ev.append((s, 1, name))
        ev.append((e, -1, name))
    ev.sort(key=lambda x: (x[0], x[1]))
    depth, last = 0, ev[0][0]
    hist = collections.Counter()
    alone = collections.Counter()
    active = collections.Counter()
    for t, dlt, name in ev:
        hist[min(depth, 4)] += t - last
        if depth == 1:
            only = [k for k, v in active.items() if v > 0]
            if only:
                alone[short(only[0])] += t - last
        last = t
        depth += dlt
        active[name] += dlt
    wall = ev[-1][0] - ev[0][0]
    print("window %.1f ms, %d kernel launches" % (wall / 1e6, len(rows)))
    for k in range(5):
        print("  %s kernels resident: %5.1f %% of the time" % (("%d" % k) if k < 4 else "4+", 100.0 * hist[k] / wall))
    tot = sum(e - s for _, s, e in rows)
    print("sum of kernel durations / wall = %.2f" % (tot / wall))
    by = collections.Counter()
    cnt = collections.Counter()
    for name, s, e in rows:
        by[short(name)] += e - s
        cnt[short(name)] += 1
    print("%-44s %8s %10s %12s" % ("kernel", "calls", "avg ms", "alone ms"))
    for k, v in by.most_common(14):
        print("%-44s %8d %10.3f %12.2f" % (k[:44], cnt[k], v / cnt[k] / 1e6, alone[k] / 1e6))


if __name__ == "__main__":
    main()
