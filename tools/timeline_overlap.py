#!/usr/bin/env python3
"""What bounds the throughput with several proofs in flight? Reads the rocpd database of
  rocprofv3 --kernel-trace -d <dir> -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-stream-pass
and reports, over the timed steps (the densest stretch of the trace, without its ramp-up and drain):
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
    # the timed steps = the densest stretch of the trace: 50 ms bins with more than half of the busiest bin's launches
    t0 = rows[0][1]
    bins = collections.Counter((s - t0) // 50_000_000 for _, s, _ in rows)
    top = max(bins.values())
    dense = sorted(b for b, v in bins.items() if v > top / 2)
    lo, hi = t0 + (dense[0] + 1) * 50_000_000, t0 + dense[-1] * 50_000_000  # drop the ramp-up and drain bins
    rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
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
        hist[min(depth, 16)] += t - last
        if depth == 1:
            only = [k for k, v in active.items() if v > 0]
            if only:
                alone[short(only[0])] += t - last
        last = t
        depth += dlt
        active[name] += dlt
    wall = ev[-1][0] - ev[0][0]
    print("window %.1f ms, %d kernel launches" % (wall / 1e6, len(rows)))
    for k in sorted(hist):
        print("  %2d kernels resident: %5.1f %% of the time" % (k, 100.0 * hist[k] / wall))
    tot = sum(e - s for _, s, e in rows)
    print("sum of kernel durations / wall = %.2f" % (tot / wall))
    by = collections.Counter()
    cnt = collections.Counter()
    for name, s, e in rows:
        by[short(name)] += e - s
        cnt[short(name)] += 1
    proofs = max(1, cnt.get("expr_eval_limbs_kernel", 1))
    print("proofs in the window: %d  (%.2f ms per proof)" % (proofs, wall / 1e6 / proofs))
    print("%-44s %8s %10s %22s" % ("kernel", "calls", "avg ms", "sum per proof, ms"))
    for k, v in by.most_common(16):
        print("%-44s %8d %10.3f %22.2f" % (k[:44], cnt[k], v / cnt[k] / 1e6, v / 1e6 / proofs))


if __name__ == "__main__":
    main()
