#!/usr/bin/env python3
"""ON THE GPU BOX: split a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py into its timed regions (separated
by > 15 ms with no kernel running) and print per region: length, proofs (random-polynomial kernels), the fraction of the time
at least one kernel runs, summed kernel time, idle gaps, and the witness uploads (count, mean duration, how long >= 2 / >= 4
of them overlap). Usage: python tools/summarize_regions.py <dir> <out.txt>"""
import os
import sqlite3
import sys


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e in iv:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            tot += cur_e - cur_s
            gaps.append((cur_e, s))
            cur_s, cur_e = s, e
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot, gaps


def overlap_time(iv, k):
    ev = []
    for s, e in iv:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    n, last, tot = 0, None, 0
    for t, d in ev:
        if n >= k and last is not None:
            tot += t - last
        n += d
        last = t
    return tot


def main():
    src, out = sys.argv[1], sys.argv[2]
    path = None
    for root, _, files in os.walk(src):
        for f in files:
            if f.endswith(".db"):
                path = os.path.join(root, f)
    c = sqlite3.connect(path)
    kern = c.execute("select start, end, name from kernels order by start").fetchall()
    cop = c.execute("select start, end, size, name from memory_copies where size >= 1048576 order by start").fetchall()
    iv = [(s, e) for s, e, _ in kern]
    _, gaps = union(iv)
    cuts = [kern[0][0]] + [g[1] for g in gaps if g[1] - g[0] > 15e6] + [kern[-1][1] + 1]
    lines = ["region  len_ms  proofs  proofs/s  busy_frac  kernel_ms_sum  gaps>0.2ms(n,total_ms)  uploads  upload_ms_mean  >=2_overlap_ms  >=4_overlap_ms  l1_avg_ms  ntt_avg_ms"]
    ki = 0
    for r in range(len(cuts) - 1):
        a, b = cuts[r], cuts[r + 1]
        ks = [(s, e, n) for s, e, n in kern if a <= s < b]
        if not ks:
            continue
        end = max(e for _, e, _ in ks)
        proofs = sum(1 for _, _, n in ks if "chacha20_fr_random_kernel" in n)
        if proofs < 8:
            continue
        busy, g = union([(s, e) for s, e, _ in ks])
        gg = [(y - x) / 1e6 for x, y in g if y - x > 0.2e6]
        cs = [(s, e) for s, e, _, _ in cop if a <= s < b]
        l1 = [(e - s) / 1e6 for s, e, n in ks if "msm_accum_seg_kernel<true>" in n or "msm_accum_seg_kernel<1" in n]
        nt = [(e - s) / 1e6 for s, e, n in ks if "ntt_step_kernel" in n]
        lines.append("%3d %8.1f %5d %7.1f %7.3f %10.1f   %4d %7.1f   %4d %7.2f %8.1f %8.1f   %.3f %.4f" % (
            r, (end - a) / 1e6, proofs, proofs / ((end - a) / 1e9), busy / (end - a), sum(e - s for s, e, _ in ks) / 1e6,
            len(gg), sum(gg), len(cs), (sum(e - s for s, e in cs) / len(cs) / 1e6) if cs else 0,
            overlap_time(cs, 2) / 1e6, overlap_time(cs, 4) / 1e6, sum(l1) / max(1, len(l1)), sum(nt) / max(1, len(nt))))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
